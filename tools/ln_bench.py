import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
rows, D = 50432, 768
x = torch.randn(rows, D, device="cuda").bfloat16(); dy = torch.randn_like(x); dres = torch.randn_like(x)
g = torch.randn(D, device="cuda"); b = torch.randn(D, device="cuda")
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
us = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-5)); print(f"ln fwd {us:.1f} us {2*rows*D*2/us/1e6:.2f} TB/s")
us = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, want_dres_colsum=True)); print(f"ln bwd+res {us:.1f} us {4*rows*D*2/us/1e6:.2f} TB/s")
us = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd)); print(f"ln bwd {us:.1f} us {3*rows*D*2/us/1e6:.2f} TB/s")
