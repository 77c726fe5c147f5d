import torch
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
n = 256*12*197*197
x = torch.empty(n, dtype=torch.bfloat16, device="cuda"); y = torch.empty_like(x)
us = timeit(lambda: x.fill_(1.0)); print(f"fill {n*2/1e6:.0f} MB: {us:.1f} us {n*2/us/1e6:.2f} TB/s")
us = timeit(lambda: y.copy_(x)); print(f"copy: {us:.1f} us {2*n*2/us/1e6:.2f} TB/s")
z = torch.empty(4*n, dtype=torch.bfloat16, device="cuda")
us = timeit(lambda: z.fill_(1.0)); print(f"fill {4*n*2/1e6:.0f} MB: {us:.1f} us {4*n*2/us/1e6:.2f} TB/s")
