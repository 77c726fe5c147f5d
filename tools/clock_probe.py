#!/usr/bin/env python3
"""Sample the GPU's clock and power (rocm-smi) while one kernel mix runs in a loop:
usage: clock_probe.py nt|tn|idle|ln  -- prints a few samples taken during ~3 s of back-to-back launches."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402
from visiontransformer_amd._lib import BF16, check, lib  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "nt"
T, N, K = 256 * 197, 3072, 768
bf = torch.bfloat16
x = torch.randn(T, K, device="cuda").to(bf)
w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(bf)
y = torch.empty(T, N, device="cuda", dtype=bf)
dy = torch.randn(T, N, device="cuda").to(bf)
dw = torch.empty(N, K, device="cuda", dtype=torch.float32)
ws = torch.empty(max(lib.vitmi_gemm_wgrad_workspace_bytes(BF16, N, K, T), 16), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
stop = False


def sampler():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if "sclk" in l or "mclk" in l or "ower" in l]
        print(" | ".join(l.split(":", 1)[-1].strip() for l in lines), flush=True)
        time.sleep(0.4)


th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
n = 0
while time.time() - t0 < 3.0:
    for _ in range(50):
        if kind == "nt":
            ops.gemm_raw(BF16, T, N, K, x, K, 1, w, K, 1, y, N, BF16)
        elif kind == "tn":
            check(lib.vitmi_gemm_wgrad(BF16, N, K, T, dy.data_ptr(), N, x.data_ptr(), K, dw.data_ptr(), K, ws.data_ptr(), s))
        elif kind == "ln":
            ops.add(dy, dy)
        n += 1
    torch.cuda.synchronize()
dt = time.time() - t0
stop = True
th.join()
print(kind, "launches", n, "avg ms", dt / max(n, 1) * 1e3)
