#!/usr/bin/env python3
"""Yardstick only (not on the product path): hipBLASLt via torch.matmul vs atspeed_gemm on the batched-forward shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib
lib = _lib.load()
st = _lib.stream_ptr()
ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
def timeit(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for name, n, k in (("qkv", 12288, 4096), ("o_proj", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)):
    for m in (2048, 3200, 7040):
        a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
        c = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
        t_lt = timeit(lambda: torch.matmul(a, w.t(), out=c))
        t_my = timeit(lambda: _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, k, n, _lib.ATSPEED_BF16, _lib.EPI_STORE, ws.data_ptr(), ws.numel(), st)))
        fl = 2.0 * m * n * k
        print(f"{name:8s} M={m:5d} N={n:6d} K={k:6d}  hipBLASLt {t_lt:8.1f} us {fl / t_lt / 1e6:7.1f} TF | atspeed {t_my:8.1f} us {fl / t_my / 1e6:7.1f} TF", flush=True)
