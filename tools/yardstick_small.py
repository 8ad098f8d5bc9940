#!/usr/bin/env python3
"""Yardstick only: hipBLASLt (torch.matmul) vs atspeed_gemm at ONE user's token counts (weight-streaming regime): GB/s of W."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib
lib = _lib.load(); st = _lib.stream_ptr()
ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
def timeit(fs, iters=30):
    for f in fs[:3]: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fs[i % len(fs)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for name, n, k in (("qkv", 12288, 4096), ("o_proj", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)):
    wl = [(torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(6)]     # rotate weights: no cache-resident W
    for m in (20, 60, 100, 220):
        a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        c = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
        t_lt = timeit([(lambda w=w: torch.matmul(a, w.t(), out=c)) for w in wl])
        t_my = timeit([(lambda w=w: _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, k, n, _lib.ATSPEED_BF16, _lib.EPI_STORE, ws.data_ptr(), ws.numel(), st))) for w in wl])
        print(f"{name:8s} M={m:4d}  hipBLASLt {t_lt:7.1f} us {n * k * 2 / t_lt / 1e3:6.0f} GB/s | atspeed {t_my:7.1f} us {n * k * 2 / t_my / 1e3:6.0f} GB/s", flush=True)
    del wl
