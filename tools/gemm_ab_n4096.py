#!/usr/bin/env python3
"""o_proj / down (N = 4096) at the token counts of the small rounds: ring kernel vs LDS-tiled kernel (ATSPEED_GEMM_BIG_MIN_FILL)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib
lib = _lib.load(); st = _lib.stream_ptr()
ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for name, n, k in (("o_proj", 4096, 4096), ("down", 4096, 11008), ("qkv", 12288, 4096), ("gate_up", 22016, 4096)):
    wl = [(torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(4)]
    for m in (512, 640, 900, 1280, 1920):
        a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        c = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
        fs = [(lambda w=w: _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, k, n, _lib.ATSPEED_BF16, _lib.EPI_STORE, ws.data_ptr(), ws.numel(), st))) for w in wl]
        for f in fs: f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20): fs[i % 4]()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{name:8s} M={m:5d}  {us:8.1f} us {2.0 * m * n * k / us / 1e6:7.1f} TF", flush=True)
