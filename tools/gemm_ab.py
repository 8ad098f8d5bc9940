#!/usr/bin/env python3
"""Time atspeed_gemm at given shapes (A/B of env-selected kernel variants).  usage: gemm_ab.py [M ...]
GEMM_AB_EPI = 0 store (default), 2 residual add, 3 SwiGLU (the shape's N is then gate + up interleaved, output N / 2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib
if os.environ.get("ATSPEED_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["ATSPEED_LIB"])     # a tuning build (make stamps / ablate / exppack)
lib = _lib.load(); st = _lib.stream_ptr()
ws = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
Ms = [int(x) for x in sys.argv[1:]] or [3200, 7040]
EPI = int(os.environ.get("GEMM_AB_EPI", "0"))
for name, n, k in (("qkv", 12288, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)):
    for m in Ms:
        a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
        no = n // 2 if EPI == 3 else n
        c = torch.zeros(m + 1, no, dtype=torch.bfloat16, device="cuda")
        if os.environ.get("GEMM_AB_ROWMAJOR"):
            f = lambda: _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, k, no, _lib.ATSPEED_BF16, EPI, ws.data_ptr(), ws.numel(), st))
        else:         # the engine's layout: packed operands (the random values do not care that they were not run through atspeed_pack_rows)
            f = lambda: _lib.check(lib.atspeed_gemm_packed(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, no, EPI, ws.data_ptr(), ws.numel(), st))
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{name:8s} epi {EPI} M={m:5d}  {us:8.1f} us {2.0 * m * n * k / us / 1e6:7.1f} TF", flush=True)
