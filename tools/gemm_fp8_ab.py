#!/usr/bin/env python3
"""Time atspeed_gemm_fp8 on the Llama-7B projection shapes (A/B of env-selected kernel variants: ATSPEED_FP8_MX; ATSPEED_LIB = another build of the library).
usage: gemm_fp8_ab.py [--no-ws] [M ...]   (--no-ws: no workspace, i.e. without the K-split forms of thin grids / narrow projections)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib
if os.environ.get("ATSPEED_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["ATSPEED_LIB"])     # another build of the library
lib = _lib.load(); st = _lib.stream_ptr()
NO_WS = "--no-ws" in sys.argv
Ms = [int(x) for x in sys.argv[1:] if x != "--no-ws"] or [7040, 26000]
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
WS = (None, 0) if NO_WS else (ws.data_ptr(), ws.numel())
rnd = lambda r, k: (torch.randn(r, k, device="cuda").clamp(-3, 3) * 60).to(torch.float8_e4m3fn).view(torch.uint8)
for name, n, k, epi in (("qkv", 12288, 4096, 0), ("o_proj", 4096, 4096, 2), ("gate_up", 22016, 4096, 3), ("down", 4096, 11008, 2)):
    for m in Ms:
        xq, wq = rnd(m, k), rnd(n, k)
        sx = torch.full((m,), 0.01, device="cuda"); sw = torch.full((n,), 0.001, device="cuda")
        ldc = n // 2 if epi == 3 else n
        c = torch.zeros(m, ldc, dtype=torch.bfloat16, device="cuda")
        f = lambda: _lib.check(lib.atspeed_gemm_fp8(xq.data_ptr(), sx.data_ptr(), wq.data_ptr(), sw.data_ptr(), c.data_ptr(), m, n, k, ldc, epi, WS[0], WS[1], st))
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 10
        print(f"{name:8s} M={m:5d}  {us:8.1f} us {2.0 * m * n * k / us / 1e6:7.1f} TF", flush=True)
