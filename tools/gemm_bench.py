#!/usr/bin/env python3
"""Micro-benchmark of the projection GEMMs at the Llama-7B shapes of the target forward
(tools for kernel tuning; prints us per launch and algorithmic GB/s of the weight stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib

lib = _lib.load()
SHAPES = {"qkv": (12288, 4096, _lib.EPI_STORE), "o_proj": (4096, 4096, _lib.EPI_RESID),
          "gate_up": (22016, 4096, _lib.EPI_SWIGLU), "down": (4096, 11008, _lib.EPI_RESID),
          "lm_head": (32859, 4096, _lib.EPI_F32)}
Ms = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [20, 60, 100, 228]
ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
st = _lib.stream_ptr()
# several weight copies so that consecutive launches do not hit a cache-resident matrix
for name, (n, k, epi) in SHAPES.items():
    nrep = 6
    ws_list = [torch.randn(n, k, device="cuda").to(torch.bfloat16) * 0.02 for _ in range(nrep)]
    for m in Ms:
        a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        if epi == _lib.EPI_F32:
            ldc = (n + 63) // 64 * 64
            c = torch.zeros(m, ldc, dtype=torch.float32, device="cuda")
        elif epi == _lib.EPI_SWIGLU:
            ldc = n // 2
            c = torch.zeros(m, ldc, dtype=torch.bfloat16, device="cuda")
        else:
            ldc = n
            c = torch.zeros(m, ldc, dtype=torch.bfloat16, device="cuda")
        def run(i):
            _lib.check(lib.atspeed_gemm(a.data_ptr(), ws_list[i % nrep].data_ptr(), c.data_ptr(), m, n, k, k, ldc, _lib.ATSPEED_BF16, epi,
                                        ws.data_ptr(), ws.numel(), st))
        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 30
        e0.record()
        for i in range(iters):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        print(f"{name:8s} M={m:4d} N={n:6d} K={k:6d}  {us:8.1f} us  {n * k * 2 / us / 1e3:7.0f} GB/s  {2.0 * m * n * k / us / 1e6:7.1f} TF")
    del ws_list
