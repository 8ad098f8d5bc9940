#!/usr/bin/env python3
"""Do two independent streams of the same GEMM sequence fill each other's tile-grid tails?  One layer's four projections
at M tokens, back to back, on 1 stream (2x the launches) vs 2 streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib
lib = _lib.load()
ws = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
SH = (("qkv", 12288, 4096, 0), ("o", 4096, 4096, 2), ("gu", 22016, 4096, 3), ("down", 4096, 11008, 2))
def mk(m):
    bufs = []
    for name, n, k, epi in SH:
        a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
        c = torch.zeros(m, n // 2 if epi == 3 else n, dtype=torch.bfloat16, device="cuda")
        bufs.append((a, w, c, m, n, k, epi))
    return bufs
def layer(bufs, st):
    for a, w, c, m, n, k, epi in bufs:
        _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, k, c.shape[1], _lib.ATSPEED_BF16, epi, ws.data_ptr(), ws.numel(), st))
for m in [int(x) for x in sys.argv[1:]] or [1920, 3200, 7040]:
    A, B = mk(m), mk(m)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    fl = sum(2.0 * m * n * k for _, n, k, _ in SH)
    reps = 10
    for mode in ("one stream", "two streams"):
        for _ in range(2):
            layer(A, s1.cuda_stream); layer(B, s1.cuda_stream if mode == "one stream" else s2.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            layer(A, s1.cuda_stream); layer(B, s1.cuda_stream if mode == "one stream" else s2.cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"M={m:5d} {mode:12s} {dt / reps * 1e3:8.3f} ms per 2 layers  {2 * fl * reps / dt / 1e12:7.1f} TF", flush=True)
