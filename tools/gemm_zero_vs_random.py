#!/usr/bin/env python3
"""Is the ring GEMM limited by the clock the chip holds under load?  Same launches on random and on all-zero operands (zeros toggle
nothing: the chip keeps a higher clock; the instruction stream and its cycle count are identical)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib
lib = _lib.load(); st = _lib.stream_ptr()
ws = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
m = 7040
for name, n, k in (("qkv", 12288, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)):
    for kind in ("random", "zeros"):
        a = (torch.randn(m, k, device="cuda") if kind == "random" else torch.zeros(m, k, device="cuda")).to(torch.bfloat16)
        w = ((torch.randn(n, k, device="cuda") * 0.02) if kind == "random" else torch.zeros(n, k, device="cuda")).to(torch.bfloat16)
        c = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
        f = lambda: _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, k, n, _lib.ATSPEED_BF16, _lib.EPI_STORE, ws.data_ptr(), ws.numel(), st))
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        print(f"{name:8s} M={m} {kind:7s} {us:8.1f} us {2.0 * m * n * k / us / 1e6:7.1f} TF", flush=True)
