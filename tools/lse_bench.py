#!/usr/bin/env python3
"""lse_rows_kernel on the verify round's logit block: GB/s vs the 8 TB/s HBM peak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib
lib = _lib.load(); st = _lib.stream_ptr()
V, ld = 32859, 32896
for rows in (121, 3872, 7744, 30976):
    lg = torch.randn(rows, ld, dtype=torch.float32, device="cuda"); out = torch.empty(rows, dtype=torch.float32, device="cuda")
    f = lambda: _lib.check(lib.atspeed_lse_rows(lg.data_ptr(), rows, V, ld, out.data_ptr(), st))
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    ref = torch.logsumexp(lg[:, :V], dim=1)
    print(f"rows {rows:5d}: {us:8.1f} us  {rows * V * 4 / us / 1e3:7.0f} GB/s  max err {float((out - ref).abs().max()):.2e}", flush=True)
