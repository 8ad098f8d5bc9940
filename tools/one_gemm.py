#!/usr/bin/env python3
"""One projection GEMM shape, a few launches (for rocprofv3 --pmc passes).  usage: one_gemm.py M N K [fp8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib
lib = _lib.load(); st = _lib.stream_ptr()
m, n, k = (int(x) for x in sys.argv[1:4])
ws = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
c = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
for _ in range(5):
    _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, k, n, _lib.ATSPEED_BF16, _lib.EPI_STORE, ws.data_ptr(), ws.numel(), st))
torch.cuda.synchronize()
