#!/usr/bin/env python3
"""One projection GEMM shape, a few launches (for rocprofv3 --pmc passes).  usage: one_gemm.py M N K [epilogue=0] [rowmajor]
epilogue 0 store, 2 residual add, 3 SwiGLU (N = gate + up interleaved); operands packed as in the engine unless `rowmajor`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib
lib = _lib.load(); st = _lib.stream_ptr()
m, n, k = (int(x) for x in sys.argv[1:4])
epi = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4].isdigit() else 0
rowmajor = "rowmajor" in sys.argv[4:]
ws = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
a = torch.randn(m + 1, k, device="cuda").to(torch.bfloat16)
w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
no = n // 2 if epi == 3 else n
c = torch.zeros(m + 1, no, dtype=torch.bfloat16, device="cuda")
for _ in range(5):
    if rowmajor:
        _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, k, no, _lib.ATSPEED_BF16, epi, ws.data_ptr(), ws.numel(), st))
    else:       # random values do not care that they were not run through atspeed_pack_rows
        _lib.check(lib.atspeed_gemm_packed(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, no, epi, ws.data_ptr(), ws.numel(), st))
torch.cuda.synchronize()
