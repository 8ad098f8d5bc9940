#!/usr/bin/env python3
"""One W8A8 projection GEMM shape, a few launches (for rocprofv3 --pmc passes).  usage: one_gemm_fp8.py M N K [epilogue=0]
Packed e4m3 operands as in the engine, a workspace for the K-split forms; 6 weight copies in rotation (no cache-resident W)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib
lib = _lib.load(); st = _lib.stream_ptr()
m, n, k = (int(x) for x in sys.argv[1:4])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ws = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
rnd = lambda r, kk: (torch.randn(r, kk, device="cuda").clamp(-3, 3) * 60).to(torch.float8_e4m3fn).view(torch.uint8)
xq = rnd(m + 1, k); wl = [rnd(n, k) for _ in range(6)]
sx = torch.full((m + 1,), 0.01, device="cuda"); sw = torch.full((n,), 0.001, device="cuda")
no = n // 2 if epi == 3 else n
c = torch.zeros(m + 1, no, dtype=torch.bfloat16, device="cuda")
for i in range(12):
    _lib.check(lib.atspeed_gemm_fp8_packed(xq.data_ptr(), sx.data_ptr(), wl[i % 6].data_ptr(), sw.data_ptr(), c.data_ptr(), m, n, k, no, epi, ws.data_ptr(), ws.numel(), st))
torch.cuda.synchronize()
