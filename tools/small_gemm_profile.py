#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib
lib = _lib.load(); st = _lib.stream_ptr()
ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for name, n, k in (("qkv", 12288, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)):
    wl = [(torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(6)]
    for m in (100, 220):
        a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        c = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
        for i in range(12):
            _lib.check(lib.atspeed_gemm(a.data_ptr(), wl[i % 6].data_ptr(), c.data_ptr(), m, n, k, k, n, _lib.ATSPEED_BF16, _lib.EPI_STORE, ws.data_ptr(), ws.numel(), st))
        torch.cuda.synchronize()
