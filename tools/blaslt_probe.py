#!/usr/bin/env python3
"""Yardstick probe: which hipBLASLt kernels torch.matmul picks for the batched-forward shapes (names encode the tiling)."""
import torch
for n, k in ((12288, 4096), (22016, 4096), (4096, 11008), (4096, 4096)):
    for m in (3200, 7040):
        a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
        c = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
        for _ in range(3):
            torch.matmul(a, w.t(), out=c)
        torch.cuda.synchronize()
