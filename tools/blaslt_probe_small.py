#!/usr/bin/env python3
"""Yardstick probe: which hipBLASLt kernels torch.matmul picks at ONE user's token counts (run under rocprofv3 --kernel-trace --stats)."""
import torch
for n, k in ((12288, 4096), (22016, 4096), (4096, 11008), (4096, 4096)):
    for m in (20, 100, 220):
        a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        wl = [(torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(4)]
        c = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
        for i in range(12):
            torch.matmul(a, wl[i % 4].t(), out=c)
        torch.cuda.synchronize()
