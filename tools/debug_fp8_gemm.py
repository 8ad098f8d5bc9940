#!/usr/bin/env python3
"""one-off: where does atspeed_gemm_fp8 differ from the dequantised reference?  usage: debug_fp8_gemm.py M N K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib
lib = _lib.load(); st = _lib.stream_ptr()
m, n, k = (int(x) for x in sys.argv[1:4])
torch.manual_seed(0)
x = (torch.randn(m, k, device="cuda") * 1.5).to(torch.bfloat16)
w = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
xq = torch.empty(m, k, dtype=torch.uint8, device="cuda"); sx = torch.empty(m, device="cuda")
wq = torch.empty(n, k, dtype=torch.uint8, device="cuda"); sw = torch.empty(n, device="cuda")
_lib.check(lib.atspeed_quant_rows_fp8(x.data_ptr(), m, k, xq.data_ptr(), sx.data_ptr(), st))
_lib.check(lib.atspeed_quant_rows_fp8(w.data_ptr(), n, k, wq.data_ptr(), sw.data_ptr(), st))
ref = (xq.view(torch.float8_e4m3fn).float() @ wq.view(torch.float8_e4m3fn).float().T) * sx[:, None] * sw[None, :]
c = torch.zeros(m, n, dtype=torch.bfloat16, device="cuda")
_lib.check(lib.atspeed_gemm_fp8(xq.data_ptr(), sx.data_ptr(), wq.data_ptr(), sw.data_ptr(), c.data_ptr(), m, n, k, n, 0, st))
torch.cuda.synchronize()
bad = (c.float() - ref).abs() > 0.03 * float(ref.abs().max())
print("bad elements", int(bad.sum()), "of", bad.numel())
if bad.any():
    rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
    print("bad rows:", rows[:40].tolist(), "... count", len(rows)); print("bad cols:", cols[:80].tolist(), "... count", len(cols))
    r0 = int(rows[0]); bc = bad[r0].nonzero().flatten()[:8].tolist()
    print("row", r0, "cols", bc, "got", [round(float(c[r0, j]), 3) for j in bc], "want", [round(float(ref[r0, j]), 3) for j in bc])
    # is the value some other column's?
    for j in bc[:4]:
        near = (ref[r0] - float(c[r0, j])).abs().argmin()
        print("  col", j, "holds the value of col", int(near), "(diff", round(float((ref[r0, near] - c[r0, j]).abs()), 4), ")")
