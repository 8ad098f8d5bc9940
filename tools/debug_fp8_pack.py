#!/usr/bin/env python3
"""one-off: atspeed_gemm_fp8 (SwiGLU) row-major vs packed, twice each: where do they differ?  usage: debug_fp8_pack.py M N K"""
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
def pack(t):
    out = torch.empty((t.shape[0] + 1) // 2 * 2, t.shape[1], dtype=t.dtype, device="cuda")
    _lib.check(lib.atspeed_pack_rows(t.data_ptr(), out.data_ptr(), t.shape[0], t.shape[1] * t.element_size(), st)); return out
def unpack(t, rows):
    out = torch.empty(rows, t.shape[1], dtype=t.dtype, device="cuda")
    _lib.check(lib.atspeed_unpack_rows(t.data_ptr(), out.data_ptr(), rows, t.shape[1] * t.element_size(), st)); return out
ldc = n // 2
xp, wp = pack(xq), pack(wq)
outs = []
for rep in range(2):
    c0 = torch.zeros((m + 1) // 2 * 2, ldc, dtype=torch.bfloat16, device="cuda"); c1 = torch.zeros_like(c0)
    _lib.check(lib.atspeed_gemm_fp8(xq.data_ptr(), sx.data_ptr(), wq.data_ptr(), sw.data_ptr(), c0.data_ptr(), m, n, k, ldc, 3, st))
    _lib.check(lib.atspeed_gemm_fp8_packed(xp.data_ptr(), sx.data_ptr(), wp.data_ptr(), sw.data_ptr(), c1.data_ptr(), m, n, k, ldc, 3, st))
    torch.cuda.synchronize()
    outs.append((c0[:m].clone(), unpack(c1, m)))
print("row-major run 0 == run 1:", torch.equal(outs[0][0], outs[1][0]), " packed run 0 == run 1:", torch.equal(outs[0][1], outs[1][1]))
bad = outs[0][0] != outs[0][1]
print("differing elements", int(bad.sum()), "of", bad.numel())
if bad.any():
    rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
    print("rows:", rows[:24].tolist(), "... count", len(rows), "last", rows[-8:].tolist()); print("cols:", cols[:48].tolist(), "... count", len(cols), "last", cols[-8:].tolist())
