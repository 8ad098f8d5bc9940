#!/usr/bin/env python3
"""Panel form of the ring kernel (gemm_ring_kernel<..., WN = 2, WM = 4>, 257-512 tokens) against what the dispatch did before it, on the
Llama-7B projections as the ENGINE runs them (packed operands, the projection's own epilogue, rotating weights): us per launch with
the `gemm_panel` switch 0 / 1 (atspeed_set_switch; round 6's K-cut form held off).  The split forms include their reduce launch.
usage: python tools/panel_sweep.py [M list]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib
lib = _lib.load(); st = _lib.stream_ptr()
ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
def pack(t):
    rows, cols = t.shape
    out = torch.empty((rows + 1) // 2 * 2, cols, dtype=t.dtype, device="cuda")
    _lib.check(lib.atspeed_pack_rows(t.data_ptr(), out.data_ptr(), rows, cols * 2, st)); return out
def timeit(fs, iters=30):
    for f in fs[:3]: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fs[i % len(fs)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
Ms = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "257,300,320,384,400,456,512").split(",")]
for name, n, k, epi in (("qkv", 12288, 4096, _lib.EPI_STORE), ("o_proj", 4096, 4096, _lib.EPI_RESID), ("gate_up", 22016, 4096, _lib.EPI_SWIGLU), ("down", 4096, 11008, _lib.EPI_RESID)):
    wl = [pack((torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)) for _ in range(5)]
    for m in Ms:
        a = pack(torch.randn(m, k, device="cuda").to(torch.bfloat16))
        ldc = {_lib.EPI_STORE: n, _lib.EPI_RESID: n, _lib.EPI_SWIGLU: n // 2}[epi]
        c = torch.zeros((m + 1) // 2 * 2, ldc, dtype=torch.bfloat16, device="cuda")
        fs = [(lambda w=w: _lib.check(lib.atspeed_gemm_packed(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, ldc, epi, ws.data_ptr(), ws.numel(), st))) for w in wl]
        cells = []
        for p in (0, 1):
            with _lib.switches(gemm_panel=p, gemm_kcut=0):
                cells.append(timeit(fs))
        flops = 2.0 * m * n * k
        print(f"{name:8s} M={m:4d}  before {cells[0]:6.1f} us  panel {cells[1]:6.1f} us  ({cells[0] / cells[1]:.2f}x; panel = {flops / cells[1] / 1e6:6.0f} TF, weights {n * k * 2 / cells[1] / 1e3:5.0f} GB/s)", flush=True)
    del wl
