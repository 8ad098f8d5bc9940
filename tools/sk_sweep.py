#!/usr/bin/env python3
"""Split-K tail of the ring GEMM (gemm_ring_kernel<..., SK>) on the Llama-7B projections in the 4-64-user band, as the ENGINE runs them (packed
operands, the projection's own epilogue, cold weights): us per launch for every (token-tile height, parts per tail tile), next to the default
dispatch.  The cost model in gemm.hip (sk_cost_us) is fitted to this table.  The `gemm_force_mt` / `gemm_sk` switches are set per cell (atspeed_set_switch).
usage: python tools/sk_sweep.py [M list]"""
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
Ms = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "320,400,640,912,1200,1600,2400,3650").split(",")]
kinds = os.environ.get("SWEEP_KINDS", "qkv,o_proj,gate_up,down").split(",")
for name, n, k, epi in (("qkv", 12288, 4096, _lib.EPI_STORE), ("o_proj", 4096, 4096, _lib.EPI_RESID), ("gate_up", 22016, 4096, _lib.EPI_SWIGLU), ("down", 4096, 11008, _lib.EPI_RESID)):
    if name not in kinds: continue
    wl = [pack((torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)) for _ in range(5)]
    for m in Ms:
        a = pack(torch.randn(m, k, device="cuda").to(torch.bfloat16))
        ldc = {_lib.EPI_STORE: n, _lib.EPI_RESID: n, _lib.EPI_SWIGLU: n // 2}[epi]
        c = torch.zeros((m + 1) // 2 * 2, ldc, dtype=torch.bfloat16, device="cuda")
        fs = [(lambda w=w: _lib.check(lib.atspeed_gemm_packed(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, ldc, epi, ws.data_ptr(), ws.numel(), st))) for w in wl]
        tn = (n + 255) // 256
        row = [f"{name:8s} M={m:5d} t256={tn * ((m + 255) // 256):5d} t128={tn * ((m + 127) // 128):5d}"]
        with _lib.switches(gemm_force_mt=0, gemm_sk=1, gemm_kcut=0):
            row.append(f"default {timeit(fs):6.1f}")
        for mt in (8, 4):
            cells = []
            for S in (0, 2, 3, 4):
                with _lib.switches(gemm_force_mt=mt, gemm_sk=S, gemm_kcut=0):
                    cells.append(f"S{S} {timeit(fs):6.1f}")
            row.append(f"mt{mt}: " + " ".join(cells))
        print(" | ".join(row), flush=True)
    del wl
