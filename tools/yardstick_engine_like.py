#!/usr/bin/env python3
"""One user's wide projections as the ENGINE runs them (packed operands, SwiGLU epilogue for gate_up, fp32 logits for the lm_head), cold weights:
us per launch and GB/s of W at 20 / 60 / 100 / 225 tokens.  A/B with ATSPEED_GEMM_WDMA=0 (tools/ab_env.sh)."""
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
def timeit(fs, iters=40):
    for f in fs[:3]: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fs[i % len(fs)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for name, n, k, epi in (("qkv", 12288, 4096, _lib.EPI_STORE), ("o_proj", 4096, 4096, _lib.EPI_RESID), ("gate_up", 22016, 4096, _lib.EPI_SWIGLU), ("down", 4096, 11008, _lib.EPI_RESID),
                        ("lm_head", 32859, 4096, _lib.EPI_F32)):
    wl = [pack((torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)) for _ in range(6)]     # rotate weights: no cache-resident W
    for m in [int(x) for x in os.environ.get("YARD_M", "20,60,100,121,225").split(",")]:
        a = pack(torch.randn(m, k, device="cuda").to(torch.bfloat16))
        ldc = {_lib.EPI_STORE: n, _lib.EPI_RESID: n, _lib.EPI_SWIGLU: n // 2, _lib.EPI_F32: (n + 63) // 64 * 64}[epi]
        c = torch.zeros((m + 1) // 2 * 2, ldc, dtype=torch.float32 if epi == _lib.EPI_F32 else torch.bfloat16, device="cuda")
        t = timeit([(lambda w=w: _lib.check(lib.atspeed_gemm_packed(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, ldc, epi, ws.data_ptr(), ws.numel(), st))) for w in wl])
        print(f"{name:8s} M={m:4d}  {t:7.1f} us {n * k * 2 / t / 1e3:6.0f} GB/s", flush=True)
    del wl
