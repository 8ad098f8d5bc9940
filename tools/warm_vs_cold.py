#!/usr/bin/env python3
"""One user's projections with the weights cold (rotating over 6 copies, > the 256 MB Infinity Cache) vs warm (same copy every launch):
how much of a small-M launch is the HBM stream and how much is fixed cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib
lib = _lib.load(); st = _lib.stream_ptr()
ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
def timeit(fs, iters=36):
    for f in fs[:6]: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fs[i % len(fs)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for name, n, k in (("qkv", 12288, 4096), ("o_proj", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)):
    wl = [(torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(6)]
    for m in [int(x) for x in os.environ.get('MS', '110,200').split(',')]:
        a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        c = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
        g = lambda w: (lambda: _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, k, n, _lib.ATSPEED_BF16, _lib.EPI_STORE, ws.data_ptr(), ws.numel(), st)))
        cold = timeit([g(w) for w in wl]); warm = timeit([g(wl[0])])
        mb = n * k * 2 / 1e6
        print(f"{name:8s} M={m:4d} W={mb:6.1f} MB  cold {cold:6.1f} us ({mb / cold / 1e3 * 1e3:5.0f} GB/s)  warm {warm:6.1f} us  -> HBM stream at 6 TB/s would be {mb / 6e3 * 1e3 / 1e3 * 1e3:5.1f} us", flush=True)
    del wl
