#!/usr/bin/env python3
"""What bounds the block-scaled fp8 ring GEMM (gemm_ring_mx_kernel)?  Times atspeed_gemm_fp8 on the product library with random and
with all-zero operands (zeros toggle nothing: a power-limited kernel speeds up, a bandwidth- or latency-bound one does not), then on the
tuning builds of `make -C atspeed_amd/csrc ablate` with the main loop's DMA / fragment reads / MFMAs removed.
usage: mx_ablate.py [M N K]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (26000, 22016, 4096)
P = C.c_void_p
def bench(lib, xq, sx, wq, sw, c, epi=0):
    lib.atspeed_gemm_fp8.argtypes = [P, P, P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P]
    f = lambda: lib.atspeed_gemm_fp8(xq.data_ptr(), sx.data_ptr(), wq.data_ptr(), sw.data_ptr(), c.data_ptr(), M, N, K, N, epi, None)
    for _ in range(3): assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    return us, 2.0 * M * N * K / us / 1e6
main = _lib.load()
sx = torch.ones(M, device="cuda"); sw = torch.ones(N, device="cuda")
c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
rnd = lambda r, k: (torch.randn(r, k, device="cuda").clamp(-3, 3) * 60).to(torch.float8_e4m3fn).view(torch.uint8)
xr, wr = rnd(M, K), rnd(N, K)
xz, wz = torch.zeros_like(xr), torch.zeros_like(wr)
print(f"M={M} N={N} K={K}")
for name, (x, w) in (("random operands", (xr, wr)), ("zero operands", (xz, wz))):
    us, tf = bench(main, x, sx, w, sw, c)
    print(f"  product library, {name:16s} {us:9.1f} us {tf:8.1f} TF", flush=True)
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe")
for n, what in ((1, "no DMA"), (2, "no fragment reads"), (3, "no MFMAs")):
    path = os.path.join(here, f"libatspeed_ablate{n}.so")
    if not os.path.exists(path):
        print("  (no", path, "- run make -C atspeed_amd/csrc ablate)"); continue
    us, tf = bench(C.CDLL(path), xr, sx, wr, sw, c)
    print(f"  ablation build, {what:17s} {us:9.1f} us {tf:8.1f} TF-equivalent", flush=True)
