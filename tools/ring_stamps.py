#!/usr/bin/env python3
"""Where do the ring GEMM's waves spend their cycles?  Runs the tuning build (tools/probe/libatspeed_stamps.so, -DATS_RING_STAMPS) and
prints, per shape, the share of wave cycles in {MFMA issue + LDS wait, vmcnt wait, barrier}."""
import ctypes as C, os, sys
import torch
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (7040, 22016, 4096)
tiles = ((N + 255) // 256) * ((M + 255) // 256)
dbg = torch.zeros(tiles * 8 * 4, dtype=torch.int64, device="cuda")
os.environ["ATSPEED_STAMP_PTR"] = hex(dbg.data_ptr())
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libatspeed_stamps.so"))
a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
P = C.c_void_p
lib.atspeed_gemm.argtypes = [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, C.c_size_t, P]
for _ in range(3):
    rc = lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c.data_ptr(), M, N, K, K, N, 1, 0, None, 0, None)
    assert rc == 0
torch.cuda.synchronize()
t = dbg.view(tiles, 8, 4).double().cpu()
tot = t.sum()
print(f"M={M} N={N} K={K}: issue+lds {t[..., 0].sum() / tot:.3f}+{t[..., 1].sum() / tot:.3f}  vmcnt {t[..., 2].sum() / tot:.3f}  barrier {t[..., 3].sum() / tot:.3f}   cycles/k-step/wave {tot / (tiles * 8 * (K // 32)):.0f}")
by_wave = t.sum((0,))            # [8][4]
print(" per wave (barrier share):", [round(float(by_wave[w_, 3] / by_wave[w_].sum()), 3) for w_ in range(8)])
print(" per wave (vmcnt share):  ", [round(float(by_wave[w_, 2] / by_wave[w_].sum()), 3) for w_ in range(8)])
