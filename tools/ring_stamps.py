#!/usr/bin/env python3
"""Where do the ring GEMM's waves spend their cycles?  Runs the tuning build (`make -C atspeed_amd/csrc stamps` ->
tools/probe/libatspeed_stamps.so, -DATS_RING_STAMPS) and prints, per shape, the share of loop cycles in {MFMA issue + LDS wait, vmcnt
wait, barrier} and the wall-clock phases of a workgroup (prologue / k-loop / epilogue incl. store acknowledgement) in shader cycles.
usage: ring_stamps.py M N K [epilogue] [random|zeros] [packed|rowmajor]   (M <= 256: the split-K mode one user's projections run in;
zeros: all-zero operands; packed (default): operands in the engine's packed layout, rowmajor: HF layout)
Also prints the clock the chip held inside the k-loop: cycle counter / 100 MHz real-time counter, median over waves (MI355X_MICROARCH.md, DVFS item 6)."""
import ctypes as C, os, sys
import torch
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (7040, 22016, 4096)
EPI = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # 0 store, 2 residual add, 3 SwiGLU (N = 2 * ffn, interleaved)
NO = N // 2 if EPI == 3 else N
ZERO = len(sys.argv) > 5 and sys.argv[5] == "zeros"
PACKED = not (len(sys.argv) > 6 and sys.argv[6] == "rowmajor")
dbg = torch.zeros(16384 * 8 * 10, dtype=torch.int64, device="cuda")
os.environ["ATSPEED_STAMP_PTR"] = hex(dbg.data_ptr())
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libatspeed_stamps.so"))
a = (torch.zeros(M, K, device="cuda") if ZERO else torch.randn(M, K, device="cuda")).to(torch.bfloat16)
ws_ = [((torch.zeros(N, K, device="cuda") if ZERO else torch.randn(N, K, device="cuda")) * 0.02).to(torch.bfloat16) for _ in range(4)]      # rotate: cold weights
c = torch.zeros(M, NO, dtype=torch.bfloat16, device="cuda")
ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
P = C.c_void_p
lib.atspeed_gemm.argtypes = [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, C.c_size_t, P]
lib.atspeed_gemm_packed.argtypes = [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, C.c_size_t, P]
lib.atspeed_pack_rows.argtypes = [P, P, C.c_int, C.c_int, P]
def pack(t):
    out = torch.empty((t.shape[0] + 1) // 2 * 2, t.shape[1], dtype=t.dtype, device="cuda")
    assert lib.atspeed_pack_rows(t.data_ptr(), out.data_ptr(), t.shape[0], t.shape[1] * 2, None) == 0
    return out
if PACKED:
    a = pack(a); ws_ = [pack(w) for w in ws_]; c = torch.zeros((M + 1) // 2 * 2, NO, dtype=torch.bfloat16, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
REPS = 40 if M > 1024 else 8            # large shapes: long enough under load for the clock to settle
for i in range(REPS):
    if i == REPS - 1: dbg.zero_(); torch.cuda.synchronize(); e0.record()
    if PACKED: rc = lib.atspeed_gemm_packed(a.data_ptr(), ws_[i % 4].data_ptr(), c.data_ptr(), M, N, K, NO, EPI, ws.data_ptr(), ws.numel(), None)
    else:      rc = lib.atspeed_gemm(a.data_ptr(), ws_[i % 4].data_ptr(), c.data_ptr(), M, N, K, K, NO, 1, EPI, ws.data_ptr(), ws.numel(), None)
    assert rc == 0
e1.record(); torch.cuda.synchronize()
t = dbg.view(-1, 8, 10).cpu()
t = t[t[:, 0, 4] != 0].double()                       # workgroups that ran
nwg = t.shape[0]
loop = t[..., :4]; tot = loop.sum()
print(f"M={M} N={N} K={K} epilogue {EPI} ({'packed' if PACKED else 'row-major'} operands): {nwg} workgroups, last launch incl. reduce {e0.elapsed_time(e1) * 1e3:.1f} us")
print(f" loop cycles: issue+lds {loop[..., 0].sum() / tot:.3f}+{loop[..., 1].sum() / tot:.3f}  vmcnt {loop[..., 2].sum() / tot:.3f}  barrier {loop[..., 3].sum() / tot:.3f}")
t0 = t[..., 4].min()
ent, l0, l1, end = (t[..., i] - t0 for i in (4, 5, 6, 7))
print(f" workgroup phases (cycles, mean over waves): entry spread {ent.mean():.0f} (max {ent.max():.0f});  prologue {(l0 - ent).mean():.0f};  "
      f"k-loop {(l1 - l0).mean():.0f};  epilogue+store ack {(end - l1).mean():.0f};  last wave ends at {end.max():.0f}")
clk = (t[..., 6] - t[..., 5]) / (t[..., 9] - t[..., 8]).clamp(min=1) * 0.1          # GHz
print(f" clock held inside the k-loop ({'zero' if ZERO else 'random'} operands): median {clk.median():.3f} GHz (min {clk.min():.3f}, max {clk.max():.3f})")
