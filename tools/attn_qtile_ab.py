#!/usr/bin/env python3
"""One user's tree attention (32 heads x 128) by query-tile height: us per launch of atspeed_tree_attention_tiled at T query rows over S slots,
16 rows per wave (the one-user kernel), qtile 64 / 128 / 256, rotating K / V buffers.  usage: python tools/attn_qtile_ab.py [T,S ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import _lib
from atspeed_amd.model import vis_bits_from_bool
lib = _lib.load(); st = _lib.stream_ptr()
heads, dh, max_slots = 32, 128, 512
H = heads * dh
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(228, 228), (121, 350), (100, 330), (60, 290), (20, 250)]
for T, S in shapes:
    q = torch.randn(T, 3 * H, device="cuda").to(torch.bfloat16)
    kv = [(torch.randn(max_slots, H, device="cuda").to(torch.bfloat16), torch.randn(max_slots, H, device="cuda").to(torch.bfloat16)) for _ in range(6)]
    vis = torch.zeros(T, S, dtype=torch.bool); vis[:, : S - T] = True; vis[:, S - T:] = torch.tril(torch.ones(T, T, dtype=torch.bool))
    bits = vis_bits_from_bool(vis, max_slots).cuda()
    out = torch.empty(T, H, dtype=torch.bfloat16, device="cuda")
    cells = []
    for qt in (64, 128, 256):
        def f(i):
            k, v = kv[i % 6]
            _lib.check(lib.atspeed_tree_attention_tiled(q.data_ptr(), 3 * H, k.data_ptr(), v.data_ptr(), bits.data_ptr(), max_slots // 64, out.data_ptr(), T, S, heads, dh,
                                                        _lib.ATSPEED_BF16, qt, 16, st))
        for i in range(6): f(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(60): f(i)
        e1.record(); torch.cuda.synchronize()
        cells.append(f"qtile {qt}: {e0.elapsed_time(e1) * 1e3 / 60:6.1f} us")
    print(f"T={T:4d} S={S:4d}  " + "  ".join(cells), flush=True)
