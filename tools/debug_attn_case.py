#!/usr/bin/env python3
"""Which rows / heads / columns of one attention case differ from the torch reference.  usage: debug_attn_case.py heads dh T [qtile rows_per_wave]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atspeed_amd import _lib, synth
from atspeed_amd.model import vis_bits_from_bool
lib = _lib.load(); st = _lib.stream_ptr()
heads, dh, T = (int(x) for x in sys.argv[1:4])
qtile, rpw = (int(x) for x in sys.argv[4:6]) if len(sys.argv) > 5 else (0, 0)
S, max_slots = 150, 256
H = heads * dh
rnd = lambda shape, seed, std=1.0: torch.from_numpy(synth.hash_normal(int(np.prod(shape)), seed, std).reshape(shape))
q = rnd((T, 3 * H), 31).to(torch.bfloat16).cuda(); kc = rnd((max_slots, H), 32).to(torch.bfloat16).cuda(); vc = rnd((max_slots, H), 33).to(torch.bfloat16).cuda()
g = torch.Generator().manual_seed(5)
vis = torch.rand(T, S, generator=g) < 0.3
vis[:, 0] = True; vis[3] = False; vis[3, 149] = True
bits = vis_bits_from_bool(vis, max_slots).cuda()
out = torch.zeros(T, H, dtype=torch.bfloat16, device="cuda")
_lib.check(lib.atspeed_tree_attention_tiled(q.data_ptr(), 3 * H, kc.data_ptr(), vc.data_ptr(), bits.data_ptr(), max_slots // 64, out.data_ptr(), T, S, heads, dh, _lib.ATSPEED_BF16, qtile, rpw, st))
torch.cuda.synchronize()
qf = q.float().cpu()[:, :H].view(T, heads, dh); kf = kc.float().cpu()[:S].view(S, heads, dh); vf = vc.float().cpu()[:S].view(S, heads, dh)
sc = torch.einsum("thd,shd->hts", qf, kf) / np.sqrt(dh)
sc = sc.masked_fill(~vis[None], float("-inf"))
ref = torch.einsum("hts,shd->thd", torch.softmax(sc, -1), vf)
bad = (out.float().cpu().view(T, heads, dh) - ref).abs() > 3e-2
print("bad elements", int(bad.sum()), "of", bad.numel())
print("bad rows:", sorted(set(torch.nonzero(bad)[:, 0].tolist())))
print("bad heads:", sorted(set(torch.nonzero(bad)[:, 1].tolist())))
print("bad cols:", sorted(set(torch.nonzero(bad)[:, 2].tolist()))[:70])
