#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atspeed_amd import _lib
from atspeed_amd.model import vis_bits_from_bool
lib = _lib.load(); st = _lib.stream_ptr()
heads, dh, T, S, max_slots = 2, 128, 16, 150, 256
H = heads * dh
q = torch.zeros(T, 3 * H, dtype=torch.bfloat16, device="cuda")          # all scores 0 -> uniform attention
kc = torch.zeros(max_slots, H, dtype=torch.bfloat16, device="cuda")
# V[key][d] = key + d/256  -> mean over visible keys identifies which keys / columns arrive
key = torch.arange(max_slots, dtype=torch.float32)[:, None]
col = torch.arange(H, dtype=torch.float32)[None, :]
vc = (col % dh).expand(max_slots, H).to(torch.bfloat16).cuda().contiguous()   # value = column index within the head
vis = torch.zeros(T, S, dtype=torch.bool); vis[:, :S] = True
bits = vis_bits_from_bool(vis, max_slots).cuda()
out = torch.zeros(T, H, dtype=torch.bfloat16, device="cuda")
_lib.check(lib.atspeed_tree_attention(q.data_ptr(), 3 * H, kc.data_ptr(), vc.data_ptr(), bits.data_ptr(), max_slots // 64, out.data_ptr(), T, S, heads, dh, _lib.ATSPEED_BF16, st))
torch.cuda.synchronize()
print("row 0 head 0 (expect 0..127):", out[0, :dh].float().cpu().numpy().round(1).tolist())
vc2 = key.expand(max_slots, H).to(torch.bfloat16).cuda().contiguous()          # value = key index
vis2 = torch.zeros(T, S, dtype=torch.bool)
for t in range(T): vis2[t, t * 9] = True                                        # one visible key per row: output = that key index
bits2 = vis_bits_from_bool(vis2, max_slots).cuda()
_lib.check(lib.atspeed_tree_attention(q.data_ptr(), 3 * H, kc.data_ptr(), vc2.data_ptr(), bits2.data_ptr(), max_slots // 64, out.data_ptr(), T, S, heads, dh, _lib.ATSPEED_BF16, st))
torch.cuda.synchronize()
print("key picked per row (expect 0,9,18,...):", out[:, 0].float().cpu().numpy().tolist())
print("same, column 77:", out[:, 77].float().cpu().numpy().tolist())
