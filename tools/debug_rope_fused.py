#!/usr/bin/env python3
"""one-off: fused vs separate RoPE pass of the fp8 / bf16 engine against each other and against the bf16 engine (which one is off?)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import synth
from atspeed_amd.model import HipLlama, vis_bits_from_bool
n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 56
V = 32000 + 256
dims = synth.LlamaDims(V, 1024, 2, 8, 2816)
g = torch.Generator().manual_seed(11)
seqs = []
for i in range(n_seq):
    T = 100 if i < 2 else int(torch.randint(61, 100, (1,), generator=g))
    ids = torch.randint(3, V, (T,), generator=g).to(torch.int32)
    vis = torch.tril(torch.ones(T, T, dtype=torch.bool))
    ar = torch.arange(T, dtype=torch.int32)
    seqs.append((ids, ar, ar.clone(), vis_bits_from_bool(vis, 256), T, 3))
def run(fp8, fuse):
    os.environ["ATSPEED_FUSE_QKV_ROPE"] = fuse
    m = HipLlama.from_synthetic(dims, 91, dtype=torch.bfloat16, max_slots=256, max_tokens=256, max_logit_rows=256)
    if fp8: m.enable_fp8()
    out = [o.clone() for o in m.forward_raw_batch(seqs)]
    torch.cuda.synchronize()
    return torch.stack([o.float() for o in out])
ref = run(False, "0")
for fp8 in (False, True):
    for fuse in ("0", "1"):
        o = run(fp8, fuse)
        d = (o - ref).abs()
        per_seq = d.amax(dim=(1, 2))
        print(f"fp8={fp8} fuse={fuse}: max |diff to bf16 unfused| {float(d.max()):.4f} mean {float(d.mean()):.5f}; worst sequences {per_seq.topk(5).indices.tolist()} {[round(float(x), 3) for x in per_seq.topk(5).values]}")
