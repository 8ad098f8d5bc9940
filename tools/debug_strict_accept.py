#!/usr/bin/env python3
"""Engine vs oracle acceptance under a small strict trie (beams shrink below K / DK)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atspeed_amd import synth
from atspeed_amd.beamSD import BSSD, last_trace
from atspeed_amd.harness import ItemIndex, SeqRecTestData, CodeTokenEncoder, encode_prompt
from atspeed_amd.model import HipLlama
from oracle import beamsd_ref as R
from oracle.llama_ref import RefLlama
rng = np.random.default_rng(3)
idx = {str(i): [f"<a_{rng.integers(6)}>", f"<b_{rng.integers(8)}>", f"<c_{rng.integers(8)}>", f"<d_{rng.integers(8)}>"] for i in range(300)}
ix = ItemIndex(idx)
train = {u: rng.integers(0, 300, size=rng.integers(1, 12)).tolist() for u in range(12)}
valid = {u: rng.integers(0, 300, size=1).tolist() for u in range(12)}
test = {u: (rng.integers(0, 300, size=1).tolist() if u % 4 else []) for u in range(12)}
data = SeqRecTestData(ix, train, valid, test)
V = ix.vocab_size
kw = dict(dtype=torch.float32, max_slots=512, max_tokens=512, max_logit_rows=448)
d = HipLlama.from_synthetic(synth.LlamaDims(V, 96, 2, 3, 256), 5, num_beams=20, resid_scale=1e-6, **kw)
t = HipLlama.from_synthetic(synth.LlamaDims(V, 128, 3, 4, 352), 6, num_beams=10, resid_scale=1e-6, align_to=d, **kw)
strict = data.strict_trie_fn()
enc = CodeTokenEncoder(ix)
rt = RefLlama(t.dims, t.export_state_dict(), max_slots=512)
rd = RefLlama(d.dims, d.export_state_dict(), max_slots=512)
for u in data.users[:3]:
    p = encode_prompt(data, u, None, enc)
    o = BSSD(t, d, {"input_ids": torch.from_numpy(p)[None].cuda()}, 4, 4, prefix_allowed_tokens_fn=strict)
    tr = last_trace(t, d)
    print("engine n_run", o["n_run"], "accept", o["total_accept_steps"], [(r["draft_len"], r["n_matches"], r["n_beams"]) for r in tr])
    try:
        ro = R.BSSD(rt, rd, p, 4, 4, 10, 20, strict)
    except ValueError as e:
        print("oracle:", e); continue
    print("engine n_run", o["n_run"], "accept", o["total_accept_steps"], [(r["draft_len"], r["n_matches"], r["n_beams"]) for r in tr])
    print("oracle n_run", ro["n_run"], "accept", ro["total_accept_steps"], [(r["draft_len"], r["n_matches"], r["step_len"]) for r in ro["rounds"]])
    print(" same items:", o["beam_sequence"][:, len(p):].cpu().tolist() == ro["beam_sequence"][:, len(p):].tolist())
    print(" engine draft ids r0:", [[x % V for x in s[:12]] for s in tr[0]["draft_ids"]])
