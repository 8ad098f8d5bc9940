#!/usr/bin/env python3
"""Acceptance of the aligned synthetic draft/target pair as a function of resid_scale (tuning aid for bench.py's
high-acceptance bracket).  usage: python tools/accept_sweep.py [users]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atspeed_amd import synth
from atspeed_amd.beamSD import BSSD_batch
from atspeed_amd.generation_trie import PositionSetConstraint
from atspeed_amd.model import HipLlama

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda", 0)
V = synth.BEAUTY.vocab_size
fn = PositionSetConstraint(synth.BEAUTY.allowed_tokens(), synth.RESPONSE_SEP)
plens = synth.prompt_lengths(n, 2025)
prompts = [{"input_ids": torch.from_numpy(synth.synthetic_prompt(int(plens[u]), synth.tensor_seed(2025, f"user{u}")))[None].to(dev)} for u in range(n)]
kw = dict(max_slots=512, max_tokens=512, max_logit_rows=384, device=dev)
for rs in (1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 2e-4):
    d = HipLlama.from_synthetic(synth.llama_68m(V), 2026, dtype=torch.bfloat16, num_beams=40, resid_scale=rs, **kw)
    t = HipLlama.from_synthetic(synth.llama_7b(V, layers), 2025, dtype=torch.bfloat16, num_beams=20, resid_scale=rs, align_to=d, **kw)
    BSSD_batch(t, d, prompts, 4, 4, prefix_allowed_tokens_fn=fn)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    o = BSSD_batch(t, d, prompts, 4, 4, prefix_allowed_tokens_fn=fn)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    acc = sum(x["total_accept_steps"] for x in o); runs = sum(x["n_run"] for x in o)
    print(f"resid_scale {rs:8.1e}: accept/run {acc / runs:.3f}  n_run/user {runs / n:.2f}  target fwd/user {sum(x['n_target_forwards'] for x in o) / n:.2f}  "
          f"{n * 20 / dt:8.1f} items/s", flush=True)
    del t, d
