#!/usr/bin/env python3
"""Workload for a kernel trace of the one-user-at-a-time loop (inference.py:162-176 as written): N users through BSSD, one call each.
usage: rocprofv3 --kernel-trace -d gpurun_out/su -o su -- python3 tools/single_user_run.py [users]; then tools/trace_gaps.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import synth
from atspeed_amd.beamSD import BSSD
from atspeed_amd.generation_trie import PositionSetConstraint
from atspeed_amd.model import HipLlama
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
V = synth.BEAUTY.vocab_size
kw = dict(max_slots=512, max_tokens=512, max_logit_rows=384, device=dev)
t = HipLlama.from_synthetic(synth.llama_7b(V, 32), 2025, dtype=torch.bfloat16, num_beams=20, **kw)
d = HipLlama.from_synthetic(synth.llama_68m(V), 2026, dtype=torch.bfloat16, num_beams=40, **kw)
fn = PositionSetConstraint(synth.BEAUTY.allowed_tokens(), synth.RESPONSE_SEP)
plens = synth.prompt_lengths(n + 2, 2025)
prompts = [{"input_ids": torch.from_numpy(synth.synthetic_prompt(int(plens[u]), synth.tensor_seed(2025, f"user{u}")))[None].to(dev)} for u in range(n + 2)]
for p in prompts[:2]: BSSD(t, d, p, 4, 4, prefix_allowed_tokens_fn=fn)
torch.cuda.synchronize()
t0 = time.perf_counter()
for p in prompts[2:]: BSSD(t, d, p, 4, 4, prefix_allowed_tokens_fn=fn)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"MARK {n} users {dt * 1e3:.1f} ms  {1e3 * dt / n:.2f} ms/user  {n * 20 / dt:.1f} items/s", flush=True)
