#!/usr/bin/env python3
"""Where does the wall time of a lock-step batch go: device stages vs host work around the C call."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import synth
from atspeed_amd.beamSD import BSSD_batch
from atspeed_amd.generation_trie import PositionSetConstraint
from atspeed_amd.model import HipLlama
dev = torch.device("cuda", 0)
V = synth.BEAUTY.vocab_size
kw = dict(max_slots=512, max_tokens=512, max_logit_rows=384, device=dev)
t = HipLlama.from_synthetic(synth.llama_7b(V, 32), 2025, dtype=torch.bfloat16, num_beams=20, **kw)
d = HipLlama.from_synthetic(synth.llama_68m(V), 2026, dtype=torch.bfloat16, num_beams=40, **kw)
fn = PositionSetConstraint(synth.BEAUTY.allowed_tokens(), synth.RESPONSE_SEP)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
plens = synth.prompt_lengths(n, 2025)
prompts = [{"input_ids": torch.from_numpy(synth.synthetic_prompt(int(plens[u]), synth.tensor_seed(2025, f"user{u}")))[None].to(dev)} for u in range(n)]
BSSD_batch(t, d, prompts, 4, 4, prefix_allowed_tokens_fn=fn)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    o = BSSD_batch(t, d, prompts, 4, 4, prefix_allowed_tokens_fn=fn)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = sum(x["draft_time_cost"] + x["target_time_cost"] + x["verify_time_cost"] for x in o) * 1e3
    print(f"wall {wall * 1e3:7.1f} ms  device stages {dev_ms:7.1f} ms  host/other {wall * 1e3 - dev_ms:6.1f} ms")
pr = cProfile.Profile(); pr.enable()
BSSD_batch(t, d, prompts, 4, 4, prefix_allowed_tokens_fn=fn); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
