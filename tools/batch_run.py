#!/usr/bin/env python3
"""Workload for a kernel trace of the SMALL lock-step batches (1 / 4 / 16 users per batch: the band where speculation pays): `reps` batches of
`users` users through BSSD_batch (mode bssd) or target_generate_batch (mode tg), full Llama-7B / Llama-68M dims, Beauty, K=20 / DK=40.
usage: rocprofv3 --kernel-trace --stats -d gpurun_out/band -o u4 -- python3 tools/batch_run.py 4 bssd [reps] [resid_scale | none] [fp8]; then tools/trace_gaps.py
Prints one MARK line: ms until the last user of a batch has its result, ms per user, items/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atspeed_amd import synth
from atspeed_amd.beamSD import BSSD, BSSD_batch, target_generate, target_generate_batch
from atspeed_amd.generation_trie import PositionSetConstraint
from atspeed_amd.model import HipLlama
users = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mode = sys.argv[2] if len(sys.argv) > 2 else "bssd"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rs = float(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] != "none" else None
fp8 = len(sys.argv) > 5 and sys.argv[5] == "fp8"
dev = torch.device("cuda", 0)
V = synth.BEAUTY.vocab_size
kw = dict(max_slots=512, max_tokens=512, max_logit_rows=384, device=dev)
r = 1.0 if rs is None else rs
d = HipLlama.from_synthetic(synth.llama_68m(V), 2026, dtype=torch.bfloat16, num_beams=40, resid_scale=r, **kw)
t = HipLlama.from_synthetic(synth.llama_7b(V, 32), 2025, dtype=torch.bfloat16, num_beams=20, resid_scale=r, align_to=(d if rs is not None else None), **kw)
if fp8:
    t.enable_fp8()          # BASELINE config 5: W8A8 target projections (weight-streaming form at 1 user, ring kernel in batches)
fn = PositionSetConstraint(synth.BEAUTY.allowed_tokens(), synth.RESPONSE_SEP)
plens = synth.prompt_lengths(users, 2025)
prompts = [{"input_ids": torch.from_numpy(synth.synthetic_prompt(int(plens[u]), synth.tensor_seed(2025, f"user{u}")))[None].to(dev)} for u in range(users)]
if mode == "bssd":
    call = (lambda: [BSSD(t, d, prompts[0], 4, 4, prefix_allowed_tokens_fn=fn)]) if users == 1 else (lambda: BSSD_batch(t, d, prompts, 4, 4, prefix_allowed_tokens_fn=fn))
else:
    call = (lambda: [target_generate(t, prompts[0], 4, prefix_allowed_tokens_fn=fn)]) if users == 1 else (lambda: target_generate_batch(t, prompts, 4, prefix_allowed_tokens_fn=fn))
call(); call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    outs = call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
acc = (sum(o["total_accept_steps"] for o in outs) / max(1, sum(o["n_run"] for o in outs))) if mode == "bssd" else 0.0
stage = ""
if mode == "bssd" and users == 1:      # the engine's own stage clocks (hipEvents; the CSV columns of inference.py:183-187) and forward counts of the last call
    o = outs[0]
    stage = (f"; stages draft {1e3 * o['draft_time_cost']:.2f} target {1e3 * o['target_time_cost']:.2f} verify {1e3 * o['verify_time_cost']:.2f} ms, "
             f"{o['n_target_forwards']} target / {o['n_draft_forwards']} draft forwards")
print(f"MARK {mode} users {users} resid_scale {rs}{' fp8' if fp8 else ''}: {dt * 1e3:.2f} ms to last result, {1e3 * dt / users:.2f} ms/user, {users * 20 / dt:.1f} items/s, accept {acc:.3f}{stage}", flush=True)
