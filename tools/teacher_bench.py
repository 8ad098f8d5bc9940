#!/usr/bin/env python3
"""Teacher-data generation rate at Llama-7B dims: K=20 beams of 5 tokens under the strict Beauty-shaped trie in
lock-step batches + one packed tree-mask forward per sample for the label's and the beams' logits."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atspeed_amd import synth
from atspeed_amd.generation_trie import SuffixTrieConstraint, Trie
from atspeed_amd.model import HipLlama
from atspeed_amd.teacher import generate_teacher_data
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
vocab = synth.BEAUTY
V = vocab.vocab_size
m = HipLlama.from_synthetic(synth.llama_7b(V, 32), 2025, dtype=torch.bfloat16, num_beams=20, max_slots=512, max_tokens=512, max_logit_rows=512, device=dev)
items = synth.synthetic_items(vocab)
fn = SuffixTrieConstraint(Trie([[1] + [int(t) for t in it] + [2] for it in items]), synth.RESPONSE_SEP, 1)
plens = synth.prompt_lengths(n, 7)
prompts = [synth.synthetic_prompt(int(plens[u]), 900 + u) for u in range(n)]
labels = [[int(t) for t in items[(37 * u) % len(items)]] + [2] for u in range(n)]
generate_teacher_data(m, prompts[:16], labels[:16], fn, beam_size=20, max_new_token=5, users_per_batch=16)
torch.cuda.synchronize()
for upb in (64, 128):
    t0 = time.perf_counter()
    out = generate_teacher_data(m, prompts, labels, fn, beam_size=20, max_new_token=5, users_per_batch=upb)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n} samples, {upb} per lock-step batch: {dt:.2f} s  {n / dt:.1f} samples/s  (teacher_output {tuple(out['teacher_output'][0].shape)}, logits {tuple(out['teacher_output_logits'][0].shape)})", flush=True)

# where the time goes (synchronised brackets around the three stages; slower than the pipelined run above)
import atspeed_amd.teacher as TT
acc = {"search": 0.0, "pack+forward": 0.0}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[name] += time.perf_counter() - t
        return r
    return w
TT.target_generate_batch = timed("search", TT.target_generate_batch)
TT._score_rows = timed("pack+forward", TT._score_rows)
t0 = time.perf_counter()
generate_teacher_data(m, prompts, labels, fn, beam_size=20, max_new_token=5, users_per_batch=128)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"breakdown over {n} samples: total {tot:.2f} s; " + "; ".join(f"{k} {v:.2f} s" for k, v in acc.items()) +
      f"; gathers + downloads {tot - sum(acc.values()):.2f} s", flush=True)
