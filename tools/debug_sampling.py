#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atspeed_amd.beamSD import BSSD, last_trace
from atspeed_amd.model import HipLlama
from oracle import beamsd_sample_ref as S
from oracle.llama_ref import RefLlama
from tests.golden.cases import CASES, build_case_inputs
name, T, seed = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
case = next(c for c in CASES if c["name"] == name)
ci = build_case_inputs(case)
kw = dict(max_slots=512, max_tokens=512, max_logit_rows=448)
tgt = HipLlama.from_state_dict(ci["target_dims"], ci["target_sd"], torch.float32, num_beams=case["K"], **kw)
drf = HipLlama.from_state_dict(ci["draft_dims"], ci["draft_sd"], torch.float32, num_beams=case["DK"], **kw)
for m in (tgt, drf):
    m.generation_config.do_sample = True; m.generation_config.temperature = T
rt, rd = RefLlama(ci["target_dims"], ci["target_sd"]), RefLlama(ci["draft_dims"], ci["draft_sd"])
P = len(ci["prompt"]); V = ci["target_dims"].vocab_size
rng = S.HashRng(seed)
ref = S.BSSD_sample(rt, rd, ci["prompt"], case["gamma"], case["max_new_tokens"], case["K"], case["DK"], ci["fn"], T, rng)
out = BSSD(tgt, drf, {"input_ids": torch.from_numpy(ci["prompt"])[None].cuda()}, case["gamma"], case["max_new_tokens"], prefix_allowed_tokens_fn=ci["fn"], seed=seed)
tr = last_trace(tgt, drf)
print("oracle rounds", [(r["draft_len"], r["n_matches"], r["step_len"]) for r in ref["rounds"]], "margin", rng.min_margin)
print("device rounds", [(r["draft_len"], r["n_matches"], r["n_beams"]) for r in tr])
for ri, (a, b) in enumerate(zip(ref["rounds"], tr)):
    for si, (x, y) in enumerate(zip(a["draft_ids"], b["draft_ids"])):
        y2 = [v for v in y if v >= 0]
        print(" round", ri, "step", si, "draft ids equal:", x == y2, len(x), len(y2))
        if x != y2:
            print("   oracle", x[:10]); print("   device", y2[:10]); break
    print(" verify trace", [{k: (v if k != "ids" else v[:6]) for k, v in t.items()} for t in a["verify"]])
print("oracle tokens", ref["beam_sequence"][:4, P:].tolist())
print("device tokens", out["beam_sequence"][:4, P:].cpu().tolist())
a = ref["beam_sequence"][:, P:].tolist(); b = out["beam_sequence"][:, P:].cpu().tolist()
for i, (x, y) in enumerate(zip(a, b)):
    if x != y: print("row", i, "oracle", x, float(ref["beam_scores"][i]), "device", y, float(out["beam_scores"][i]))
print("set equal:", sorted(map(tuple, a)) == sorted(map(tuple, b)))
