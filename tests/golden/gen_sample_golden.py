#!/usr/bin/env python3
"""Golden vectors of the reference's SAMPLING mode (generation_config.do_sample = True): runs the REAL
/root/reference/code/beamSD.py on CPU for a list of torch seeds and stores its outputs.

Harness-side only (reference files untouched): gen_golden.HFAdapter plus
  * `_get_logits_warper(generation_config)` -> `[TemperatureLogitsWarper(temperature)]` (what transformers 4.41 returned
    for the reference's config; 5.x folded warpers into `_get_logits_processor`), and
  * `_get_logits_processor` called with do_sample switched off, so that it returns only the prefix-constrained processor
    as in 4.41.
Usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_sample_golden.py   (writes tests/golden/bssd_sample_golden.json)
"""
import json
import os
import sys

import torch
from transformers import LogitsProcessorList, TemperatureLogitsWarper

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import gen_golden as G                      # noqa: E402  (imports the reference modules)
from cases import CASES, build_case_inputs  # noqa: E402

SAMPLE_CASES = [("k20_dk40_sigma01_s7", 1.0), ("k5_dk10_indep", 1.0), ("k20_dk40_sigma0", 0.7), ("k10_dk40_sigma01", 1.3)]
SEEDS = list(range(12))


def adapt(m, temperature):
    def glp(**kw):
        gc = kw["generation_config"]
        ds, gc.do_sample = gc.do_sample, False
        try:
            return m.hf._get_logits_processor(**kw)
        finally:
            gc.do_sample = ds
    m.generation_config.do_sample = True
    m.generation_config.temperature = temperature
    m._get_logits_warper = lambda gc: LogitsProcessorList([TemperatureLogitsWarper(float(gc.temperature))])
    m._get_logits_processor = glp
    return m


def main():
    out = []
    for name, temp in SAMPLE_CASES:
        case = next(c for c in CASES if c["name"] == name)
        ci = build_case_inputs(case)
        target = adapt(G.HFAdapter(ci["target_dims"], ci["target_sd"], case["K"]), temp)
        draft = adapt(G.HFAdapter(ci["draft_dims"], ci["draft_sd"], case["DK"]), temp)
        inputs = {"input_ids": torch.from_numpy(ci["prompt"])[None, :]}
        P = len(ci["prompt"])
        runs = []
        for seed in SEEDS:
            rounds = []
            orig = G.ref_beamsd.verify

            def vw(*a, **k):
                o = orig(*a, **k)
                rounds.append(int(o["n_matches"]))
                return o
            G.ref_beamsd.verify = vw
            try:
                torch.manual_seed(seed)
                o = G.ref_beamsd.BSSD(target, draft, inputs, case["gamma"], case["max_new_tokens"], prefix_allowed_tokens_fn=ci["fn"])
                torch.manual_seed(seed)
                tg = G.ref_beamsd.target_generate(target, inputs, case["max_new_tokens"], prefix_allowed_tokens_fn=ci["fn"])
                runs.append({"seed": seed, "tokens": o["beam_sequence"][:, P:].tolist(), "scores": [float(x) for x in o["beam_scores"]],
                             "n_run": int(o["n_run"]), "n_matches": rounds, "tg_tokens": tg["beam_sequence"][:, P:].tolist(),
                             "tg_scores": [float(x) for x in tg["beam_scores"]]})
            except Exception as e:          # the reference's own defects under sampling are recorded, not hidden
                runs.append({"seed": seed, "reference_error": f"{type(e).__name__}: {e}"[:200]})
            finally:
                G.ref_beamsd.verify = orig
        ok = [r for r in runs if "reference_error" not in r]
        print(f"{name:24s} T={temp}: {len(ok)}/{len(runs)} seeds ran; mean accept/run "
              f"{sum(sum(r['n_matches']) for r in ok) / max(1, sum(r['n_run'] for r in ok)):.3f}")
        out.append({"name": name, "temperature": temp, "runs": runs})
    with open(os.path.join(HERE, "bssd_sample_golden.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
