#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING AND RUNNING the reference
(`/root/reference/code/{beamSD,generation_trie}.py`) in this container.  The reference
cannot travel to the GPU box, so only its INPUT RECIPES (seeds / dims) and OUTPUTS are
stored; weights and prompts are re-derived from `atspeed_amd.synth`'s hash PRNG.

Run (CPU, ~1 min):  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

Harness notes (harness-side only, the reference files are untouched):
  * `torch.cuda.synchronize` is a no-op (the reference Timer syncs unconditionally,
    beamSD.py:24,33) — no GPU here;
  * transformers 5.15 wants a Cache object, the reference passes/slices tuples
    (beamSD.py:102,418-429) -> `HFAdapter` converts both ways;
  * per-round traces are captured by re-binding `beamSD.verify` / `draft_beam_search`,
    which BSSD resolves at call time (beamSD.py:511-515).
"""
from __future__ import annotations

import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/code")
sys.dont_write_bytecode = True

torch.cuda.synchronize = lambda *a, **k: None  # noqa: E731  (no GPU; Timer syncs unconditionally)

import beamSD as ref_beamsd            # noqa: E402  the reference
import generation_trie as ref_trie     # noqa: E402  the reference

from transformers import LlamaConfig, LlamaForCausalLM   # noqa: E402
from transformers.cache_utils import DynamicCache        # noqa: E402

from atspeed_amd import synth                            # noqa: E402
from tests.golden.cases import CASES, HANDOVER, TRIE_CASES, build_case_inputs, chain_sequences   # noqa: E402


class HFAdapter:
    """Makes an HF-5.x Llama look like the 4.41 model object the reference drives."""

    def __init__(self, dims: synth.LlamaDims, state_dict, num_beams: int):
        cfg = LlamaConfig(vocab_size=dims.vocab_size, hidden_size=dims.hidden, intermediate_size=dims.ffn,
                          num_hidden_layers=dims.n_layers, num_attention_heads=dims.n_heads,
                          num_key_value_heads=dims.n_heads, rms_norm_eps=dims.rms_eps,
                          rope_theta=dims.rope_theta, max_position_embeddings=2048,
                          attn_implementation="eager", tie_word_embeddings=False,
                          bos_token_id=1, eos_token_id=2, pad_token_id=0)
        self.hf = LlamaForCausalLM(cfg).to(torch.float32).eval()
        missing = self.hf.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state_dict.items()},
                                          strict=False)
        assert not [m for m in missing.missing_keys if "rotary" not in m], missing
        self.generation_config = self.hf.generation_config
        self.generation_config.num_beams = num_beams
        self.generation_config.do_sample = False
        self.dtype = torch.float32
        self.device = torch.device("cpu")
        self.n_layers = dims.n_layers

    def _get_logits_processor(self, **kw):
        return self.hf._get_logits_processor(**kw)

    def __call__(self, input_ids, attention_mask, position_ids, past_key_values):
        cache = DynamicCache(config=self.hf.config)
        if past_key_values is not None:
            for l, (k, v) in enumerate(past_key_values):
                cache.update(k, v, l)
        out = self.hf(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                      past_key_values=cache, use_cache=True)
        kv = tuple((out.past_key_values.layers[l].keys, out.past_key_values.layers[l].values)
                   for l in range(self.n_layers))
        return SimpleNamespace(logits=out.logits.to(torch.float32), past_key_values=kv)


def run_case(case) -> dict:
    ci = build_case_inputs(case)
    target = HFAdapter(ci["target_dims"], ci["target_sd"], case["K"])
    draft = HFAdapter(ci["draft_dims"], ci["draft_sd"], case["DK"])
    fn = ci["fn"]
    if case["mask"] == "chain":          # the reference's OWN chained Trie and mask factory (generation_trie.py:19-21,92-98)
        a, b = chain_sequences(ci["prompt"], ci["items"])
        ta = ref_trie.Trie(a)
        ta.append(ref_trie.Trie(b), HANDOVER)
        fn = ref_trie.prefix_allowed_tokens_fn(ta)
    from transformers import LogitsProcessorList
    procs = LogitsProcessorList(ci["procs"]) if ci["procs"] else None
    inputs = {"input_ids": torch.from_numpy(ci["prompt"])[None, :]}

    rounds = []
    orig_verify, orig_draft = ref_beamsd.verify, ref_beamsd.draft_beam_search

    def draft_wrap(*a, **k):
        o = orig_draft(*a, **k)
        rounds.append({"step_len": list(o["step_len"]),
                       "draft_ids": [x.tolist() for x in o["step_seq_tokens"]]})
        rounds[-1]["draft_len"] = len(o["step_beam_indices"])
        return o

    def verify_wrap(*a, **k):
        o = orig_verify(*a, **k)
        rounds[-1]["n_matches"] = int(o["n_matches"])
        rounds[-1]["beam_scores"] = o["beam_scores"].tolist()
        return o

    ref_beamsd.verify, ref_beamsd.draft_beam_search = verify_wrap, draft_wrap
    ref_error = None
    try:
        out = ref_beamsd.BSSD(target, draft, inputs, case["gamma"], case["max_new_tokens"], logits_processor=procs,
                              prefix_allowed_tokens_fn=fn)
    except RuntimeError as e:
        # known reference defect (DESIGN.md "reference quirks"): after a NON-first round with
        # n_matches == draft_len - 1 the draft cache is K entries shorter than the mask the
        # reference builds for it (beamSD.py:387-392,424-429); HF 5.x refuses the shape.
        ref_error = str(e)
    finally:
        ref_beamsd.verify, ref_beamsd.draft_beam_search = orig_verify, orig_draft
    tg = ref_beamsd.target_generate(target, inputs, case["max_new_tokens"], logits_processor=procs, prefix_allowed_tokens_fn=fn)

    P = len(ci["prompt"])
    if ref_error is not None:
        return {"name": case["name"], "reference_error": ref_error, "rounds_before_error": rounds,
                "tg_tokens": tg["beam_sequence"][:, P:].tolist(),
                "tg_scores": [float(x) for x in tg["beam_scores"].tolist()], "min_gap_final": 1.0}
    res = {
        "name": case["name"],
        "bssd_tokens": out["beam_sequence"][:, P:].tolist(),
        "bssd_scores": [float(x) for x in out["beam_scores"].tolist()],
        "n_run": int(out["n_run"]),
        "total_accept_steps": int(out["total_accept_steps"]),
        "total_accept_tokens": int(out["total_accept_tokens"]),
        "ave_accept_tokens": float(out["ave_accept_tokens"]),
        "rounds": rounds,
        "tg_tokens": tg["beam_sequence"][:, P:].tolist(),
        "tg_scores": [float(x) for x in tg["beam_scores"].tolist()],
        "prompt_echo_ok": bool((out["beam_sequence"][:, :P] == torch.from_numpy(ci["prompt"])[None]).all()),
    }
    # score gaps: fixtures must not sit on near-ties, the build's tie-break is its own
    s = np.asarray(res["tg_scores"], dtype=np.float64)
    res["min_gap_final"] = float(np.min(np.abs(np.diff(s)))) if len(s) > 1 else 1.0
    # last-position logits of the target on the bare prompt (fp32 logits pin, 1e-3 tolerance)
    n = P
    mask = (torch.tril(torch.ones(n, n)) == 0) * torch.finfo(torch.float32).min
    with torch.no_grad():
        lo = target(torch.from_numpy(ci["prompt"])[None], mask[None, None], torch.arange(n)[None], None).logits[0, -1]
    res["prompt_last_logits_sample"] = [float(x) for x in lo[31990:32010].tolist()]
    res["prompt_last_lse"] = float(torch.logsumexp(lo.detach(), -1))
    return res


def run_trie_case(tc) -> dict:
    seqs = tc["sequences"]
    t = ref_trie.Trie([list(s) for s in seqs])
    res = {"name": tc["name"], "len": len(t), "iter": [list(x) for x in t],
           "gets": [[list(q), t.get(list(q))] for q in tc["queries"]]}
    if tc.get("append"):
        t2 = ref_trie.Trie([list(s) for s in tc["append"]["sequences"]])
        t.append(t2, tc["append"]["bos"])
        res["gets_appended"] = [[list(q), t.get(list(q))] for q in tc["queries"] + tc["append"]["queries"]]
    t3 = ref_trie.Trie.load_from_dict(t.trie_dict)
    res["loaded_len"] = len(t3)
    fn = ref_trie.prefix_allowed_tokens_fn(ref_trie.Trie([list(s) for s in seqs]))
    res["fn"] = [[list(q), fn(0, torch.tensor(list(q), dtype=torch.long))] for q in tc["queries"]]
    return res


def main():
    outs = []
    only = set(sys.argv[1:])                 # `gen_golden.py name ...`: (re)generate these cases only and merge them into the stored file
    if only:
        with open(os.path.join(HERE, "bssd_golden.json")) as f:
            old = {c["name"]: c for c in json.load(f)}
        for case in CASES:
            if case["name"] in only:
                old[case["name"]] = run_case(case)
                r = old[case["name"]]
                print(f"{case['name']:28s} " + (f"n_run={r['n_run']} accept={r['total_accept_steps']} rounds={[x['n_matches'] for x in r['rounds']]} "
                                                  f"bssd==tg:{r['bssd_tokens'] == r['tg_tokens']} min_gap={r['min_gap_final']:.2e}"
                                                  if "reference_error" not in r else "REFERENCE RAISED " + r["reference_error"][:90]))
        with open(os.path.join(HERE, "bssd_golden.json"), "w") as f:
            json.dump([old[c["name"]] for c in CASES if c["name"] in old], f)
        return
    for case in CASES:
        r = run_case(case)
        if "reference_error" in r:
            print(f"{case['name']:28s} REFERENCE RAISED: {r['reference_error'][:90]}")
            outs.append(r)
            continue
        print(f"{case['name']:28s} n_run={r['n_run']} accept={r['total_accept_steps']} "
              f"rounds={[x['n_matches'] for x in r['rounds']]} bssd==tg:{r['bssd_tokens'] == r['tg_tokens']} "
              f"min_gap={r['min_gap_final']:.2e}")
        outs.append(r)
    with open(os.path.join(HERE, "bssd_golden.json"), "w") as f:
        json.dump(outs, f)
    touts = [run_trie_case(tc) for tc in TRIE_CASES]
    with open(os.path.join(HERE, "trie_golden.json"), "w") as f:
        json.dump(touts, f)
    print("wrote", len(outs), "BSSD cases and", len(touts), "trie cases")


if __name__ == "__main__":
    main()
