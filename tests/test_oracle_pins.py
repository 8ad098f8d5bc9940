"""The oracle (oracle/*.py) is only trustworthy because these tests pin it to outputs of
the REAL reference (tests/golden/*.json, produced by tests/golden/gen_golden.py importing
/root/reference/code) and to the installed HF Llama.  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import beamsd_ref as R
from oracle.llama_ref import RefLlama
from oracle.trie_ref import RefTrie, ref_position_set_fn, ref_suffix_trie_fn, ref_whole_sentence_fn
from tests.golden.cases import CASES, TRIE_CASES, build_case_inputs

SCORE_TOL = 1e-3   # north star: fp32 scores/logits within 1e-3; token ids bit-exact


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_bssd_equals_reference(case, bssd_golden):
    gold = bssd_golden[case["name"]]
    ci = build_case_inputs(case)
    tgt = RefLlama(ci["target_dims"], ci["target_sd"])
    drf = RefLlama(ci["draft_dims"], ci["draft_sd"])
    P = len(ci["prompt"])
    out = R.BSSD(tgt, drf, ci["prompt"], case["gamma"], case["max_new_tokens"], case["K"], case["DK"], ci["fn"], procs=ci["procs"])
    tg = R.target_generate(tgt, ci["prompt"], case["max_new_tokens"], case["K"], ci["fn"], procs=ci["procs"])
    assert tg["beam_sequence"][:, P:].tolist() == gold["tg_tokens"]
    np.testing.assert_allclose(tg["beam_scores"].numpy(), gold["tg_scores"], atol=SCORE_TOL, rtol=0)
    # losslessness of the greedy branch: identical to plain beam search
    assert out["beam_sequence"][:, P:].tolist() == tg["beam_sequence"][:, P:].tolist()
    if "reference_error" in gold:
        # the reference itself fails here (known defect, see gen_golden.py); only the
        # target_generate outputs and the lossless property pin this case
        return
    assert gold["prompt_echo_ok"]
    assert out["beam_sequence"][:, P:].tolist() == gold["bssd_tokens"]
    np.testing.assert_allclose(out["beam_scores"].numpy(), gold["bssd_scores"], atol=SCORE_TOL, rtol=0)
    assert out["n_run"] == gold["n_run"]
    assert out["total_accept_steps"] == gold["total_accept_steps"]
    assert out["total_accept_tokens"] == gold["total_accept_tokens"]
    assert out["ave_accept_tokens"] == pytest.approx(gold["ave_accept_tokens"])
    assert [r["n_matches"] for r in out["rounds"]] == [r["n_matches"] for r in gold["rounds"]]
    assert [r["step_len"] for r in out["rounds"]] == [r["step_len"] for r in gold["rounds"]]
    assert [r["draft_ids"] for r in out["rounds"]] == [r["draft_ids"] for r in gold["rounds"]]


def test_oracle_logits_equal_reference_model(bssd_golden):
    case = CASES[0]
    gold = bssd_golden[case["name"]]
    ci = build_case_inputs(case)
    m = RefLlama(ci["target_dims"], ci["target_sd"])
    inp = R._causal_inputs(torch.from_numpy(ci["prompt"]))
    lo = m.forward(inp.ids, inp.pos, inp.slots, inp.vis, n_logit_rows=1)[0]
    np.testing.assert_allclose(lo[31990:32010].numpy(), gold["prompt_last_logits_sample"], atol=1e-4, rtol=0)
    assert float(torch.logsumexp(lo, -1)) == pytest.approx(gold["prompt_last_lse"], abs=1e-4)


def test_oracle_llama_equals_hf_with_tree_mask():
    """Third-party arithmetic (transformers Llama) restated in oracle/llama_ref.py: check it
    against the installed HF model on a tree-shaped mask with a KV cache."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from transformers.cache_utils import DynamicCache
    from atspeed_amd import synth
    dims = synth.LlamaDims(300, 64, 2, 4, 160)
    sd = synth.synthetic_state_dict(dims, 5, std=0.08)
    cfg = LlamaConfig(vocab_size=300, hidden_size=64, intermediate_size=160, num_hidden_layers=2,
                      num_attention_heads=4, num_key_value_heads=4, rms_norm_eps=dims.rms_eps, rope_theta=dims.rope_theta,
                      attn_implementation="eager", tie_word_embeddings=False)
    hf = LlamaForCausalLM(cfg).float().eval()
    hf.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    ref = RefLlama(dims, sd)
    g = torch.Generator().manual_seed(0)
    P, B = 9, 5
    ids0 = torch.randint(0, 300, (P,), generator=g)
    ids1 = torch.randint(0, 300, (B,), generator=g)
    neg = torch.finfo(torch.float32).min
    with torch.no_grad():
        cache = DynamicCache(config=cfg)
        m0 = (torch.tril(torch.ones(P, P)) == 0) * neg
        o0 = hf(input_ids=ids0[None], attention_mask=m0[None, None], position_ids=torch.arange(P)[None],
                past_key_values=cache, use_cache=True)
        vis1 = torch.cat((torch.ones(B, P, dtype=torch.bool), torch.eye(B, dtype=torch.bool)), 1)
        vis1[:, 3] = False      # hide one prompt slot too
        m1 = (~vis1) * neg
        o1 = hf(input_ids=ids1[None], attention_mask=m1[None, None].float(), position_ids=torch.full((1, B), P),
                past_key_values=o0.past_key_values, use_cache=True)
    l0 = ref.forward(ids0, torch.arange(P), torch.arange(P), torch.tril(torch.ones(P, P, dtype=torch.bool)))
    l1 = ref.forward(ids1, torch.full((B,), P), torch.arange(P, P + B), vis1)
    np.testing.assert_allclose(l0.numpy(), o0.logits[0].numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(l1.numpy(), o1.logits[0].numpy(), atol=2e-5, rtol=0)


@pytest.mark.parametrize("tc", TRIE_CASES, ids=[t["name"] for t in TRIE_CASES])
def test_oracle_trie_equals_reference(tc, trie_golden):
    gold = trie_golden[tc["name"]]
    t = RefTrie(tc["sequences"])
    assert len(t) == gold["len"]
    assert [list(x) for x in t] == gold["iter"]
    for q, exp in gold["gets"]:
        assert t.get(q) == exp and t[q] == exp
    fn = ref_whole_sentence_fn(RefTrie(tc["sequences"]))
    for q, exp in gold["fn"]:
        assert fn(0, torch.tensor(q, dtype=torch.long)) == exp
    if tc.get("append"):
        t.append(RefTrie(tc["append"]["sequences"]), tc["append"]["bos"])
        for q, exp in gold["gets_appended"]:
            assert t.get(q) == exp
    assert len(RefTrie.load_from_dict(t.trie_dict)) == gold["loaded_len"]


def test_oracle_mask_fns():
    fn = ref_position_set_fn({0: [5, 6], 1: [7], 2: [2]}, [90, 91])
    assert fn(0, torch.tensor([1, 4, 90, 91])) == [5, 6]
    assert fn(0, torch.tensor([1, 90, 91, 4, 90, 91, 5])) == [7]          # last separator counts
    assert fn(0, torch.tensor([1, 4])) is None
    tr = RefTrie([[1, 5, 7, 2], [1, 6, 7, 2]])
    sfn = ref_suffix_trie_fn(tr, [90, 91], 1)
    assert sorted(sfn(0, torch.tensor([3, 90, 91]))) == [5, 6]
    assert sfn(0, torch.tensor([3, 90, 91, 5])) == [7]
    assert sfn(0, torch.tensor([3, 90, 91, 9])) == []


# ------------------------------------------------------------------ sampling branch (beamSD.py:293-321,332-369)
def _sample_models(name):
    import json, os
    from oracle.llama_ref import RefLlama
    from tests.golden.cases import CASES, build_case_inputs
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bssd_sample_golden.json")))
    g = next(x for x in gold if x["name"] == name)
    case = next(c for c in CASES if c["name"] == name)
    ci = build_case_inputs(case)
    return g, case, ci, RefLlama(ci["target_dims"], ci["target_sd"]), RefLlama(ci["draft_dims"], ci["draft_sd"])


@pytest.mark.parametrize("name", ["k20_dk40_sigma01_s7", "k5_dk10_indep", "k20_dk40_sigma0", "k10_dk40_sigma01"])
def test_sampling_oracle_reproduces_the_reference_seed_for_seed(name):
    """With torch's generator and the reference's call order the restatement returns the real reference's sampled beams,
    per-round n_matches and scores for every recorded seed (tests/golden/gen_sample_golden.py ran the reference)."""
    from oracle import beamsd_sample_ref as S
    g, case, ci, rt, rd = _sample_models(name)
    P = len(ci["prompt"])
    for r in g["runs"]:
        assert "reference_error" not in r
        torch.manual_seed(r["seed"])
        o = S.BSSD_sample(rt, rd, ci["prompt"], case["gamma"], case["max_new_tokens"], case["K"], case["DK"], ci["fn"], g["temperature"], S.TorchRng())
        assert o["beam_sequence"][:, P:].tolist() == r["tokens"], (name, r["seed"])
        assert [x["n_matches"] for x in o["rounds"]] == r["n_matches"] and o["n_run"] == r["n_run"]
        np.testing.assert_allclose(o["beam_scores"].numpy(), np.array(r["scores"], dtype=np.float32), atol=1e-3, rtol=0)
        torch.manual_seed(r["seed"])
        tg = S.target_generate_sample(rt, ci["prompt"], case["max_new_tokens"], case["K"], ci["fn"], g["temperature"], S.TorchRng())
        assert tg["beam_sequence"][:, P:].tolist() == r["tg_tokens"]
        np.testing.assert_allclose(tg["beam_scores"].numpy(), np.array(r["tg_scores"], dtype=np.float32), atol=1e-3, rtol=0)


def test_hash_rng_sampling_has_the_reference_law():
    """The counter-based generator the device uses (Gumbel top-n draws, hashed uniforms / subsets) against torch's generator
    on the same restatement: accepted steps per verification and the marginal of the best beam's first code agree within
    sampling error per code (the parity definition of the sampling branch is statistical; this is its pin)."""
    from oracle import beamsd_sample_ref as S
    g, case, ci, rt, rd = _sample_models("k10_dk40_sigma01")
    P, N = len(ci["prompt"]), 120
    stats = {}
    for mode in ("torch", "hash"):
        acc, runs, first = [], [], []
        for seed in range(N):
            if mode == "torch":
                torch.manual_seed(1000 + seed)
                rng = S.TorchRng()
            else:
                rng = S.HashRng(5000 + seed)
            o = S.BSSD_sample(rt, rd, ci["prompt"], case["gamma"], case["max_new_tokens"], case["K"], case["DK"], ci["fn"], g["temperature"], rng)
            acc.append(o["total_accept_steps"]); runs.append(o["n_run"])
            first.append(np.bincount(np.unique(o["beam_sequence"][:, P].numpy()) - 32000, minlength=64) > 0)
        stats[mode] = (np.array(acc, dtype=np.float64), np.array(runs, dtype=np.float64), np.array(first).mean(0))
    (a0, r0, h0), (a1, r1, h1) = stats["torch"], stats["hash"]
    se = np.sqrt(a0.var() / N + a1.var() / N)
    assert abs(a0.mean() - a1.mean()) < 4 * se + 1e-9, (a0.mean(), a1.mean(), se)
    se_r = np.sqrt(r0.var() / N + r1.var() / N)
    assert abs(r0.mean() - r1.mean()) < 4 * se_r + 1e-9
    # per code: the fraction of runs whose beams contain it as first code (one Bernoulli per run, so runs are independent)
    pm = (h0 + h1) / 2
    m = (pm > 0.05) & (pm < 0.95)
    z = np.abs(h0 - h1)[m] / np.sqrt(pm[m] * (1 - pm[m]) * 2 / N)
    assert m.sum() >= 5 and z.max() < 4.0, (z.max(), int(m.sum()))


def test_fp64_arbiter_mode_of_the_oracle_agrees_with_the_fp32_oracle():
    """RefLlama(dtype=float64) + beamsd_ref.SCORE_DTYPE = float64 (the arbiter tests/test_fulldims_gpu.py judges near-tied full-dims decisions with):
    the same search in double precision gives the golden case's items, rounds and -- within fp32 re-association noise -- scores."""
    import torch
    from oracle import beamsd_ref as R
    from oracle.llama_ref import RefLlama
    from tests.golden.cases import CASES, build_case_inputs
    case = next(c for c in CASES if c["name"] == "k20_dk40_sigma01_s7")
    ci = build_case_inputs(case)
    args = (ci["prompt"], case["gamma"], case["max_new_tokens"], case["K"], case["DK"], ci["fn"])
    a = R.BSSD(RefLlama(ci["target_dims"], ci["target_sd"]), RefLlama(ci["draft_dims"], ci["draft_sd"]), *args)
    R.SCORE_DTYPE = torch.float64
    try:
        b = R.BSSD(RefLlama(ci["target_dims"], ci["target_sd"], dtype=torch.float64), RefLlama(ci["draft_dims"], ci["draft_sd"], dtype=torch.float64), *args)
    finally:
        R.SCORE_DTYPE = torch.float32
    assert b["beam_scores"].dtype == torch.float64
    assert a["beam_sequence"].tolist() == b["beam_sequence"].tolist() and a["n_run"] == b["n_run"]
    assert [r["n_matches"] for r in a["rounds"]] == [r["n_matches"] for r in b["rounds"]]
    assert float((a["beam_scores"].double() - b["beam_scores"]).abs().max()) < 1e-4


def test_fp64_arbiter_helpers_judge_a_list_against_the_fp64_search():
    """tests/arbiter.py (what the K = 20 full-dims GPU test calls when the engine and the fp32 oracle disagree): the fp64 plain beam search gives the
    golden items; a list that equals it has gap 0 without a forward; a list with two ranks swapped has exactly the fp64 score difference of the two
    items as its gap; fp32 scoring of the same sequences agrees with the fp64 scoring to re-association noise."""
    import torch
    from oracle.llama_ref import RefLlama
    from tests.arbiter import fp64_gap, fp64_truth, oracle_scores_of
    from tests.golden.cases import CASES, build_case_inputs
    case = next(c for c in CASES if c["name"] == "k20_dk40_sigma01_s7")
    ci = build_case_inputs(case)
    rt = RefLlama(ci["target_dims"], ci["target_sd"])
    t_items, t_sc = fp64_truth(rt, ci["prompt"], case["K"], ci["fn"])
    gold = next(g for g in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bssd_golden.json"))) if g["name"] == case["name"])
    assert t_items == gold["tg_tokens"]                                   # the real reference's plain beam search
    assert fp64_gap(rt, ci["prompt"], t_items, t_items, t_sc) == 0.0
    swapped = [list(x) for x in t_items]
    swapped[3], swapped[4] = swapped[4], swapped[3]
    gap = fp64_gap(rt, ci["prompt"], swapped, t_items, t_sc)
    assert abs(gap - abs(t_sc[3] - t_sc[4])) < 1e-9 and gap > 0
    s32 = oracle_scores_of(rt, ci["prompt"], t_items)
    assert max(abs(a - b) for a, b in zip(s32, t_sc)) < 1e-4
