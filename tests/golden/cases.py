"""Fixture case table shared by the generator (gen_golden.py, runs the reference) and the
tests (which re-derive the same inputs from the hash PRNG and compare with the stored
reference outputs).  Inputs only — no reference code."""
from __future__ import annotations

import numpy as np

from atspeed_amd import synth
from atspeed_amd.generation_trie import PositionSetConstraint, SuffixTrieConstraint, Trie, prefix_allowed_tokens_fn

TINY_T = dict(hidden=128, n_layers=2, n_heads=4, ffn=352)
TINY_D = dict(hidden=96, n_layers=2, n_heads=3, ffn=256)      # draft with its own shape (68M-like: few wide heads)

_BASE = dict(P=24, gamma=4, max_new_tokens=4, K=20, DK=40, mask="pos", seed=2025, draft="perturb", sigma=0.1,
             head_std=0.25)


def _c(name, **kw):
    d = dict(_BASE)
    d.update(kw)
    d["name"] = name
    return d


CASES = [
    _c("k20_dk40_sigma0", sigma=0.0),                       # draft == target: every step accepted
    _c("k20_dk40_sigma002", sigma=0.02),
    _c("k20_dk40_sigma01", sigma=0.1),                      # mixed acceptance
    _c("k20_dk40_sigma01_s7", sigma=0.1, seed=7),
    _c("k20_dk40_sigma02_s11", sigma=0.2, seed=11),
    _c("k20_dk40_sigma03", sigma=0.3),                      # (almost) nothing accepted
    _c("k20_dk20_sigma002", sigma=0.02, DK=20),             # DK == K: needs exact top-K set equality
    _c("k10_dk40_sigma01", sigma=0.1, K=10),                # second shipped beam size (inference.sh:16)
    _c("k1_dk1_sigma0", sigma=0.0, K=1, DK=1),              # BASELINE config 0: greedy
    _c("k1_dk1_sigma03", sigma=0.3, K=1, DK=1),
    _c("k5_dk10_indep", draft="independent", K=5, DK=10),   # unrelated draft of a different shape
    _c("k20_dk40_indep_p40", draft="independent", P=40),
    _c("k20_dk40_gamma2", sigma=0.05, gamma=2),             # gamma < max_new_tokens - 1
    _c("k20_dk40_new5_gamma3", sigma=0.02, gamma=3, max_new_tokens=5, mask="pos7"),
    _c("k20_dk40_new7_gamma2", sigma=0.0, gamma=2, max_new_tokens=7, mask="pos7"),   # full accept -> draft re-ingests its last block (beamSD.py:402-416)
    _c("k6_dk12_new7_gamma2_s5", sigma=0.05, gamma=2, max_new_tokens=7, mask="pos7", K=6, DK=12, seed=5),
    _c("k6_dk12_new7_gamma3_s9", sigma=0.1, gamma=3, max_new_tokens=7, mask="pos7", K=6, DK=12, seed=9),
    _c("k8_dk16_trie", sigma=0.05, K=8, DK=16, mask="trie"),  # strict suffix trie (teacher-data style)
    _c("k20_dk40_trie", sigma=0.1, mask="trie"),
    # round 3: the optional arguments of BSSD (beamSD.py:460-481) and chained tries (generation_trie.py:19-21,55-57,67-68)
    _c("k5_dk10_nomask", sigma=0.02, K=5, DK=10, mask="none"),          # prefix_allowed_tokens_fn=None: every token a candidate, no id filter
    _c("k20_dk40_nomask", sigma=0.05, mask="none"),
    _c("k8_dk16_chain", sigma=0.05, K=8, DK=16, mask="chain"),           # whole-sentence trie A with an appended trie B taking over after 2 codes
    _c("k8_dk16_proc", sigma=0.05, K=8, DK=16, procs="bias"),            # mask + an extra logits processor
    _c("k5_dk10_proc_nomask", sigma=0.05, K=5, DK=10, mask="none", procs="bias+favor"),   # processors only: the id filter is on (beamSD.py:80)
]

HANDOVER = 0        # bos_token_id of the chained-trie case: a token no prompt and no item holds


class HashBias:
    """An extra logits processor for the fixtures: scores + b, b a fixed hash-PRNG vector over the vocabulary (any torch device)."""

    def __init__(self, vocab: int, seed: int, std: float = 1.5):
        import torch
        self.b = torch.from_numpy(synth.hash_normal(vocab, synth.tensor_seed(seed, "proc.bias"), std))

    def __call__(self, input_ids, scores):
        return scores + self.b.to(scores.device)


class Favor:
    """scores + boost for tokens >= lo: keeps an unmasked search inside the item codes, where the reference's id filter lets beams live."""

    def __init__(self, lo: int, boost: float):
        self.lo, self.boost = lo, boost

    def __call__(self, input_ids, scores):
        out = scores.clone()
        out[:, self.lo:] += self.boost
        return out


def build_processors(case, vocab_size: int):
    kind = case.get("procs")
    if not kind:
        return []
    procs = [HashBias(vocab_size, case["seed"])]
    if "favor" in kind:
        procs.append(Favor(synth.LLAMA_VOCAB, 40.0))
    return procs


def chain_sequences(prompt, items):
    """(sequences of trie A, sequences of trie B): A = prompt ++ first two codes ++ HANDOVER, B = last two codes ++ eos"""
    pre = [int(t) for t in prompt]
    a = [pre + [int(it[0]), int(it[1]), HANDOVER] for it in items]
    b = [[int(it[2]), int(it[3]), synth.EOS_ID] for it in items]
    return a, b



def build_case_inputs(case):
    vocab = synth.TINY
    V = vocab.vocab_size
    seed = case["seed"]
    tdims = synth.LlamaDims(V, **TINY_T)
    tsd = synth.synthetic_state_dict(tdims, seed, std=0.05, head_std=case["head_std"])
    if case["draft"] == "perturb":
        ddims = tdims
        dsd = synth.perturbed_state_dict(tsd, seed + 1, case["sigma"])
    else:
        ddims = synth.LlamaDims(V, **TINY_D)
        dsd = synth.synthetic_state_dict(ddims, seed + 1000, std=0.05, head_std=case["head_std"])
    prompt = synth.synthetic_prompt(case["P"], synth.tensor_seed(seed, "prompt"))
    items = synth.synthetic_items(vocab, seed)
    if case["mask"] == "pos":
        fn = PositionSetConstraint(vocab.allowed_tokens(), synth.RESPONSE_SEP)
    elif case["mask"] == "pos7":
        # 7 generated positions: the 4 code levels, then levels 0..2 again (exercises max_new_tokens > 4)
        al = vocab.allowed_tokens()
        for i in range(4, 7):
            al[i] = al[i - 4]
        al[7] = [synth.EOS_ID]
        fn = PositionSetConstraint(al, synth.RESPONSE_SEP)
    elif case["mask"] == "trie":
        trie = Trie([[synth.BOS_ID] + [int(t) for t in it] + [synth.EOS_ID] for it in items])
        fn = SuffixTrieConstraint(trie, synth.RESPONSE_SEP, synth.BOS_ID)
    elif case["mask"] == "chain":
        a, b = chain_sequences(prompt, items)
        ta = Trie(a)
        ta.append(Trie(b), HANDOVER)
        fn = prefix_allowed_tokens_fn(ta)
    else:
        fn = None
    return dict(target_dims=tdims, target_sd=tsd, draft_dims=ddims, draft_sd=dsd, prompt=prompt, fn=fn,
                items=items, vocab=vocab, procs=build_processors(case, V))


def _trie_seqs(n, seed):
    it = synth.synthetic_items(synth.CodeVocab("t", (5, 7, 6, 4), n), seed)
    return [[1] + [int(t) for t in r] + [2] for r in it]


TRIE_CASES = [
    dict(name="small", sequences=_trie_seqs(60, 3),
         queries=[[], [1], [1, 32000], [1, 32001, 32005], [1, 32004, 32011, 32013], [9], [1, 31999], [1, 32000, 32005, 32012, 32018],
                  [1, 32000, 32005, 32012, 32018, 2]],
         append=dict(sequences=[[7, 8, 9], [7, 8, 10], [5, 6]], bos=1, queries=[[7], [7, 8], [5], [1, 7], [3, 7, 8]])),
    dict(name="dups_and_prefixes", sequences=[[1, 2, 3], [1, 2, 3], [1, 2], [4], [1, 5, 6, 7]],
         queries=[[], [1], [1, 2], [1, 2, 3], [4], [4, 4], [1, 5, 6]]),
    dict(name="empty", sequences=[], queries=[[], [1]]),
]
