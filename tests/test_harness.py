"""Harness either side of the hot path: item index, test-set construction, prompt encoding and the
ranking metrics, on CPU.  Pinned to the REFERENCE: tests/golden/harness_golden.json holds the outputs of the reference's own
`computeTopNAccuracy` (utils.py:215-271), `SeqRecDataset(mode="test")` (data.py:112-278) and
`BaseDataset.get_prefix_allowed_tokens_fn` (data.py:84-104), produced by tests/golden/gen_harness_golden.py importing them.
Facts about the real Beauty / Games files are checked when the data files are present (this container); scikit-learn and
hand-computed values stay as independent cross-checks of the formulas."""
import json
import math
import os

import numpy as np
import pytest

from atspeed_amd import synth
from atspeed_amd.harness import (CodeTokenEncoder, InferenceResult, ItemIndex, SeqRecTestData, computeTopNAccuracy,
                                 encode_prompt, load_interactions)

REF_DATA = "/root/reference/data"
has_ref = os.path.isdir(REF_DATA)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(ROOT, "tests", "golden", "harness_golden.json")) as f:
        return json.load(f)


def _digest(strings):
    import hashlib
    h = hashlib.sha256()
    for s in strings:
        h.update(s.encode("utf-8"))
        h.update(b"\x00")
    return h.hexdigest()


def test_metrics_equal_the_references_function(gold):
    g = gold["metrics"]
    p, r, n, m = computeTopNAccuracy(g["truth"], g["pred"], g["topN"])
    assert (p, r, n, m) == (g["precision"], g["recall"], g["ndcg"], g["mrr"])          # rounded to 4 digits on both sides: exact
    s = g["strings_first40"]
    truth = [[f"i{x}" for x in t] for t in g["truth"][:40]]
    pred = [[f"i{x}" for x in q] for q in g["pred"][:40]]
    assert computeTopNAccuracy(truth, pred, s["topN"]) == (s["precision"], s["recall"], s["ndcg"], s["mrr"])


def _int_keys(d):
    return {int(k): v for k, v in d.items()}


@pytest.mark.parametrize("variant", ["default", "his5_prefix", "nolimit_sep"])
def test_test_split_equals_the_references_seqrecdataset(gold, variant):
    """Same users in the same order, same label strings, same prompt TEXT as `SeqRecDataset(args, mode="test")` built from the same
    index / train / valid / test dicts (the fixture carries the inputs)."""
    g = gold["synthetic"]
    v = g["variants"][variant]
    ix = ItemIndex(g["index"])
    d = SeqRecTestData(ix, _int_keys(g["train"]), _int_keys(g["valid"]), _int_keys(g["test"]), **v["kw"])
    sp = v["split"]
    assert len(d) == sp["n_users"]
    assert [len(u.history) for u in d.users] == sp["history_len"]
    for i, u in enumerate(d.users):
        ref = sp["users"][str(i)]
        assert [ix.item_string(x) for x in u.labels] == ref["labels"]
        assert d.text(u) == ref["text"]
    assert _digest(d.text(u) for u in d.users) == sp["prompts_digest"]
    assert _digest("|".join(ix.item_string(x) for x in u.labels) for u in d.users) == sp["labels_digest"]


def test_position_set_mask_equals_the_references_closure(gold):
    """`get_prefix_allowed_tokens_fn` (data.py:84-104) driven by a stub tokenizer with the ids the extended tokenizer assigns:
    allowed-token dict, new-token order and the closure's answers (incl. None without a separator, last separator wins)."""
    g = gold["synthetic"]
    m = g["mask"]
    ix = ItemIndex(g["index"])
    assert len(ix.new_tokens) == m["n_new_tokens"] and _digest(ix.new_tokens) == m["new_tokens_digest"]
    assert len(ix.all_items()) == m["all_items"]
    assert {str(i): v for i, v in ix.allowed_tokens().items()} == m["allowed_tokens"]
    d = SeqRecTestData(ix, _int_keys(g["train"]), _int_keys(g["valid"]), _int_keys(g["test"]))
    fn = d.get_prefix_allowed_tokens_fn()
    import torch
    for sentence, want in m["fn"]:
        got = fn(0, torch.tensor(sentence))
        assert (None if got is None else sorted(got)) == want
        if want is not None:                                                # and the compiled automaton allows the same set there
            fsm = fn.compile(sentence)
            assert fsm.allowed(fsm.start).tolist() == want


@pytest.mark.skipif(not has_ref, reason="reference data files are only in the build container")
@pytest.mark.parametrize("name", ["beauty", "games"])
def test_real_splits_equal_the_references(gold, name):
    g = gold["real"][name]
    d = SeqRecTestData.load(REF_DATA, name)
    ix = d.index
    sp = g["split"]
    assert len(d) == sp["n_users"]
    assert _digest(d.text(u) for u in d.users) == sp["prompts_digest"]
    assert _digest("|".join(ix.item_string(x) for x in u.labels) for u in d.users) == sp["labels_digest"]
    for i, ref in sp["users"].items():
        u = d.users[int(i)]
        assert d.text(u) == ref["text"] and [ix.item_string(x) for x in u.labels] == ref["labels"]
    m = g["mask"]
    assert len(ix.new_tokens) == m["n_new_tokens"] and _digest(ix.new_tokens) == m["new_tokens_digest"] and len(ix.all_items()) == m["all_items"]
    al = ix.allowed_tokens()
    assert {str(i): [v[0], v[-1] + 1, len(v)] for i, v in al.items()} == m["allowed_ranges"]
    assert _digest(json.dumps(al[i]) for i in sorted(al, key=str)) == m["allowed_digest"]
    fn = d.get_prefix_allowed_tokens_fn()
    import torch
    for sentence, want in m["fn"]:
        got = sorted(fn(0, torch.tensor(sentence)))
        assert (got if len(got) < 8 else [got[0], got[-1] + 1, len(got)]) == want


def tiny_index():
    rng = np.random.default_rng(0)
    idx = {}
    for i in range(40):
        idx[str(i)] = [f"<a_{rng.integers(5)}>", f"<b_{rng.integers(7)}>", f"<c_{rng.integers(7)}>", f"<d_{rng.integers(9)}>"]
    idx["40"] = idx["3"]          # two items sharing one code tuple, as in the real files
    return ItemIndex(idx)


def test_metrics_hand_values():
    truth = [[5], [1, 2], []]
    pred = [[9, 5, 7, 8], [2, 9, 1, 8], [1, 2, 3, 4]]
    p, r, n, m = computeTopNAccuracy(truth, pred, [1, 2, 4])
    # user 0: hit at rank 2; user 1: hits at ranks 1 and 3; user 2 skipped (empty truth)
    assert p == [round((0 + 1) / 2 / 1, 4), round((1 / 2 + 1 / 2) / 2, 4), round((1 / 4 + 2 / 4) / 2, 4)]
    assert r == [round((0 + 0.5) / 2, 4), round((1 + 0.5) / 2, 4), round((1 + 1) / 2, 4)]
    ndcg4_u0 = (1 / math.log2(3)) / 1.0
    ndcg4_u1 = (1 + 1 / math.log2(4)) / (1 + 1 / math.log2(3))
    assert n[2] == round((ndcg4_u0 + ndcg4_u1) / 2, 4)
    assert m == [round((0 + 1) / 2, 4), round((0.5 + 1) / 2, 4), round((0.5 + 1) / 2, 4)]


def test_ndcg_matches_sklearn():
    from sklearn.metrics import ndcg_score
    rng = np.random.default_rng(1)
    n_items, K = 50, 10
    truth, pred, rel, score = [], [], [], []
    for _ in range(64):
        t = rng.choice(n_items, size=rng.integers(1, 4), replace=False).tolist()
        order = rng.permutation(n_items)[:K].tolist()
        truth.append(t); pred.append(order)
        y = np.zeros(n_items); y[t] = 1
        s = np.full(n_items, -1.0)
        for rank, it in enumerate(order):
            s[it] = K - rank
        rel.append(y); score.append(s)
    _, _, ndcg, _ = computeTopNAccuracy(truth, pred, [5, 10])
    for k, mine in zip((5, 10), ndcg):
        assert abs(mine - ndcg_score(np.array(rel), np.array(score), k=k)) < 1e-4 + 5e-5


def test_item_index_ids_and_masks():
    ix = tiny_index()
    assert ix.new_tokens == sorted(ix.new_tokens) and ix.vocab_size == 32000 + len(ix.new_tokens)
    assert all(ix.token_id[t] == 32000 + r for r, t in enumerate(ix.new_tokens))
    al = ix.allowed_tokens()
    assert sorted(al) == [0, 1, 2, 3, 4] and al[4] == [2]
    # sorted token strings group by level letter: each position owns one contiguous id range
    for i in range(4):
        assert al[i] == list(range(al[i][0], al[i][0] + len(al[i])))
    assert ix.decode(ix.item_codes[7]) == 7
    assert ix.decode(ix.item_codes[40]) == 3 and ix.same_item(3, 40)
    assert ix.decode((32000, 32000, 32000, 32000)) == -1
    tr = ix.trie()
    assert len(tr) == len(set(ix.item_codes.values()))
    assert sorted(tr.get([1])) == sorted({c[0] for c in ix.item_codes.values()})
    c = ix.item_codes[7]
    assert tr.get([1] + list(c)) == [2] and tr.get([1] + list(c) + [2]) == []


def test_test_data_and_prompt_encoding():
    ix = tiny_index()
    train = {0: [1, 2, 3], 1: list(range(30)), 2: [5]}
    valid = {0: [4], 1: [31], 2: []}
    test = {0: [9], 1: [32, 33], 2: [], 3: [7]}
    d = SeqRecTestData(ix, train, valid, test, max_his_len=20)
    assert [u.uid for u in d.users] == [0, 1, 3]                       # user 2 has no test item
    assert d.users[0].history == [1, 2, 3, 4] and d.users[0].labels == [9]
    assert d.users[1].history == (list(range(30)) + [31])[-20:] and len(d.users[1].history) == 20
    assert d.users[2].history == []
    txt = d.text(d.users[0])
    assert txt.startswith("Below is an instruction") and txt.endswith("### Response:")
    assert ", ".join("".join(ix.indices[i]) for i in [1, 2, 3, 4]) in txt
    enc = CodeTokenEncoder(ix)
    ids = encode_prompt(d, d.users[0], None, enc)
    assert ids[0] == 1 and tuple(ids[-2:]) == synth.RESPONSE_SEP
    assert [int(t) for t in ids if t >= 32000] == [c for i in [1, 2, 3, 4] for c in ix.item_codes[i]]
    fn = d.get_prefix_allowed_tokens_fn()
    assert sorted(fn(0, ids)) == ix.allowed_tokens()[0]
    strict = d.strict_trie_fn()
    assert sorted(strict(0, ids)) == sorted({c[0] for c in ix.item_codes.values()})
    nxt = np.concatenate([ids, [ix.item_codes[7][0]]])
    assert sorted(strict(0, nxt)) == sorted({c[1] for c in ix.item_codes.values() if c[0] == ix.item_codes[7][0]})

    class Tok:                                                          # any object with .encode works as the tokenizer
        def encode(self, text):
            return [1] + [3 + (ord(ch) % 100) for ch in text[:12]] + list(synth.RESPONSE_SEP)
    ids2 = encode_prompt(d, d.users[0], Tok())
    assert tuple(ids2[-2:]) == synth.RESPONSE_SEP and len(ids2) == 15


def test_result_metrics_compare_by_code_tuple():
    ix = tiny_index()
    r = InferenceResult(uids=[0, 1], predictions=[[3, 5, -1, 8], [6, 7, 8, 9]], scores=[[0] * 4] * 2, labels=[[40], [9]],
                        rows=[{"n_run": 3, "total_accept_steps": 0}] * 2, wall_s=1.0)
    m = r.metrics(ix, topN=(1, 4, 20))
    assert m["topN"] == [1, 4]                                          # cut-offs beyond the K returned beams are dropped
    assert m["recall"] == [0.5, 1.0]                                    # item 40 shares item 3's codes: predicting 3 is a hit
    assert r.counters()["items_per_s"] == 8.0


@pytest.mark.skipif(not has_ref, reason="reference data files are only in the build container")
def test_real_index_files_match_the_survey():
    b = ItemIndex.load(REF_DATA, "beauty")
    assert len(b.indices) == 12035 and len(set(b.item_codes.values())) == 12023 and b.vocab_size == synth.BEAUTY.vocab_size == 32859
    al = b.allowed_tokens()
    assert [(al[i][0], al[i][-1] + 1) for i in range(4)] == [(32000, 32091), (32091, 32347), (32347, 32603), (32603, 32859)] and al[4] == [2]
    assert {i: list(v) for i, v in synth.BEAUTY.allowed_tokens().items()} == al
    g = ItemIndex.load(REF_DATA, "games")
    assert len(g.indices) == 17332 and g.vocab_size == synth.GAMES.vocab_size == 33014
    assert len(b.trie()) == 12023


@pytest.mark.skipif(not has_ref, reason="reference data files are only in the build container")
def test_real_test_split_statistics():
    d = SeqRecTestData.load(REF_DATA, "beauty")
    assert len(d) == 3553
    H = np.array([len(u.history) for u in d.users])
    assert abs(H.mean() - 7.33) < 0.01 and np.median(H) == 5 and H.max() == 20
    tr, va, te = load_interactions(REF_DATA, "games")                   # only the sequential_*.txt form exists for Games
    g = SeqRecTestData(ItemIndex.load(REF_DATA, "games"), tr, va, te)
    assert len(g) == 8696
    ids = encode_prompt(d, d.users[0])
    assert len(ids) == 4 + 5 * len(d.users[0].history) - 1 + 4 + 2 if d.users[0].history else True


def test_capacity_for_sizes_the_arenas_from_the_longest_prompt():
    """`python -m atspeed_amd.inference` sizes max_slots / max_tokens / max_logit_rows from the data (code/utils.py:119 allows 512-token prompts:
    with 3 x 40 draft tokens they do not fit the library's default 512 slots)."""
    from atspeed_amd.harness import capacity_for
    assert capacity_for(108, 20, 40) == dict(max_slots=512, max_tokens=512, max_logit_rows=384)        # the Beauty mean: the defaults
    c = capacity_for(512, 20, 40)
    assert c["max_slots"] >= 512 + 3 * 40 + 3 * 20 and c["max_slots"] % 64 == 0 and c["max_slots"] <= 2048
    assert c["max_tokens"] >= 512 + 3 * 40 and c["max_logit_rows"] >= 1 + 3 * 40
    assert capacity_for(400, 20, 40)["max_slots"] > 512                                             # the VERDICT's ~390-token threshold
    assert capacity_for(300, 64, 64, gamma=4, max_new_tokens=7)["max_logit_rows"] >= 64 + 4 * 64
    import pytest
    with pytest.raises(ValueError):
        capacity_for(1900, 20, 40)
