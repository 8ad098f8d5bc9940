"""Product-side constraint layer (atspeed_amd/generation_trie.py) vs the reference's golden
vectors and vs the oracle restatement; FSM compilation checks.  CPU only."""
import numpy as np
import pytest
import torch

from atspeed_amd import synth
from atspeed_amd.generation_trie import (PositionSetConstraint, SuffixTrieConstraint, Trie,
                                         WholeSentenceTrieConstraint, prefix_allowed_tokens_fn)
from oracle.trie_ref import RefTrie, ref_flatten, ref_position_set_fn, ref_suffix_trie_fn
from tests.golden.cases import TRIE_CASES


@pytest.mark.parametrize("tc", TRIE_CASES, ids=[t["name"] for t in TRIE_CASES])
def test_trie_equals_reference_golden(tc, trie_golden):
    gold = trie_golden[tc["name"]]
    t = Trie(tc["sequences"])
    assert len(t) == gold["len"]
    assert [list(x) for x in t] == gold["iter"]
    for q, exp in gold["gets"]:
        assert t.get(q) == exp and t[q] == exp
    fn = prefix_allowed_tokens_fn(Trie(tc["sequences"]))
    for q, exp in gold["fn"]:
        assert fn(0, torch.tensor(q, dtype=torch.long)) == exp
    if tc.get("append"):
        t.append(Trie(tc["append"]["sequences"]), tc["append"]["bos"])
        for q, exp in gold["gets_appended"]:
            assert t.get(q) == exp
    assert len(Trie.load_from_dict(t.trie_dict)) == gold["loaded_len"]


def test_trie_add_and_incremental_len():
    t = Trie()
    assert len(t) == 0 and t.get([]) == []
    t.add([1, 2, 3]); t.add([1, 2, 4]); t.add([1, 2, 3])
    assert len(t) == 3 and t.get([1, 2]) == [3, 4] and t.get([9]) == []


def test_flatten_matches_nested_lookup():
    items = synth.synthetic_items(synth.BEAUTY)
    seqs = [[1] + [int(x) for x in r] + [2] for r in items]
    t = Trie(seqs)
    fsm = t.flatten([1])
    fsm.validate(synth.BEAUTY.vocab_size)
    rng = np.random.default_rng(0)
    for r in rng.integers(0, len(seqs), 200):
        node, pre = 0, [1]
        for tok in seqs[r][1:]:
            assert sorted(t.get(pre)) == fsm.allowed(node).tolist()
            node = fsm.step(node, tok)
            pre.append(tok)
        assert fsm.allowed(node).tolist() == []
    # breadth-first layout: node counts per depth match the number of distinct prefixes
    assert fsm.n_nodes == 1 + sum(len({tuple(s[1:d]) for s in seqs}) for d in range(2, 7))
    assert fsm.max_children <= 256


def test_position_set_constraint_and_compile():
    al = synth.BEAUTY.allowed_tokens()
    c = PositionSetConstraint(al, synth.RESPONSE_SEP)
    ref = ref_position_set_fn(al, synth.RESPONSE_SEP)
    p = synth.synthetic_prompt(30, 1)
    for extra in ([], [32000], [32000, 32100], [32000, 32100, 32400, 32700]):
        s = torch.tensor(list(p) + extra)
        assert c(0, s) == ref(0, s)
    assert c(0, torch.tensor([1, 5, 6])) is None
    fsm = c.compile(p)
    fsm.validate(synth.BEAUTY.vocab_size)
    assert fsm.start == 0 and fsm.n_nodes == 6
    for i in range(4):
        lo, hi = synth.BEAUTY.level_range(i)
        assert fsm.allowed(i).tolist() == list(range(lo, hi))
    assert fsm.allowed(4).tolist() == [2] and fsm.allowed(5).tolist() == []
    assert c.compile(list(p) + [32000]).start == 1
    with pytest.raises(TypeError):
        c.compile([1, 5, 6])


def test_suffix_trie_constraint_and_compile():
    items = synth.synthetic_items(synth.TINY)
    seqs = [[1] + [int(x) for x in r] + [2] for r in items]
    c = SuffixTrieConstraint(Trie(seqs), synth.RESPONSE_SEP, 1)
    ref = ref_suffix_trie_fn(RefTrie(seqs), synth.RESPONSE_SEP, 1)
    p = list(synth.synthetic_prompt(20, 3))
    it = [int(x) for x in items[17]]
    for n in range(5):
        s = torch.tensor(p + it[:n])
        assert sorted(c(0, s)) == sorted(ref(0, s))
        fsm = c.compile(p + it[:n])
        assert fsm.allowed(fsm.start).tolist() == sorted(c(0, s))
    with pytest.raises(KeyError):
        c.compile(p + [32000 + 63, 32000])       # not a prefix of any item


def test_whole_sentence_constraint_compile():
    t = Trie([[1, 5, 7], [1, 5, 8], [1, 6, 9]])
    c = WholeSentenceTrieConstraint(t)
    assert c.compile([1, 5]).allowed(c.compile([1, 5]).start).tolist() == [7, 8]
    with pytest.raises(ValueError):
        c.compile([1, 4])


def _same_csr(fsm, ref):
    return fsm.row_ptr.tolist() == ref[0] and fsm.tok.tolist() == ref[1] and fsm.nxt.tolist() == ref[2]


def test_native_trie_flatten_equals_python_flattening():
    """`Trie.flatten` runs the C-ABI `atspeed_trie_flatten` (host code: works without a GPU); the comparator is the plain-Python
    breadth-first flattening in the oracle.  Empty tries, duplicate and prefix sequences, a root prefix that is absent."""
    cases = [[], [[5]], [[1, 2, 3], [1, 2, 3], [1, 2], [4], [1, 5, 6, 7]],
             [[1] + [int(x) for x in r] + [2] for r in synth.synthetic_items(synth.GAMES)]]
    for seqs in cases:
        t = Trie(seqs)
        for pre in ((), (1,), (1, 2), (9, 9)):
            assert _same_csr(t.flatten(pre), ref_flatten(RefTrie(seqs).trie_dict, pre)), (len(seqs), pre)


def _chain_cases():
    it = synth.synthetic_items(synth.CodeVocab("t", (5, 7, 6, 4), 80), 3)
    a = [[1] + [int(x) for x in r[:2]] + [9] for r in it]               # two code levels, then the hand-over token 9
    b = [[int(x) for x in r[2:]] + [2] for r in it]
    c = [[7, 8], [7, 5, 6]]
    return [(a, 9, b, None, None), (a, 9, b, 2, c), ([[1, 2, 3], [1, 4]], 3, [[5, 6], [2, 7]], None, None)]


def test_chained_trie_flattens_to_one_automaton_that_answers_like_get(trie_golden):
    """`Trie.append` (generation_trie.py:19-21,55-57,67-68) on the device path: the chain becomes ONE CSR automaton.  For every
    sentence -- all prefixes of every trie's sequences, concatenations across the hand-over, sentences that miss -- the automaton's
    node after `walk` allows exactly what the oracle's `RefTrie.get` returns (as a set: the reference's list may hold a token twice),
    and allowed tokens lead where the longer sentence leads.  The reference's own answers (golden `gets_appended`) are checked too."""
    for a, bos_a, b, bos_b, c in _chain_cases():
        t, rt = Trie(a), RefTrie(a)
        tb, rtb = Trie(b), RefTrie(b)
        if c is not None:
            tb.append(Trie(c), bos_b); rtb.append(RefTrie(c), bos_b)
        t.append(tb, bos_a); rt.append(rtb, bos_a)
        fsm = t.flatten()
        fsm.validate(1 << 20)
        sentences = {()}
        for seqs in (a, b, c or []):
            for sq in seqs:
                for j in range(len(sq) + 1):
                    sentences.add(tuple(sq[:j]))
        for sa in a[:20]:                                                # across the hand-over: A's prefix without the bos token, then B
            for sb in b[:20]:
                for j in range(len(sb) + 1):
                    sentences.add(tuple(sa[:-1]) + tuple(sb[:j]))
        sentences |= {(424242,), (1, 424242), tuple(a[0][:2]) + (424242,)}
        for sq in sorted(sentences):
            want = sorted(set(rt.get(list(sq))))
            assert sorted(set(t.get(list(sq)))) == want
            try:
                node = fsm.walk(0, sq)
            except KeyError:
                assert want == [], sq
                continue
            assert fsm.allowed(node).tolist() == want, sq
            for tok in want[:3]:                                         # consistency of the edges: step == walk of the longer sentence
                assert fsm.step(node, tok) == fsm.walk(0, sq + (tok,))
    tc = next(c for c in TRIE_CASES if c.get("append"))
    t = Trie(tc["sequences"]); t.append(Trie(tc["append"]["sequences"]), tc["append"]["bos"])
    fsm = t.flatten()
    for q, exp in trie_golden[tc["name"]]["gets_appended"]:
        try:
            got = fsm.allowed(fsm.walk(0, q)).tolist()
        except KeyError:
            got = []
        assert got == sorted(set(exp)), q


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/data"), reason="reference data files are only in the build container")
def test_native_trie_flatten_on_the_real_item_tries():
    """The strict tries of the real Beauty / Games indices (12 023 / 17 289 distinct code tuples, keyed bos + codes + eos as
    inference.py:130 builds them): native CSR == Python flattening; node counts per depth as SURVEY.md 8a row T records."""
    from atspeed_amd.harness import ItemIndex
    for name, depth_nodes in (("beauty", [91, 6539, 11172, 12023]), ("games", [248, 11317, 16628, 17289])):
        tr = ItemIndex.load("/root/reference/data", name).trie()
        fsm = tr.flatten([1])
        assert _same_csr(fsm, ref_flatten(tr.trie_dict, [1]))
        level, counts = [0], []
        for _ in range(4):
            level = [int(n) for p in level for n in fsm.nxt[fsm.row_ptr[p]: fsm.row_ptr[p + 1]]]
            counts.append(len(level))
        assert counts == depth_nodes
