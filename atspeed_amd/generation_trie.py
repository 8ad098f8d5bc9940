"""Constraint layer of the beam-SD path: `Trie`, `prefix_allowed_tokens_fn` and the mask
functions the reference installs, plus their compilation to the device-side automaton.

Drop-in surface (same names / semantics as the reference):
  * `Trie(sequences=[])`, `.add/.get/.append/.load_from_dict/__iter__/__len__/__getitem__`
        <- reference `code/generation_trie.py:7-88`
  * `prefix_allowed_tokens_fn(trie)`            <- `code/generation_trie.py:92-98`
        (also `code/utils.py:201-207`)
  * `PositionSetConstraint`  (`fn(batch_id, sentence)`) <- `code/data.py:84-104`, the mask
        `code/inference.py:131` really uses
  * `SuffixTrieConstraint`                      <- `code/generate_teacher_data.py:174-188`

New here: every constraint can `compile()` itself into a `ConstraintFSM` — a CSR
automaton (node -> sorted child tokens -> next node) that the HIP scan kernels walk on the
device, replacing the reference's per-beam host call + `.tolist()` round trip
(`code/generation_trie.py:94`, `code/data.py:98`).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np


class Trie:
    """Prefix tree over token-id sequences, stored as nested dicts in `trie_dict`."""

    def __init__(self, sequences: Optional[Iterable[Sequence[int]]] = None):
        self.trie_dict: Dict[int, Dict] = {}
        self.len = 0
        self.append_trie: Optional["Trie"] = None
        self.bos_token_id: Optional[int] = None
        if sequences:
            for seq in sequences:
                self.add(seq)

    # -- mutation ---------------------------------------------------------------
    def add(self, sequence: Sequence[int]) -> None:
        level = self.trie_dict
        for token in sequence:
            nxt = level.get(token)
            if nxt is None:
                nxt = level[token] = {}
            level = nxt
        self.len += 1

    def append(self, trie: "Trie", bos_token_id: int) -> None:
        """Chain a second trie that takes over where this one has no continuation."""
        self.append_trie = trie
        self.bos_token_id = bos_token_id

    # -- lookup -----------------------------------------------------------------
    def _chained(self) -> bool:
        """the reference tests `if append_trie` (generation_trie.py:54,67): an appended trie with len() == 0 is falsy there"""
        return self.append_trie is not None and len(self.append_trie) != 0

    def _descend(self, prefix: Sequence[int]) -> Tuple[Optional[Dict], int]:
        level = self.trie_dict
        for depth, token in enumerate(prefix):
            nxt = level.get(token)
            if nxt is None:
                return None, depth
            level = nxt
        return level, len(prefix)

    def get(self, prefix_sequence: Sequence[int]) -> List[int]:
        prefix = [int(t) for t in prefix_sequence]
        level, depth = self._descend(prefix)
        chained = self._chained()
        if level is None:
            if chained:
                return self.append_trie.get(prefix[depth:])
            return []
        children = list(level)
        if chained and self.bos_token_id in children:
            children.remove(self.bos_token_id)
            children.extend(self.append_trie.trie_dict)
        return children

    def __getitem__(self, prefix_sequence: Sequence[int]) -> List[int]:
        return self.get(prefix_sequence)

    def __len__(self) -> int:
        return self.len

    def __iter__(self) -> Iterator[List[int]]:
        """Depth-first, insertion-ordered walk yielding every root-to-leaf sequence."""
        path: List[int] = []
        stack: List[Iterator] = [iter(self.trie_dict.items())]
        if not self.trie_dict:
            yield []
            return
        while stack:
            try:
                token, child = next(stack[-1])
            except StopIteration:
                stack.pop()
                if path:
                    path.pop()
                continue
            path.append(token)
            if child:
                stack.append(iter(child.items()))
            else:
                yield list(path)
                path.pop()

    @staticmethod
    def load_from_dict(trie_dict: Dict) -> "Trie":
        trie = Trie()
        trie.trie_dict = trie_dict
        trie.len = sum(1 for _ in trie)
        return trie

    # -- device form ------------------------------------------------------------
    def flatten(self, root_prefix: Sequence[int] = ()) -> "ConstraintFSM":
        """CSR automaton of the subtree below `root_prefix` (breadth-first node ids, children sorted by token id), built by
        the native `atspeed_trie_flatten` (include/atspeed_hip.h: the C-ABI counterpart of `Trie.__init__` / `_add_to_trie`,
        reference `code/generation_trie.py:8-14,40-44`).  A chained trie (`append`, `:19-21,55-57,67-68`) becomes ONE automaton
        over the nodes of every trie of the chain (`_flatten_chain`); `start` is then the node `root_prefix` leads to."""
        if self._chained():
            fsm = self._flatten_chain()
            return ConstraintFSM(fsm.row_ptr, fsm.tok, fsm.nxt, fsm.walk(0, [int(t) for t in root_prefix]), levels=fsm.levels)
        import ctypes as C
        from . import _lib
        root, _ = self._descend([int(t) for t in root_prefix])
        # root-to-leaf sequences of the subtree carry every node (a sequence that is a prefix of another adds none)
        seqs = [s for s in Trie.load_from_dict(root)] if root else []
        offsets = np.zeros(len(seqs) + 1, np.int32)
        if seqs:
            offsets[1:] = np.cumsum([len(s) for s in seqs])
        tokens = np.asarray([t for s in seqs for t in s], np.int32) if seqs else np.zeros(1, np.int32)
        lib = _lib.load()
        n_nodes, n_edges = C.c_int32(), C.c_int32()
        _lib.check(lib.atspeed_trie_flatten(tokens.ctypes.data, offsets.ctypes.data, len(seqs), None, None, None,
                                            C.byref(n_nodes), C.byref(n_edges)))
        row_ptr = np.zeros(n_nodes.value + 1, np.int32)
        tok = np.zeros(max(n_edges.value, 1), np.int32)
        nxt = np.zeros(max(n_edges.value, 1), np.int32)
        _lib.check(lib.atspeed_trie_flatten(tokens.ctypes.data, offsets.ctypes.data, len(seqs), row_ptr.ctypes.data, tok.ctypes.data,
                                            nxt.ctypes.data, C.byref(n_nodes), C.byref(n_edges)))
        return ConstraintFSM(row_ptr, tok[: n_edges.value], nxt[: n_edges.value], 0)

    def _flatten_chain(self) -> "ConstraintFSM":
        """The reference's chained lookup (`_get_from_trie`, generation_trie.py:46-70) as one automaton.  Walking a sentence through
        trie A: while the tokens are A's, stay in A; the first token A does not have hands the REST of the sentence (that token
        included) to the appended trie B, from B's root, for good (`:66-68`).  Where the walk ends inside A, the allowed set is the
        node's children with `bos_token_id` replaced by B's root tokens (`:52-58`).  So: nodes = A's nodes then B's (B flattened the
        same way, its own chain included); an A node that has the bos child allows (children - {bos}) + B's root tokens, the latter
        leading to B's first-level nodes (a token both have stays A's: A is asked first).  `levels` keeps every trie's own automaton for
        walking arbitrary prompt tokens, which need not be allowed ones."""
        chain, t = [], self
        while t is not None:
            chain.append(t)
            t = t.append_trie if t._chained() else None
        raws = []
        for tr in chain:
            plain = Trie.load_from_dict(tr.trie_dict)        # the same nodes without the chain
            raws.append(plain.flatten())
        offs = np.concatenate([[0], np.cumsum([r.n_nodes for r in raws])]).astype(np.int64)
        row_ptr, tok, nxt = [0], [], []
        for li, (tr, raw) in enumerate(zip(chain, raws)):
            nxt_raw = raws[li + 1] if li + 1 < len(raws) else None
            b_tok = nxt_raw.allowed(0) if nxt_raw is not None else None
            b_nxt = (nxt_raw.nxt[nxt_raw.row_ptr[0]: nxt_raw.row_ptr[1]] + offs[li + 1]) if nxt_raw is not None else None
            for n in range(raw.n_nodes):
                lo, hi = int(raw.row_ptr[n]), int(raw.row_ptr[n + 1])
                a_tok, a_nxt = raw.tok[lo:hi], raw.nxt[lo:hi].astype(np.int64) + offs[li]
                if nxt_raw is not None and tr.bos_token_id in a_tok:
                    keep = a_tok != tr.bos_token_id
                    a_tok, a_nxt = a_tok[keep], a_nxt[keep]
                    new = ~np.isin(b_tok, a_tok)
                    m_tok = np.concatenate([a_tok, b_tok[new]])
                    m_nxt = np.concatenate([a_nxt, b_nxt[new]])
                    order = np.argsort(m_tok, kind="stable")
                    a_tok, a_nxt = m_tok[order], m_nxt[order]
                tok += a_tok.tolist()
                nxt += a_nxt.tolist()
                row_ptr.append(len(tok))
        return ConstraintFSM(np.asarray(row_ptr, np.int32), np.asarray(tok, np.int32), np.asarray(nxt, np.int32), 0,
                             levels=[(int(o), r) for o, r in zip(offs[:-1], raws)])


@dataclass
class ConstraintFSM:
    """node n allows tokens tok[row_ptr[n]:row_ptr[n+1]] (ascending); taking tok[e] moves to nxt[e]."""
    row_ptr: np.ndarray   # int32 [n_nodes + 1]
    tok: np.ndarray       # int32 [n_edges]
    nxt: np.ndarray       # int32 [n_edges]
    start: int = 0
    free: bool = False    # no mask at all (prefix_allowed_tokens_fn=None): the arrays are empty, every token is a candidate
    levels: Optional[list] = None    # chained tries: [(node offset, that trie's own automaton)] for walking arbitrary tokens
    # (min_item_token, eos_token) of the post-top-k id filter; None = the reference's hard-coded (32000, 2) (code/beamSD.py:80-86);
    # min_item_token <= 0 keeps every pick (atspeed_fsm_set_id_filter)
    id_filter: Optional[Tuple[int, int]] = None

    @property
    def n_nodes(self) -> int:
        return len(self.row_ptr) - 1

    @property
    def max_children(self) -> int:
        return int(np.max(np.diff(self.row_ptr))) if self.n_nodes else 0

    def allowed(self, node: int) -> np.ndarray:
        return self.tok[self.row_ptr[node]: self.row_ptr[node + 1]]

    def step(self, node: int, token: int) -> int:
        lo, hi = int(self.row_ptr[node]), int(self.row_ptr[node + 1])
        j = lo + int(np.searchsorted(self.tok[lo:hi], token))
        if j >= hi or self.tok[j] != token:
            raise KeyError((node, token))
        return int(self.nxt[j])

    def walk(self, node: int, tokens: Sequence[int]) -> int:
        """The node a sentence leads to, as the reference's `Trie.get` walks it (tokens need not be allowed ones).  Plain automaton:
        `step` per token (KeyError = the reference returns []).  Chained tries: inside trie i of the chain use its OWN children (the
        bos child included); the first token it does not have restarts at the root of trie i + 1 with that token."""
        if not self.levels:
            for t in tokens:
                node = self.step(node, int(t))
            return node
        li = max(i for i, (off, _) in enumerate(self.levels) if off <= node)
        local = node - self.levels[li][0]
        for t in tokens:
            while True:
                try:
                    local = self.levels[li][1].step(local, int(t))
                    break
                except KeyError:
                    li += 1
                    local = 0
                    if li >= len(self.levels):
                        raise
        return self.levels[li][0] + local

    def validate(self, vocab_size: int) -> None:
        assert self.row_ptr[0] == 0 and np.all(np.diff(self.row_ptr) >= 0)
        assert len(self.tok) == len(self.nxt) == self.row_ptr[-1]
        if len(self.tok):
            assert self.tok.min() >= 0 and self.tok.max() < vocab_size
            assert self.nxt.min() >= 0 and self.nxt.max() < self.n_nodes
        for n in range(self.n_nodes):
            a = self.allowed(n)
            assert np.all(a[1:] > a[:-1]), "children must be strictly ascending"


def _tokens_after_last(sentence: List[int], sep: List[int]) -> Optional[int]:
    """How many tokens follow the last occurrence of `sep` in `sentence`."""
    m = len(sep)
    for start in range(len(sentence) - m, -1, -1):
        if sentence[start: start + m] == sep:
            return len(sentence) - start - m
    return None


def _as_list(sentence) -> List[int]:
    return [int(x) for x in (sentence.tolist() if hasattr(sentence, "tolist") else sentence)]


class PositionSetConstraint:
    """Allowed set = f(number of tokens generated after "Response:").

    `allowed_tokens[i]` is the id set legal at generated position i and
    `allowed_tokens[L] = {eos}` — the dict `BaseDataset.get_prefix_allowed_tokens_fn`
    builds (reference `code/data.py:84-94`).  Calling the object reproduces
    `code/data.py:96-102`: scan from the end for the separator, index by distance.
    """

    def __init__(self, allowed_tokens: Dict[int, Iterable[int]], sep: Sequence[int], id_filter: Optional[Tuple[int, int]] = None):
        self.allowed_tokens = {int(i): list(v) for i, v in allowed_tokens.items()}
        self.sep = [int(s) for s in sep]
        self.id_filter = id_filter                    # see ConstraintFSM.id_filter
        self._fsm: Optional[ConstraintFSM] = None     # CSR arrays are prompt-independent; only `start` varies

    def __call__(self, batch_id, sentence) -> Optional[List[int]]:
        i = _tokens_after_last(_as_list(sentence), self.sep)
        if i is None:
            return None          # same as the reference falling out of its loop
        return list(self.allowed_tokens[i])

    def compile(self, prompt: Sequence[int]) -> ConstraintFSM:
        i0 = _tokens_after_last([int(t) for t in prompt], self.sep)
        if i0 is None:
            raise TypeError("separator not found in prompt: the reference's mask function returns None here "
                            "(code/data.py:97-102) and the HF processor then fails with TypeError")
        if self._fsm is None:
            n_pos = max(self.allowed_tokens) + 1
            row_ptr, tok, nxt = [0], [], []
            for i in range(n_pos):
                ids = sorted(set(self.allowed_tokens.get(i, [])))
                tok += ids
                nxt += [min(i + 1, n_pos)] * len(ids)
                row_ptr.append(len(tok))
            row_ptr.append(len(tok))   # terminal node: nothing allowed (reference: KeyError)
            self._fsm = ConstraintFSM(np.asarray(row_ptr, np.int32), np.asarray(tok, np.int32), np.asarray(nxt, np.int32), 0)
        f = self._fsm
        if i0 >= f.n_nodes:
            raise KeyError(i0)         # reference: allowed_tokens[i] KeyError past the last position
        return ConstraintFSM(f.row_ptr, f.tok, f.nxt, i0, id_filter=self.id_filter)


class SuffixTrieConstraint:
    """Strict item trie keyed on `[bos] + tokens generated after "Response:"`
    (reference `code/generate_teacher_data.py:174-188`)."""

    def __init__(self, trie: Trie, sep: Sequence[int], bos_token_id: int = 1, id_filter: Optional[Tuple[int, int]] = None):
        self.trie = trie
        self.sep = [int(s) for s in sep]
        self.bos_token_id = int(bos_token_id)
        self.id_filter = id_filter
        self._fsm: Optional[ConstraintFSM] = None

    def _suffix(self, s: List[int]) -> List[int]:
        m = len(self.sep)
        for end in range(m, len(s) + 1):        # first occurrence wins, as in the reference loop
            if s[end - m: end] == self.sep:
                return s[end:]
        raise NameError("sentence_")            # the reference leaves `sentence_` unbound

    def __call__(self, batch_id, sentence) -> List[int]:
        return self.trie.get([self.bos_token_id] + self._suffix(_as_list(sentence)))

    def compile(self, prompt: Sequence[int]) -> ConstraintFSM:
        if self._fsm is None:
            self._fsm = self.trie.flatten([self.bos_token_id])
        fsm = self._fsm
        node = fsm.walk(fsm.start, self._suffix([int(x) for x in prompt]))
        return ConstraintFSM(fsm.row_ptr, fsm.tok, fsm.nxt, node, levels=fsm.levels, id_filter=self.id_filter)


class WholeSentenceTrieConstraint:
    """`prefix_allowed_tokens_fn(trie)` of the reference: the ENTIRE sentence, prompt
    included, is looked up in the trie (`code/generation_trie.py:92-98`)."""

    def __init__(self, trie: Trie, id_filter: Optional[Tuple[int, int]] = None):
        self.trie = trie
        self.id_filter = id_filter
        self._fsm: Optional[ConstraintFSM] = None

    def __call__(self, batch_id, sentence) -> List[int]:
        return self.trie.get(_as_list(sentence))

    def compile(self, prompt: Sequence[int]) -> ConstraintFSM:
        if self._fsm is None:
            self._fsm = self.trie.flatten()
        fsm = self._fsm
        try:
            node = fsm.walk(fsm.start, [int(t) for t in prompt])
        except KeyError:
            raise ValueError("`prefix_allowed_tokens_fn` returned an empty list for batch ID 0.") from None
        return ConstraintFSM(fsm.row_ptr, fsm.tok, fsm.nxt, node, levels=fsm.levels, id_filter=self.id_filter)


_FREE = ConstraintFSM(np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32), 0, free=True)


def free_constraint() -> ConstraintFSM:
    """The automaton of a call without a mask (`prefix_allowed_tokens_fn=None`, no logits processor): every token is a candidate and
    the reference's post-top-k id filter is off (`code/beamSD.py:80`).  One shared object, so a batch of users shares it."""
    return _FREE


def prefix_allowed_tokens_fn(candidate_trie: Trie) -> WholeSentenceTrieConstraint:
    """Same call as the reference factory; the returned callable is also compilable."""
    return WholeSentenceTrieConstraint(candidate_trie)
