"""ORACLE (test infrastructure, not product code): restatement of the reference's
constraint layer — the prefix tree of `code/generation_trie.py:7-88`, the whole-sentence
mask function of `code/generation_trie.py:92-98`, the position-set mask that
`code/inference.py:131` actually installs (`code/data.py:84-104`) and the suffix-keyed
strict-trie function of `code/generate_teacher_data.py:174-188`.

Pinned in tests/test_oracle_pins.py against the imported reference Trie (golden
vectors under tests/golden/trie_*.json were produced by the reference class itself).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence


class RefTrie:
    """Nested-dict prefix tree, same observable behaviour as the reference class."""

    def __init__(self, sequences: Optional[Iterable[Sequence[int]]] = None):
        self.trie_dict: Dict = {}
        self.len = 0
        self.append_trie = None
        self.bos_token_id = None
        for seq in sequences or []:          # generation_trie.py:11-14
            self.add(seq)

    def add(self, sequence: Sequence[int]) -> None:   # generation_trie.py:23-25,40-44
        node = self.trie_dict
        for tok in sequence:
            node = node.setdefault(tok, {})
        self.len += 1                       # counts inserts, duplicates included

    def append(self, trie: "RefTrie", bos_token_id: int) -> None:   # generation_trie.py:19-21
        self.append_trie = trie
        self.bos_token_id = bos_token_id

    def get(self, prefix: Sequence[int]) -> List[int]:   # generation_trie.py:27-30,47-70
        node = self.trie_dict
        prefix = list(prefix)
        for i, tok in enumerate(prefix):
            if tok in node:
                node = node[tok]
            elif self.append_trie is not None:
                # the reference recurses with the REMAINING prefix into the appended trie (:67-68)
                return self.append_trie.get(prefix[i:])
            else:
                return []
        out = list(node.keys())
        if self.append_trie is not None and self.bos_token_id in out:   # :55-57
            out.remove(self.bos_token_id)
            out += list(self.append_trie.trie_dict.keys())
        return out

    __getitem__ = get                                    # generation_trie.py:87-88

    def __len__(self) -> int:                            # generation_trie.py:84-85
        return self.len

    def __iter__(self):                                  # generation_trie.py:72-82 (DFS, leaves only)
        def walk(prefix, node):
            if node:
                for tok in node:
                    yield from walk(prefix + [tok], node[tok])
            else:
                yield prefix
        return walk([], self.trie_dict)

    @staticmethod
    def load_from_dict(trie_dict: Dict) -> "RefTrie":   # generation_trie.py:32-37
        t = RefTrie()
        t.trie_dict = trie_dict
        t.len = sum(1 for _ in t)
        return t


def ref_whole_sentence_fn(trie: RefTrie):
    """generation_trie.py:92-98 — looks up the ENTIRE sentence (prompt included)."""
    def fn(batch_id, sentence):
        return trie.get([int(x) for x in sentence.tolist()])
    return fn


def _find_sep_from_end(sentence: List[int], sep: Sequence[int]) -> Optional[int]:
    """Number of tokens after the LAST occurrence of `sep` (data.py:97-102 scans the
    reversed sentence for the reversed separator and returns the first hit)."""
    n, m = len(sentence), len(sep)
    sep = list(sep)
    for i in range(0, n - m + 1):
        if sentence[n - i - m: n - i] == sep:
            return i
    return None


def ref_position_set_fn(allowed_tokens: Dict[int, Iterable[int]], sep: Sequence[int]):
    """data.py:84-104 — allowed set depends only on how many tokens follow "Response:"."""
    allowed = {i: list(v) for i, v in allowed_tokens.items()}
    def fn(batch_id, sentence):
        i = _find_sep_from_end([int(x) for x in sentence.tolist()], sep)
        if i is None:
            return None                                   # reference falls off the loop -> None
        return list(allowed[i])
    return fn


def ref_suffix_trie_fn(trie: RefTrie, sep: Sequence[int], bos: int = 1):
    """generate_teacher_data.py:174-188 — strict trie keyed on [bos] + generated suffix."""
    sep = list(sep)
    def fn(batch_id, sentence):
        s = [int(x) for x in sentence.tolist()]
        # the reference loop runs i = len..0 WITHOUT a break, so the match with the
        # smallest i (the FIRST occurrence of the separator) is the one that sticks
        key = None
        for i in range(len(sep), len(s) + 1):
            if s[i - len(sep): i] == sep:
                key = [bos] + s[i:]
                break
        if key is None:
            raise NameError("sentence_")   # reference: unbound local when "Response:" is absent
        return trie.get(key)
    return fn


def ref_flatten(trie_dict: Dict, root_prefix: Sequence[int] = ()):
    """Plain-Python flattening of a nested-dict trie into the CSR automaton (breadth-first node ids, children ascending by
    token id): the comparator for the native `atspeed_trie_flatten`.  Returns (row_ptr, tok, nxt) lists."""
    root = trie_dict
    for t in root_prefix:
        root = root.get(t)
        if root is None:
            root = {}
            break
    nodes, row_ptr, tok, nxt = [root], [0], [], []
    i = 0
    while i < len(nodes):
        for t in sorted(nodes[i]):
            tok.append(int(t))
            nxt.append(len(nodes))
            nodes.append(nodes[i][t])
        row_ptr.append(len(tok))
        i += 1
    return row_ptr, tok, nxt
