"""Dataset -> prompt -> beam-SD -> ranking metrics: the harness either side of the hot path.

Mirrors, at the level of token ids, what the reference does around `BSSD`:
  * `ItemIndex`           — the item-index JSON (`<dataset>.LCRec-1e-3lr.json`): new tokens sorted as in
                            `code/data.py:46-57`, ids 32000 + rank (`code/finetune_llama.py:84`), the per-position
                            allowed-token sets of `code/data.py:84-96` and the strict item trie of `code/inference.py:130`;
  * `SeqRecTestData`      — `SeqRecDataset(mode="test")`: history = train + valid interactions cut to the last
                            `max_his_len`, labels = the test items, users without test items skipped
                            (`code/data.py:139-160,232-262`); the instruction text is the reference's `sft_prompt`
                            (`code/data.py:19-20,245-247`);
  * `computeTopNAccuracy` — precision / recall / NDCG / MRR at each cut-off, as `code/utils.py:215-271`;
  * `run_inference`       — the loop of `code/inference.py:162-187` (BSSD per user, timing columns of its CSV), with
                            users decoded in lock-step batches and the K returned code sequences mapped back to items.

No tokenizer ships offline.  With one (anything with `.encode(text) -> ids`, e.g. the HF LlamaTokenizer extended with
the new tokens) prompts are tokenised text as in the reference; without one, `CodeTokenEncoder` lays the history out
directly as code-token ids between synthetic template tokens and the real "Response:" separator ids, which is all the
decoding path reads from a prompt (the position-set mask searches the separator, `code/data.py:97-102`).
"""
from __future__ import annotations

import json
import math
import os
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import synth
from .generation_trie import PositionSetConstraint, SuffixTrieConstraint, Trie

SFT_PROMPT = ("Below is an instruction that describes a task. Write a response that appropriately completes the request."
              "\n\n### Instruction:\n{}\n\n### Response:")
INSTRUCTION = ("The user has interacted with items {} in chronological order. Can you predict the next possible item that the "
               "user may expect?")
BASE_VOCAB = 32000          # Llama vocabulary; new item tokens are appended after it (finetune_llama.py:84)
EOS = 2


# ------------------------------------------------------------------------------------------------ item index
class ItemIndex:
    """item id -> tuple of code tokens; token ids as the reference's extended tokenizer assigns them."""

    def __init__(self, indices: Dict[str, Sequence[str]], base_vocab: int = BASE_VOCAB, eos: int = EOS):
        self.indices = {int(k): tuple(v) for k, v in indices.items()}
        self.base_vocab, self.eos = base_vocab, eos
        self.new_tokens: List[str] = sorted({t for idx in self.indices.values() for t in idx})      # data.py:46-57
        self.token_id: Dict[str, int] = {t: base_vocab + r for r, t in enumerate(self.new_tokens)}
        self.vocab_size = base_vocab + len(self.new_tokens)
        self.n_levels = len(next(iter(self.indices.values())))
        self.item_codes: Dict[int, Tuple[int, ...]] = {i: tuple(self.token_id[t] for t in idx) for i, idx in self.indices.items()}
        self._by_codes: Dict[Tuple[int, ...], int] = {}
        for i in sorted(self.item_codes):                       # duplicates (same codes): the lowest item id names the code
            self._by_codes.setdefault(self.item_codes[i], i)

    @classmethod
    def load(cls, data_path: str, dataset: str, index_file: str = ".LCRec-1e-3lr.json", **kw) -> "ItemIndex":
        with open(os.path.join(data_path, dataset, dataset + index_file)) as f:
            return cls(json.load(f), **kw)

    def item_string(self, item: int) -> str:
        return "".join(self.indices[item])                      # data.py:149-160 ("remapped" items)

    def all_items(self) -> set:
        return {"".join(idx) for idx in self.indices.values()}  # data.py:59-68

    def allowed_tokens(self) -> Dict[int, List[int]]:
        """position -> sorted allowed ids, EOS at the last position (data.py:84-96)."""
        allowed: Dict[int, set] = {}
        for codes in self.item_codes.values():
            for i, t in enumerate(codes):
                allowed.setdefault(i, set()).add(t)
        allowed[len(allowed)] = {self.eos}
        return {i: sorted(v) for i, v in allowed.items()}

    def trie(self, bos: int = 1) -> Trie:
        """Strict item trie, keys `[bos] + code ids + [eos]` as `tokenizer.encode(item) + [eos]` builds them (inference.py:130)."""
        return Trie([[bos] + list(c) + [self.eos] for c in sorted(set(self.item_codes.values()))])

    def decode(self, codes: Sequence[int]) -> int:
        """generated code ids -> item id, -1 when the tuple names no item (possible under the position-set mask)."""
        return self._by_codes.get(tuple(int(c) for c in codes[: self.n_levels]), -1)

    def same_item(self, a: int, b: int) -> bool:
        return self.item_codes.get(a) == self.item_codes.get(b)


# ------------------------------------------------------------------------------------------------ interactions
def _read_sequential(path: str) -> Dict[int, List[int]]:
    out: Dict[int, List[int]] = {}
    with open(path) as f:
        for line in f:
            p = line.split()
            if p:
                out[int(p[0]) - 1] = [int(x) - 1 for x in p[1:]]       # files are 1-based, the dicts 0-based
    return out


def load_interactions(data_path: str, dataset: str) -> Tuple[Dict[int, List[int]], Dict[int, List[int]], Dict[int, List[int]]]:
    """train / valid / test dicts uid -> item ids: `*_dict.npy` (data.py:139-142) or the `sequential_*.txt` they were made from."""
    d = os.path.join(data_path, dataset)
    out = []
    for npy, txt in (("training_dict.npy", "sequential_train.txt"), ("validation_dict.npy", "sequential_valid.txt"),
                     ("testing_dict.npy", "sequential_test.txt")):
        if os.path.exists(os.path.join(d, npy)):
            raw = np.load(os.path.join(d, npy), allow_pickle=True).item()
            out.append({int(k): [int(x) for x in v] for k, v in raw.items()})
        else:
            out.append(_read_sequential(os.path.join(d, txt)))
    return out[0], out[1], out[2]


@dataclass
class TestUser:
    uid: int
    history: List[int]          # item ids, oldest first, already cut to max_his_len
    labels: List[int]           # test items (ground truth)


class SeqRecTestData:
    """`SeqRecDataset(args, mode="test")` (data.py:232-262) at the item-id level."""

    def __init__(self, index: ItemIndex, train: Dict[int, List[int]], valid: Dict[int, List[int]], test: Dict[int, List[int]],
                 max_his_len: int = 20, add_prefix: bool = False, his_sep: str = ", "):
        self.index, self.max_his_len, self.add_prefix, self.his_sep = index, max_his_len, add_prefix, his_sep
        self.users: List[TestUser] = []
        for uid in test:                                         # dict order, as the reference iterates
            items = test[uid]
            if len(items):
                history = list(train.get(uid, [])) + list(valid.get(uid, []))
                if max_his_len > 0:
                    history = history[-max_his_len:]
                self.users.append(TestUser(uid, history, list(items)))

    @classmethod
    def load(cls, data_path: str, dataset: str, index_file: str = ".LCRec-1e-3lr.json", **kw) -> "SeqRecTestData":
        return cls(ItemIndex.load(data_path, dataset, index_file), *load_interactions(data_path, dataset), **kw)

    def __len__(self) -> int:
        return len(self.users)

    def text(self, u: TestUser) -> str:
        """The prompt string the reference tokenises (data.py:240-247)."""
        his = [self.index.item_string(i) for i in u.history]
        if self.add_prefix:
            his = [str(k + 1) + ". " + s for k, s in enumerate(his)]
        # the reference's llama branch joins with a literal ", " whatever --his_sep says (data.py:246); `his_sep` is kept as a
        # constructor argument for flag compatibility only (pinned by tests/golden/harness_golden.json, variant nolimit_sep)
        return SFT_PROMPT.format(INSTRUCTION.format(", ".join(his)))

    def get_all_items(self) -> set:
        return self.index.all_items()

    def get_prefix_allowed_tokens_fn(self, sep: Sequence[int] = synth.RESPONSE_SEP) -> PositionSetConstraint:
        """The mask `inference.py:131` uses: position sets keyed on the tokens after "Response:" (data.py:84-104)."""
        return PositionSetConstraint(self.index.allowed_tokens(), tuple(sep))

    def strict_trie_fn(self, sep: Sequence[int] = synth.RESPONSE_SEP) -> SuffixTrieConstraint:
        """The strict item trie keyed on the generated suffix."""
        return SuffixTrieConstraint(self.index.trie(), tuple(sep))


class CodeTokenEncoder:
    """Tokenizer stand-in for offline runs: BOS, a fixed synthetic template, the history's code tokens separated by a
    comma token, and the real "Response:" separator ids at the end.  Deterministic; lengths 4+5H+6 tokens."""

    def __init__(self, index: ItemIndex, sep: Sequence[int] = synth.RESPONSE_SEP, seed: int = 2025):
        self.index, self.sep = index, tuple(sep)
        banned = set(self.sep)
        tpl = [int(t) for t in synth.hash_randint(16, seed, 3, index.base_vocab) if int(t) not in banned]
        self.head, self.tail, self.comma = tpl[:3], tpl[3:7], 1919      # ',' in the Llama vocabulary

    def __call__(self, data: SeqRecTestData, u: TestUser) -> np.ndarray:
        ids = [1] + self.head
        for k, item in enumerate(u.history):
            if k:
                ids.append(self.comma)
            ids.extend(self.index.item_codes[item])
        ids += self.tail + list(self.sep)
        return np.asarray(ids, dtype=np.int64)


def encode_prompt(data: SeqRecTestData, u: TestUser, tokenizer=None, encoder: Optional[CodeTokenEncoder] = None) -> np.ndarray:
    if tokenizer is not None:
        return np.asarray(tokenizer.encode(data.text(u)), dtype=np.int64)       # collator.py:63-72 at batch size 1
    return (encoder or CodeTokenEncoder(data.index))(data, u)


# ------------------------------------------------------------------------------------------------ metrics
def computeTopNAccuracy(GroundTruth: Sequence[Sequence], predictedIndices: Sequence[Sequence], topN: Sequence[int], rank=None):
    """precision, recall, NDCG, MRR lists (one entry per cut-off, rounded to 4 digits) — utils.py:215-271:
    users with an empty ground truth are skipped; IDCG uses min(len(truth), N) ideal hits."""
    precision, recall, NDCG, MRR = [], [], [], []
    for n in topN:
        s_p = s_r = s_n = s_m = 0.0
        users = 0
        for truth, pred in zip(GroundTruth, predictedIndices):
            if len(truth) == 0:
                continue
            users += 1
            hit, dcg, idcg, mrr = 0, 0.0, 0.0, 0.0
            left = len(truth)
            for j in range(n):
                if pred[j] in truth:
                    dcg += 1.0 / math.log2(j + 2)
                    if mrr == 0.0:
                        mrr = 1.0 / (j + 1.0)
                    hit += 1
                if left > 0:
                    idcg += 1.0 / math.log2(j + 2)
                    left -= 1
            s_p += hit / n
            s_r += hit / len(truth)
            s_n += dcg / idcg if idcg != 0 else 0.0
            s_m += mrr
        precision.append(round(s_p / users, 4))
        recall.append(round(s_r / users, 4))
        NDCG.append(round(s_n / users, 4))
        MRR.append(round(s_m / users, 4))
    return precision, recall, NDCG, MRR


# ------------------------------------------------------------------------------------------------ the inference loop
@dataclass
class InferenceResult:
    uids: List[int] = field(default_factory=list)
    predictions: List[List[int]] = field(default_factory=list)      # K item ids per user, best first (-1 = no such item)
    scores: List[List[float]] = field(default_factory=list)
    labels: List[List[int]] = field(default_factory=list)
    rows: List[Dict[str, float]] = field(default_factory=list)      # per-user timing / acceptance columns (inference.py:152-156,181-187)
    wall_s: float = 0.0

    def metrics(self, index: ItemIndex, topN: Sequence[int] = (1, 5, 10, 20)) -> Dict[str, List[float]]:
        # items that share a code tuple are one item to the generator: compare by code tuple
        truth = [[index.item_codes[i] for i in lab] for lab in self.labels]
        pred = [[index.item_codes.get(i, ()) for i in p] for p in self.predictions]
        k = min(len(p) for p in pred) if pred else 0
        tn = [n for n in topN if n <= k]
        p, r, n, m = computeTopNAccuracy(truth, pred, tn)
        return {"topN": list(tn), "precision": p, "recall": r, "ndcg": n, "mrr": m}

    def timing_mean(self) -> Dict[str, float]:
        """Column means, like the `timing_mean_*.csv` of inference.py:189."""
        if not self.rows:
            return {}
        return {k: float(np.mean([row[k] for row in self.rows])) for k in self.rows[0]}

    def counters(self) -> Dict[str, float]:
        runs = sum(r["n_run"] for r in self.rows)
        acc = sum(r["total_accept_steps"] for r in self.rows)
        beams = len(self.predictions[0]) if self.predictions else 0
        return {"users": len(self.rows), "items_per_s": len(self.rows) * beams / self.wall_s if self.wall_s else 0.0,
                "mean_accept_len": acc / runs if runs else 0.0}


def capacity_for(max_prompt: int, beam: int, draft_beam: int, gamma: int = 4, max_new_tokens: int = 4) -> Dict[str, int]:
    """KV slots / tokens / logit rows a model pair needs for prompts of up to `max_prompt` tokens (`HipLlama(max_slots=, max_tokens=,
    max_logit_rows=)`).  The reference truncates prompts at `cutoff_len` = 512 tokens (code/utils.py:119) and grows its KV cache as it goes
    (beamSD.py:418-429); here the arenas are sized up front: a verification round occupies the prompt, up to gamma draft blocks of DK
    tokens and the K tokens of every earlier round (engine.hip: base + n0 + dl * DK), its packed forward is prompt + gamma * DK tokens and
    reads 1 + gamma * DK logit rows.  Never below the library's defaults (512 / 512 / 384), slots in multiples of 64, at most 2048."""
    dl = max(1, min(gamma, max_new_tokens - 1))
    slots = max_prompt + dl * draft_beam + max_new_tokens * (draft_beam + beam)
    slots = max(512, (slots + 63) // 64 * 64)
    if slots > 2048:
        raise ValueError(f"a prompt of {max_prompt} tokens needs {slots} KV slots; the library's arenas hold at most 2048")
    tokens = max(512, (max_prompt + dl * draft_beam + beam + 63) // 64 * 64)
    rows = max(384, (beam + dl * draft_beam + 63) // 64 * 64)
    return dict(max_slots=slots, max_tokens=tokens, max_logit_rows=rows)


def longest_prompt(data: "SeqRecTestData", L: int = 0, R: Optional[int] = None, tokenizer=None) -> int:
    """Token count of the longest prompt among users [L, R) (what `capacity_for` sizes the arenas from)."""
    enc = CodeTokenEncoder(data.index)
    stop_r = min(len(data), R) if R is not None else len(data)
    return max((len(encode_prompt(data, u, tokenizer, enc)) for u in data.users[L:stop_r]), default=1)


def run_inference(target, draft, data: SeqRecTestData, gamma: int = 4, max_new_tokens: int = 4, L: int = 0, R: Optional[int] = None,
                  users_per_batch: int = 128, prefix_allowed_tokens_fn=None, tokenizer=None, baseline: bool = False,
                  device=None, decoder: str = "bssd") -> InferenceResult:
    """Users [L, R) of the test set through beam-SD (inference.py:123-124,162-187), `users_per_batch` at a time in lock step.
    `baseline=True` also runs `target_generate` per user and records its time and the speed-up columns.
    `decoder="beam"`: the same users through the plain constrained beam search of the target (`target_generate_batch`, beamSD.py:544-595) --
    the same items (the method is lossless), and on MI355X the faster decoder once a lock-step batch is MFMA-bound (bench.py's
    `speedup_curve`): speculation pays while the target forward is a weight stream."""
    import torch
    from .beamSD import BSSD_batch, target_generate, target_generate_batch
    if decoder not in ("bssd", "beam"):
        raise ValueError(f"decoder must be 'bssd' or 'beam', not {decoder!r}")

    fn = prefix_allowed_tokens_fn if prefix_allowed_tokens_fn is not None else data.get_prefix_allowed_tokens_fn()
    dev = device if device is not None else target.device
    enc = CodeTokenEncoder(data.index)
    stop_r = min(len(data), R) if R is not None else len(data)
    sel = data.users[L:stop_r]
    res = InferenceResult()
    prompts = [{"input_ids": torch.from_numpy(encode_prompt(data, u, tokenizer, enc))[None].to(dev)} for u in sel]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for lo in range(0, len(sel), users_per_batch):
        if decoder == "beam":
            outs = target_generate_batch(target, prompts[lo:lo + users_per_batch], max_new_tokens, prefix_allowed_tokens_fn=fn)
            for o in outs:                                  # the CSV columns of the beam-SD path (inference.py:152-156): no draft, no verification
                o.update(draft_time_cost=0.0, target_time_cost=o["device_time_cost"], verify_time_cost=0.0, n_run=0, total_accept_steps=0,
                         total_accept_tokens=0, ave_accept_tokens=0.0)
        else:
            outs = BSSD_batch(target, draft, prompts[lo:lo + users_per_batch], gamma, max_new_tokens, prefix_allowed_tokens_fn=fn)
        for u, pr, o in zip(sel[lo:lo + users_per_batch], prompts[lo:lo + users_per_batch], outs):
            P = pr["input_ids"].shape[1]
            # a user the library ended without a valid beam (status != 0: every beam of a step lost to the id filter, beamSD.py:80-86, where the
            # reference dies) predicts nothing; otherwise the first n_valid beams are the result
            nv = 0 if o.get("status", 0) != 0 else int(o.get("n_valid", o["beam_scores"].shape[0]))
            gen = o["beam_sequence"][:nv, P:].cpu().tolist()
            res.uids.append(u.uid)
            res.predictions.append([data.index.decode(g) for g in gen])
            res.scores.append([float(s) for s in o["beam_scores"][:nv].cpu().tolist()])
            res.labels.append(list(u.labels))
            res.rows.append({"draft_time_cost": o["draft_time_cost"], "target_time_cost": o["target_time_cost"],
                             "verify_time_cost": o["verify_time_cost"], "total_time_cost": o["time_cost"], "n_run": o["n_run"],
                             "total_accept_steps": o["total_accept_steps"], "total_accept_tokens": o["total_accept_tokens"],
                             "ave_accept_tokens": o["ave_accept_tokens"]})
    torch.cuda.synchronize(dev)
    res.wall_s = time.perf_counter() - t0
    if baseline:
        for row, pr in zip(res.rows, prompts):
            tg = target_generate(target, pr, max_new_tokens, prefix_allowed_tokens_fn=fn)
            row["generalBS_time_cost"] = tg["time_cost"]
            row["speedup"] = tg["time_cost"] / row["total_time_cost"] if row["total_time_cost"] else 0.0
            row["overhead"] = (row["total_time_cost"] * max_new_tokens) / (tg["time_cost"] * row["n_run"]) if row["n_run"] else 0.0
    return res


def reduce_metrics(local: InferenceResult, index: ItemIndex, topN: Sequence[int] = (1, 5, 10, 20)) -> Dict[str, List[float]]:
    """Ranking metrics over the users of ALL ranks: per-rank sums travel in one all-gather (users are sharded)."""
    import torch
    import torch.distributed as dist

    m = local.metrics(index, topN)
    n = len(local.predictions)
    vec = torch.tensor([n] + [x * n for key in ("precision", "recall", "ndcg", "mrr") for x in m[key]], dtype=torch.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            vec = vec.cuda()
        parts = [torch.empty_like(vec) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, vec)
        vec = torch.stack(parts).sum(0).cpu()
    tot = float(vec[0])
    k = len(m["topN"])
    vals = (vec[1:] / tot).tolist() if tot else [0.0] * (4 * k)
    return {"topN": m["topN"], "users": int(tot), "precision": vals[:k], "recall": vals[k:2 * k], "ndcg": vals[2 * k:3 * k],
            "mrr": vals[3 * k:]}
