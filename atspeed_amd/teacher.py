"""Teacher-data generation on the hot-path kernels (reference `code/generate_teacher_data.py:211-244`).

For every training sample the reference stores three things from the teacher (the target model):
  0. `teacher_logits`         — logits at the 5 positions that predict the label tokens (4 codes + EOS), teacher forced;
  1. `teacher_output`         — the K=20 beam sequences (5 tokens) of a constrained beam search under the strict item trie;
  2. `teacher_output_logits`  — the teacher's logits along each of those K sequences, item-token columns only (`32000:`).

The reference gets 1 from HF `generate` one sample at a time and 2 by repeating the prompt K times.  Here 1 is
`target_generate_batch` (users in lock step) and 0 + 2 are ONE packed forward per chunk of samples under a tree mask: per sample
the prompt once, then the label tokens and the K beams as branches that each see the prompt and their own prefix — the same
visibility-bitset attention the verification forward of beam-SD uses, with one segment (and KV arena) per sample.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

from .beamSD import target_generate_batch
from .model import HipLlama


def _branch_inputs(prompt: np.ndarray, branches: Sequence[Sequence[int]], max_slots: int):
    """Packed inputs: prompt rows, then every branch's tokens EXCEPT its last (the logits of a row predict the next token;
    the prompt's last row predicts every branch's first token).  Returns ids, pos, slots, the visibility bitset (built
    directly as 64-bit words: row r of the prompt sees slots [0, r], a branch row sees the prompt and its own prefix) and, per
    branch, the packed rows whose logits predict its tokens."""
    P = len(prompt)
    W = max_slots // 64
    ids, pos, rows, masks = [int(t) for t in prompt], list(range(P)), [], [(1 << (r + 1)) - 1 for r in range(P)]
    full_prompt = (1 << P) - 1
    r = P
    for b in branches:
        br_rows, own = [P - 1], 0
        for j, tok in enumerate(b[:-1]):
            ids.append(int(tok)); pos.append(P + j)
            own |= 1 << r
            masks.append(full_prompt | own)
            br_rows.append(r)
            r += 1
        rows.append(br_rows)
    T = r
    if T > max_slots:
        raise ValueError(f"score_branches: {T} packed tokens exceed max_slots {max_slots}")
    bits = np.frombuffer(b"".join(m.to_bytes(W * 8, "little") for m in masks), dtype=np.uint64).reshape(T, W)
    i32 = lambda x: torch.tensor(x, dtype=torch.int32)
    return i32(ids), i32(pos), torch.arange(T, dtype=torch.int32), torch.from_numpy(bits.view(np.int64).copy()), rows, T


def _check_capacity(model: HipLlama, T: int) -> None:
    if T > model.max_tokens or T > model.max_slots:
        raise ValueError(f"score_branches: {T} packed tokens exceed the model's capacity")


@torch.no_grad()
def score_branches(model: HipLlama, prompt: np.ndarray, branches: Sequence[Sequence[int]]) -> List[torch.Tensor]:
    """fp32 logits [len(branch), vocab] for each branch (row j predicts branch[j]) from ONE forward."""
    return score_branches_batch(model, [prompt], [branches])[0]


def _score_rows(model: HipLlama, prompts, branches):
    """One forward for all samples.  Returns the logits of every scored row, [rows, vocab], and per sample / branch the indices
    into it (row j of a branch predicts its token j)."""
    seqs, index, base = [], [], 0
    for prompt, brs in zip(prompts, branches):
        ids, pos, slots, vis, rows, T = _branch_inputs(np.asarray(prompt), brs, model.max_slots)
        _check_capacity(model, T)
        first = len(prompt) - 1                       # only rows from the prompt's last one on reach the lm_head
        seqs.append((ids, pos, slots, vis, T, T - first))
        index.append([[base + r - first for r in br] for br in rows])
        base += T - first
    return model.forward_raw_batch(seqs, return_all=True), index


@torch.no_grad()
def score_branches_batch(model: HipLlama, prompts: Sequence[np.ndarray], branches: Sequence[Sequence[Sequence[int]]]) -> List[List[torch.Tensor]]:
    """`score_branches` for many samples in ONE forward (`HipLlama.forward_raw_batch`: every sample is a segment with a KV arena of
    its own)."""
    logits, index = _score_rows(model, prompts, branches)
    dev = model.device
    return [[logits[torch.tensor(r, device=dev)] for r in rows] for rows in index]


@torch.no_grad()
def generate_teacher_data(model: HipLlama, prompts: Sequence[np.ndarray], labels: Sequence[Sequence[int]], strict_trie_fn,
                          beam_size: int = 20, max_new_token: int = 5, users_per_batch: int = 32, item_col0: int = 32000,
                          score_batch: int = 32) -> Dict[str, List]:
    """The three teacher tensors per sample, as the reference stores them (lists over samples):
    `teacher_logits` [L, V], `teacher_output` [K, L] (int64), `teacher_output_logits` [K, L, V - item_col0]."""
    old = model.generation_config.num_beams
    model.generation_config.num_beams = beam_size
    try:
        dev = model.device
        out: Dict[str, List] = {"teacher_logits": [], "teacher_output": [], "teacher_output_logits": []}
        for lo in range(0, len(prompts), users_per_batch):
            chunk = prompts[lo: lo + users_per_batch]
            gens = target_generate_batch(model, [{"input_ids": torch.from_numpy(np.asarray(p, dtype=np.int64))[None].to(dev)} for p in chunk],
                                         max_new_token, prefix_allowed_tokens_fn=strict_trie_fn)
            beams = [g["beam_sequence"][:, len(p):].cpu() for p, g in zip(chunk, gens)]      # [K, L] each
            for s0 in range(0, len(chunk), score_batch):
                sl = slice(s0, s0 + score_batch)
                logits, index = _score_rows(model, chunk[sl], [[list(lab)] + b.tolist() for lab, b in
                                                               zip(labels[lo: lo + users_per_batch][sl], beams[sl])])
                # two gathers and two downloads per chunk: [n, L, V] for the labels, [n, K, L, V - item_col0] for the beams
                idx = torch.tensor(index, device=dev)                                          # [n, 1 + K, L]
                lab_logits = logits[idx[:, 0]].cpu()
                beam_logits = logits[:, item_col0:][idx[:, 1:]].cpu()
                for i, b in enumerate(beams[sl]):
                    out["teacher_logits"].append(lab_logits[i])
                    out["teacher_output"].append(b)
                    out["teacher_output_logits"].append(beam_logits[i])
        return out
    finally:
        model.generation_config.num_beams = old
