"""Teacher-data generation on the hot-path kernels (reference `code/generate_teacher_data.py:211-244`).

For every training sample the reference stores three things from the teacher (the target model):
  0. `teacher_logits`         — logits at the 5 positions that predict the label tokens (4 codes + EOS), teacher forced;
  1. `teacher_output`         — the K=20 beam sequences (5 tokens) of a constrained beam search under the strict item trie;
  2. `teacher_output_logits`  — the teacher's logits along each of those K sequences, item-token columns only (`32000:`).

The reference gets 1 from HF `generate` one sample at a time and 2 by repeating the prompt K times.  Here 1 is
`target_generate_batch` (users in lock step) and 0 + 2 are ONE packed forward per sample under a tree mask: the prompt
once, then the label tokens and the K beams as branches that each see the prompt and their own prefix — the same
visibility-bitset attention the verification forward of beam-SD uses.
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


@torch.no_grad()
def score_branches(model: HipLlama, prompt: np.ndarray, branches: Sequence[Sequence[int]]) -> List[torch.Tensor]:
    """fp32 logits [len(branch), vocab] for each branch (row j predicts branch[j]) from ONE forward."""
    ids, pos, slots, vis, rows, T = _branch_inputs(prompt, branches, model.max_slots)
    if T > model.max_tokens or T > model.max_slots or T > model.max_logit_rows:
        raise ValueError(f"score_branches: {T} packed tokens exceed the model's capacity")
    dev = model.device
    logits = model.forward_raw(ids.to(dev), pos.to(dev), slots.to(dev), vis.to(dev), T, T)
    return [logits[torch.tensor(r, device=dev)] for r in rows]


@torch.no_grad()
def generate_teacher_data(model: HipLlama, prompts: Sequence[np.ndarray], labels: Sequence[Sequence[int]], strict_trie_fn,
                          beam_size: int = 20, max_new_token: int = 5, users_per_batch: int = 32, item_col0: int = 32000) -> Dict[str, List]:
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
            for p, lab, g in zip(chunk, labels[lo: lo + users_per_batch], gens):
                beams = g["beam_sequence"][:, len(p):]                                   # [K, L]
                sc = score_branches(model, np.asarray(p), [list(lab)] + beams.cpu().tolist())
                out["teacher_logits"].append(sc[0].cpu())
                out["teacher_output"].append(beams.cpu())
                out["teacher_output_logits"].append(torch.stack(sc[1:])[:, :, item_col0:].cpu())
        return out
    finally:
        model.generation_config.num_beams = old
