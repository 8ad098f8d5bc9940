"""ORACLE (test infrastructure, not product code): fp32 CPU restatement of the
reference's beam speculative decoding, greedy branch — `code/beamSD.py`:
one_step_beam_search (:40-106), _draft_beam_search (:108-179), _target_beam_search
(:190-232), verify (:242-456, greedy lines), BSSD (:458-542) and target_generate
(:544-595).  The sampling branch (:293-321, :332-369) is out of scope (SURVEY.md 8a V').

Formulation differences from the reference (results identical; checked by the golden
fixtures generated from the imported reference, tests/golden/gen_golden.py):
  * the KV cache is slot-addressed and additive masks are boolean visibility rows over
    slots, so "truncate the cache" (:418-429) becomes "reuse slots from `base` on";
    non-first rounds do NOT keep the K masked garbage slots the reference over-retains
    (`:259,387-392`) — they are invisible in the reference, absent here;
  * top-k ties (unspecified in `torch.topk`) are broken by ascending flat index, and a
    pick whose score is -inf (fewer finite candidates than beams) is
    never a beam.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from .llama_ref import RefLlama

LLAMA_VOCAB = 32000   # beamSD.py:81
EOS_ID = 2            # beamSD.py:81

# Decision margins (test aid): when a list is installed here, every top-k appends the smallest gap between neighbours
# among its k + 1 best finite scores.  A comparison with an engine that sums in another order (or in bf16) is only
# meaningful where this margin is above that engine's rounding noise.
MARGINS: Optional[List[float]] = None
# Type of the scores (log-softmax, beam scores).  fp32 is the reference's (HF 4.41 upcasts the logits, beamSD.py:58 works on them as they come);
# tests/test_fulldims_gpu.py sets float64 around runs of the ARBITER (RefLlama(dtype=torch.float64)): the same search in double precision.
SCORE_DTYPE = torch.float32


@dataclass
class StepInputs:
    """One forward's inputs: T tokens written at cache `slots`, visibility over [0, S)."""
    ids: torch.Tensor      # [T] long
    pos: torch.Tensor      # [T] long
    slots: torch.Tensor    # [T] long
    vis: torch.Tensor      # [T, S] bool, S = slots.max() + 1


def _pad_vis(v: torch.Tensor, width: int) -> torch.Tensor:
    if v.shape[1] >= width:
        return v[:, :width]
    return torch.cat((v, torch.zeros(v.shape[0], width - v.shape[1], dtype=torch.bool)), dim=1)


def topk_desc_stable(flat: torch.Tensor, k: int):
    """Top-k by (score desc, flat index asc): the tie-break this build defines."""
    s = torch.sort(flat, descending=True, stable=True)
    k = min(k, flat.numel())
    return s.values[:k], s.indices[:k]


def constrain(seqs: torch.Tensor, logp: torch.Tensor, fn: Callable) -> torch.Tensor:
    """transformers PrefixConstrainedLogitsProcessor.__call__ with _num_beams = rows
    (poked at beamSD.py:56,281): every row belongs to batch 0."""
    out = torch.full_like(logp, float("-inf"))
    for r in range(logp.shape[0]):
        allowed = fn(0, seqs[r])
        if len(allowed) == 0:
            raise ValueError("`prefix_allowed_tokens_fn` returned an empty list for batch ID 0.")
        idx = torch.as_tensor(list(allowed), dtype=torch.long)
        out[r, idx] = logp[r, idx]
    return out


def expand_and_prune(logits: torch.Tensor, beam_scores: torch.Tensor, beam_seq: torch.Tensor,
                     beam_size: int, fn: Optional[Callable], drop_disallowed: bool = True, procs=()):
    """beamSD.py:57-86 given the logits rows: log-softmax over the FULL vocab, mask,
    add beam scores, flatten, top-k, split into (parent, token), drop disallowed picks."""
    n, V = logits.shape
    logp = torch.log_softmax(logits.to(SCORE_DTYPE), dim=-1)                      # :58
    if fn is not None:
        if n == 1 and beam_size != 1:                                             # :61-62
            logp = constrain(beam_seq[:1], logp, fn)
        else:                                                                     # :64
            logp = constrain(beam_seq, logp, fn)
    # extra logits processors (beamSD.py:469-478: HF appends the caller's LogitsProcessorList after its own prefix processor)
    for proc in procs:
        rows = beam_seq[:1] if (n == 1 and beam_size != 1) else beam_seq          # :61-62: K copies in, row 0 out
        logp = proc(rows, logp)
    flat = (logp + beam_scores.to(SCORE_DTYPE)[:, None]).reshape(-1)              # :69-70
    scores, idx = topk_desc_stable(flat, beam_size)                               # :76
    if MARGINS is not None:
        top = topk_desc_stable(flat, beam_size + 1)[0]
        top = top[torch.isfinite(top)]
        if len(top) > 1:
            MARGINS.append(float((top[:-1] - top[1:]).min()))
    parents, toks = idx // V, idx % V                                             # :77-78
    if (fn is not None or len(procs) != 0) and drop_disallowed:                   # :80-86 (one_step only; `len(logits_processor) != 0`)
        keep = (toks >= LLAMA_VOCAB) | (toks == EOS_ID)
        # fewer than beam_size finite candidates: WHICH -inf entries torch.topk returns is
        # unspecified (they are ties); the reference keeps those whose token id happens to
        # pass the id test.  This build defines a -inf pick as "not a beam".
        keep &= torch.isfinite(scores)
        idx, scores, parents, toks = idx[keep], scores[keep], parents[keep], toks[keep]
    return idx, scores, parents, toks


def one_step(model: RefLlama, inp: StepInputs, beam_size: int, beam_scores: torch.Tensor,
             beam_seq: torch.Tensor, fn: Optional[Callable], procs=()) -> Dict:
    """beamSD.py:40-106."""
    n = len(beam_scores)
    logits = model.forward(inp.ids, inp.pos, inp.slots, inp.vis, n_logit_rows=n)  # :52,57
    idx, scores, parents, toks = expand_and_prune(logits, beam_scores, beam_seq, beam_size, fn, procs=procs)
    m = len(toks)
    new_seq = torch.cat((beam_seq[parents], toks[:, None]), dim=-1)               # :87
    S = inp.vis.shape[1]
    parent_rows = inp.vis[-n:][parents]                                           # :89
    vis = torch.cat((parent_rows, torch.eye(m, dtype=torch.bool)), dim=1)
    nxt = StepInputs(ids=toks.clone(), pos=(inp.pos[-1:] + 1).repeat(m),         # :91
                     slots=torch.arange(S, S + m), vis=vis)
    return {"seq_tokens": idx, "beam_sequence": new_seq, "beam_scores": scores,
            "beam_indices": parents, "beam_tokens": toks, "next_inputs": nxt, "logits": logits}


def draft_beam_search(model: RefLlama, inp: StepInputs, draft_len: int, beam_size: int,
                      beam_scores: torch.Tensor, beam_seq: torch.Tensor, fn, procs=()) -> Dict:
    """beamSD.py:108-179."""
    out = {"step_len": [len(beam_scores)], "step_seq_tokens": [], "step_beam_sequence": [beam_seq],
           "step_beam_indices": [], "step_beam_tokens": [], "step_inputs": [], "step_scores": []}
    for _ in range(draft_len):                                                    # :136
        o = one_step(model, inp, beam_size, beam_scores, beam_seq, fn, procs)
        inp, beam_scores, beam_seq = o["next_inputs"], o["beam_scores"], o["beam_sequence"]
        out["step_len"].append(len(beam_scores))
        out["step_beam_sequence"].append(beam_seq)
        out["step_seq_tokens"].append(o["seq_tokens"])
        out["step_beam_indices"].append(o["beam_indices"])
        out["step_beam_tokens"].append(o["beam_tokens"])
        out["step_inputs"].append(inp)
        out["step_scores"].append(beam_scores)
    out["beam_scores"] = beam_scores
    return out


def target_beam_search(model: RefLlama, inp: StepInputs, draft: Dict) -> Dict:
    """beamSD.py:190-232: ONE forward over round inputs ++ every draft step's tokens."""
    blocks = [inp] + draft["step_inputs"]
    width = max(b.vis.shape[1] for b in blocks)                                   # :205
    packed = StepInputs(ids=torch.cat([b.ids for b in blocks]),                   # :203
                        pos=torch.cat([b.pos for b in blocks]),                   # :210-211
                        slots=torch.cat([b.slots for b in blocks]),
                        vis=torch.cat([_pad_vis(b.vis, width) for b in blocks], dim=0))  # :206-209
    rows = sum(draft["step_len"])                                                 # :224
    logits = model.forward(packed.ids, packed.pos, packed.slots, packed.vis, n_logit_rows=rows)
    return {"next_token_scores": logits, "packed": packed}


def verify(tin: StepInputs, draft: Dict, target: Dict, beam_size: int,
           beam_scores: torch.Tensor, beam_seq: torch.Tensor, fn, procs=()) -> Dict:
    """beamSD.py:242-456, greedy branch."""
    draft_len = len(draft["step_beam_indices"])
    step_len = draft["step_len"]
    scores_all = target["next_token_scores"]
    V = scores_all.shape[-1]
    packed: StepInputs = target["packed"]
    n0 = len(tin.ids)
    n_matches = 0
    lo, hi = 0, step_len[0]                                                       # :277
    hit = hit4 = None
    trace = []
    for i in range(draft_len + 1):                                                # :278
        rows = scores_all[lo:hi]                                                  # :279
        if n_matches != draft_len:                                                # :282-283
            lo, hi = hi, hi + step_len[i + 1]
        if i > 0:
            rows = rows[hit]                                                      # :284
        seqs = draft["step_beam_sequence"][i][hit] if i > 0 else draft["step_beam_sequence"][i]  # :290
        if i > 0:
            beam_scores = beam_scores[hit4]                                       # :295-296
        # round 1, i == 0: one logit row masked with the prompt (:287-288); expand_and_prune
        # takes row 0 of seqs (= the round's beam_sequence, K copies of the prompt) for it
        idx, beam_scores, parents, toks = expand_and_prune(rows, beam_scores, seqs, beam_size, fn,
                                                           drop_disallowed=False, procs=procs)
        if i > 0:                                                                 # :326-328
            parents = hit[parents]
            idx = parents * V + toks
        trace.append({"target_ids": idx.tolist(), "target_scores": beam_scores.tolist()})
        if n_matches == draft_len:                                                # :329-330
            break
        d_ids = draft["step_seq_tokens"][i].tolist()                              # :371
        t_ids = idx.tolist()                                                      # :372
        t_set = set(t_ids)
        hit = torch.tensor([k for k, d in enumerate(d_ids) if d in t_set], dtype=torch.long)   # :373-374
        pos_of = {d: k for k, d in enumerate(d_ids)}
        found = torch.tensor([pos_of[y] for y in t_ids if y in pos_of], dtype=torch.long)      # :375
        hit4 = torch.sort(found, stable=True).indices                             # :376
        if len(hit) == beam_size:                                                 # :377-380
            n_matches += 1
        else:
            break

    new_seq = torch.cat((draft["step_beam_sequence"][n_matches][parents], toks[:, None]), dim=-1)  # :383
    # rows of block n_matches inside the packed target input (:384-398)
    blk_lo = n0 - step_len[0] + sum(step_len[:n_matches])
    blk_rows = packed.vis[blk_lo: blk_lo + step_len[n_matches]]
    base = int(packed.slots[blk_lo + step_len[n_matches] - 1]) + 1               # compact (quirk 2)
    m = len(toks)
    vis = torch.cat((_pad_vis(blk_rows, base)[parents], torch.eye(m, dtype=torch.bool)), dim=1)    # :396-400
    pos_next = packed.pos[blk_lo] + 1                                             # :401
    nxt_t = StepInputs(ids=toks.clone(), pos=pos_next.repeat(m), slots=torch.arange(base, base + m), vis=vis)
    nxt_d = nxt_t
    if n_matches == draft_len and draft_len > 0:                                  # :402-416
        # the draft never ran on its own last step's tokens: feed them together with the new beams
        last = draft["step_inputs"][draft_len - 1]
        width = base + m
        vis_d = torch.cat((_pad_vis(last.vis, width), vis), dim=0)
        nxt_d = StepInputs(ids=torch.cat((last.ids, toks)), pos=torch.cat((last.pos, nxt_t.pos)),
                           slots=torch.cat((last.slots, nxt_t.slots)), vis=vis_d)
    return {"n_matches": n_matches, "beam_sequence": new_seq, "beam_scores": beam_scores,
            "target_inputs": nxt_t, "draft_inputs": nxt_d, "trace": trace}


def _causal_inputs(ids: torch.Tensor) -> StepInputs:
    n = len(ids)                                                                  # :487-495
    return StepInputs(ids=ids.clone(), pos=torch.arange(n), slots=torch.arange(n),
                      vis=torch.tril(torch.ones(n, n, dtype=torch.bool)))


@torch.no_grad()
def BSSD(target: RefLlama, draft: RefLlama, input_ids, gamma: int, max_new_tokens: int,
         beam_size: int, draft_beam_size: int, fn: Optional[Callable] = None, procs=()) -> Dict:
    """beamSD.py:458-542.  `input_ids` is the [P] prompt (the reference reads batch row 0).  `fn` None = no mask; `procs` = extra logits
    processors `(input_ids [n, len], scores [n, V]) -> scores` (the LogitsProcessorList argument, :465,469-478)."""
    ids = torch.as_tensor(np.asarray(input_ids), dtype=torch.long).reshape(-1)
    cur_len = len(ids)
    max_len = cur_len + max_new_tokens
    tin = din = _causal_inputs(ids)
    beam_scores = torch.zeros(1, dtype=SCORE_DTYPE)                               # :498
    beam_seq = ids[None, :].repeat(beam_size, 1)                                  # :499
    accept_steps: List[int] = []
    rounds = []
    while cur_len < max_len:                                                      # :503
        draft_len = min(gamma, max_len - cur_len - 1)                             # :504
        if draft_len == 0:                                                        # :505-509
            o = one_step(target, tin, beam_size, beam_scores, beam_seq, fn, procs)
            beam_seq, beam_scores = o["beam_sequence"], o["beam_scores"]
            break
        d = draft_beam_search(draft, din, draft_len, draft_beam_size, beam_scores, beam_seq, fn, procs)   # :511
        t = target_beam_search(target, tin, d)                                                     # :513
        v = verify(tin, d, t, beam_size, beam_scores, beam_seq, fn, procs)                         # :515
        rounds.append({"draft_len": draft_len, "n_matches": v["n_matches"], "step_len": list(d["step_len"]),
                       "draft_ids": [x.tolist() for x in d["step_seq_tokens"]],
                       "draft_scores": [x.tolist() for x in d["step_scores"]],
                       "verify": v["trace"]})
        beam_seq, beam_scores = v["beam_sequence"], v["beam_scores"]
        tin, din = v["target_inputs"], v["draft_inputs"]
        cur_len += v["n_matches"] + 1                                             # :522
        accept_steps.append(v["n_matches"])
    n_run = len(accept_steps)
    total = sum(accept_steps)
    return {"beam_sequence": beam_seq, "beam_scores": beam_scores, "n_run": n_run,
            "total_accept_steps": total, "total_accept_tokens": total * beam_size,
            "ave_accept_tokens": (total * beam_size / n_run) if n_run else 0.0,    # :538 (ZeroDivision in the reference when n_run == 0)
            "rounds": rounds}


@torch.no_grad()
def target_generate(model: RefLlama, input_ids, max_new_tokens: int, beam_size: int,
                    fn: Optional[Callable] = None, procs=()) -> Dict:
    """beamSD.py:544-595: plain constrained beam search on the target."""
    ids = torch.as_tensor(np.asarray(input_ids), dtype=torch.long).reshape(-1)
    inp = _causal_inputs(ids)
    beam_scores = torch.zeros(1, dtype=SCORE_DTYPE)
    beam_seq = ids[None, :].repeat(beam_size, 1)
    steps = []
    for _ in range(max_new_tokens):                                               # :579-588
        o = one_step(model, inp, beam_size, beam_scores, beam_seq, fn, procs)
        inp, beam_scores, beam_seq = o["next_inputs"], o["beam_scores"], o["beam_sequence"]
        steps.append({"ids": o["seq_tokens"].tolist(), "scores": beam_scores.tolist()})
    return {"beam_sequence": beam_seq, "beam_scores": beam_scores, "steps": steps}
