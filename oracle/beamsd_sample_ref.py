"""ORACLE (test infrastructure, not product code): the SAMPLING branch of the reference's beam speculative
decoding — `code/beamSD.py` one_step_beam_search with `do_sample` (:65-75), verify (:293-321, :332-369), the
bonus draw (:303-309) and the final sort (:529-531, :589-591) — restated on the slot-addressed structures of
`beamsd_ref.py`.  Two random sources:

  * `TorchRng` — `torch.multinomial` / `torch.rand` / `torch.randperm` in the reference's own call order.  With the same
    `torch.manual_seed` this reproduces the reference's outputs EXACTLY (tests/golden/bssd_sample_golden.json, produced by
    running the real reference; the harness supplies `_get_logits_warper`, which transformers 5.x no longer has).
  * `HashRng`  — the counter-based generator the HIP kernels use (`atspeed_amd/synth.py:hash_u32`): a draw without
    replacement is a top-n of `log w + Gumbel` (the Plackett-Luce law of `torch.multinomial`'s sequential draws), a
    uniform is `((h >> 9) + 0.5) * 2^-23`, a random subset is the n smallest hashes.  Same distribution as TorchRng
    (checked statistically in tests/test_oracle_pins.py), bit-comparable with the device (tests/test_bssd_gpu.py).

Deviation, documented: when a rejected step leaves no residual mass (`new_probs.sum() == 0`, :354-361) the reference
resamples uniformly over the whole flattened vocabulary (disallowed tokens included, scores -inf); here the remaining
draws come from the target distribution itself.  It needs p <= q on every non-accepted candidate, i.e. p == q.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from atspeed_amd import synth
from .beamsd_ref import (EOS_ID, LLAMA_VOCAB, StepInputs, _causal_inputs, _pad_vis, constrain, target_beam_search)
from .llama_ref import RefLlama

# purposes of a random stream (shared with the device: atspeed_amd/csrc/scan.hip)
P_STEP, P_ACCEPT, P_PERM, P_RESID, P_BONUS = 1, 2, 3, 4, 5


class TorchRng:
    """The reference's calls, in its order, on torch's global generator."""
    exact = True

    def begin(self, purpose: int, rnd: int, step: int, model_tag: int = 0):
        pass

    def multinomial(self, weights: torch.Tensor, n: int) -> torch.Tensor:
        return torch.multinomial(weights, num_samples=n)

    def uniform(self, n: int) -> torch.Tensor:
        return torch.rand(n, dtype=torch.float32)

    def subset(self, n_items: int, n: int) -> torch.Tensor:
        return torch.randperm(n_items)[:n]


class HashRng:
    """Counter-based: every draw is a pure function of (seed, purpose, round, step, model, element id)."""
    exact = False

    def __init__(self, seed: int):
        self.seed = int(seed) & 0xFFFFFFFF
        self.sub = 0
        self.min_margin = float("inf")      # smallest gap between a selected and a rejected key (parity diagnostics)

    def begin(self, purpose: int, rnd: int, step: int, model_tag: int = 0):
        ctr = (purpose & 0xFF) | ((rnd & 0xFF) << 8) | ((step & 0xFF) << 16) | ((model_tag & 0xFF) << 24)
        self.sub = int(synth.hash_u32(np.array([ctr], dtype=np.uint64), self.seed)[0])

    def _u01(self, ids: np.ndarray) -> np.ndarray:
        h = synth.hash_u32(ids.astype(np.uint64), self.sub)
        return ((h >> np.uint64(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)

    def gumbel(self, ids: np.ndarray) -> np.ndarray:
        u = self._u01(ids)
        return (-np.log(-np.log(u, dtype=np.float32), dtype=np.float32)).astype(np.float32)

    def multinomial_log(self, logw: torch.Tensor, n: int) -> torch.Tensor:
        """n indices without replacement with probability proportional to exp(logw): top-n of logw + Gumbel(id)."""
        lw = logw.numpy().astype(np.float32)
        ids = np.nonzero(np.isfinite(lw))[0]
        keys = lw[ids] + self.gumbel(ids)
        order = np.lexsort((ids, -keys))                 # key desc, id asc
        if len(order) > n:
            self.min_margin = min(self.min_margin, float(keys[order[n - 1]] - keys[order[n]]))
        return torch.from_numpy(ids[order[:n]].astype(np.int64))

    def multinomial(self, weights: torch.Tensor, n: int) -> torch.Tensor:
        return self.multinomial_log(torch.log(weights), n)

    def uniform_ids(self, ids: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(self._u01(ids))

    def subset_ids(self, ids: np.ndarray, n: int) -> torch.Tensor:
        """positions (into `ids`) of the n smallest hashes, ties by position."""
        h = synth.hash_u32(ids.astype(np.uint64), self.sub)
        order = np.lexsort((np.arange(len(ids)), h))
        return torch.from_numpy(order[:n].astype(np.int64))


def _tempered(logits: torch.Tensor, seqs: torch.Tensor, beam_size: int, fn, temperature: float) -> torch.Tensor:
    """log-softmax over the full vocabulary (:58,:285), prefix mask (:60-64,:286-291), temperature warper (:65-66,:293-294)."""
    logp = torch.log_softmax(logits.to(torch.float32), dim=-1)
    if fn is not None:
        if logits.shape[0] == 1 and beam_size != 1:
            logp = constrain(seqs[:1], logp, fn)
        else:
            logp = constrain(seqs, logp, fn)
    return logp / temperature


def one_step_sample(model: RefLlama, inp: StepInputs, beam_size: int, beam_scores: torch.Tensor, beam_seq: torch.Tensor,
                    fn, temperature: float, rng) -> Dict:
    """beamSD.py:40-106 with do_sample: multinomial over softmax of the flattened tempered scores (:72-75)."""
    n = len(beam_scores)
    logits = model.forward(inp.ids, inp.pos, inp.slots, inp.vis, n_logit_rows=n)
    V = logits.shape[-1]
    flat = (_tempered(logits, beam_seq, beam_size, fn, temperature) + beam_scores.to(torch.float32)[:, None]).reshape(-1)
    probs = torch.softmax(flat, dim=-1)                                           # :72
    idx = rng.multinomial_log(flat, beam_size) if not rng.exact else rng.multinomial(probs, beam_size)   # :73
    scores = flat[idx]                                                            # :74
    parents, toks = idx // V, idx % V
    if fn is not None:                                                            # :80-86
        keep = ((toks >= LLAMA_VOCAB) | (toks == EOS_ID)) & torch.isfinite(scores)
        idx, scores, parents, toks = idx[keep], scores[keep], parents[keep], toks[keep]
    m = len(toks)
    new_seq = torch.cat((beam_seq[parents], toks[:, None]), dim=-1)
    S = inp.vis.shape[1]
    vis = torch.cat((inp.vis[-n:][parents], torch.eye(m, dtype=torch.bool)), dim=1)
    nxt = StepInputs(ids=toks.clone(), pos=(inp.pos[-1:] + 1).repeat(m), slots=torch.arange(S, S + m), vis=vis)
    return {"seq_tokens": idx, "beam_sequence": new_seq, "beam_scores": scores, "beam_indices": parents, "beam_tokens": toks,
            "next_inputs": nxt, "probs": probs, "flat": flat}


def draft_beam_search_sample(model, inp, draft_len, beam_size, beam_scores, beam_seq, fn, temperature, rng, rnd) -> Dict:
    out = {"step_len": [len(beam_scores)], "step_seq_tokens": [], "step_beam_sequence": [beam_seq], "step_beam_indices": [],
           "step_beam_tokens": [], "step_inputs": [], "step_scores": [], "step_probs": [], "step_flat": []}
    for i in range(draft_len):
        rng.begin(P_STEP, rnd, i, 1)
        o = one_step_sample(model, inp, beam_size, beam_scores, beam_seq, fn, temperature, rng)
        inp, beam_scores, beam_seq = o["next_inputs"], o["beam_scores"], o["beam_sequence"]
        out["step_len"].append(len(beam_scores))
        out["step_beam_sequence"].append(beam_seq)
        out["step_seq_tokens"].append(o["seq_tokens"])
        out["step_beam_indices"].append(o["beam_indices"])
        out["step_beam_tokens"].append(o["beam_tokens"])
        out["step_inputs"].append(inp)
        out["step_scores"].append(beam_scores)
        out["step_probs"].append(o["probs"])
        out["step_flat"].append(o["flat"])
    out["beam_scores"] = beam_scores
    return out


def verify_sample(tin: StepInputs, draft: Dict, target: Dict, beam_size: int, beam_scores: torch.Tensor, beam_seq: torch.Tensor,
                  fn, temperature: float, rng, rnd: int) -> Dict:
    """beamSD.py:242-456 with do_sample (:293-321 distributions, :332-369 accept / resample)."""
    draft_len = len(draft["step_beam_indices"])
    step_len = draft["step_len"]
    scores_all = target["next_token_scores"]
    V = scores_all.shape[-1]
    packed: StepInputs = target["packed"]
    n0 = len(tin.ids)
    n_matches = 0
    lo, hi = 0, step_len[0]
    hit = None
    trace = []
    for i in range(draft_len + 1):
        rows = scores_all[lo:hi]
        if n_matches != draft_len:
            lo, hi = hi, hi + step_len[i + 1]
        if i > 0:
            rows = rows[hit]                                                      # :284
        seqs = draft["step_beam_sequence"][i][hit] if i > 0 else draft["step_beam_sequence"][i]
        bs = (_tempered(rows, seqs, beam_size, fn, temperature) + beam_scores.to(torch.float32)[:, None]).reshape(-1)   # :297-298
        if n_matches == draft_len:                                                # :303-309 bonus draw from the target
            rng.begin(P_BONUS, rnd, i, 0)
            if rng.exact:
                nxt = rng.multinomial(torch.softmax(bs, -1), beam_size)
                beam_scores = bs[nxt]
                parents, toks = nxt // V, nxt % V
                if i > 0:
                    parents = hit[parents]
            else:                                                                 # element ids = flat ids in the draft's beam space
                if i > 0:
                    tbs = torch.full((step_len[i], V), float("-inf"), dtype=torch.float32)
                    tbs[hit] = bs.view(-1, V)
                    bs = tbs.reshape(-1)
                nxt = rng.multinomial_log(bs, beam_size)
                beam_scores = bs[nxt]
                parents, toks = nxt // V, nxt % V
            trace.append({"bonus": True, "ids": (parents * V + toks).tolist()})
            break
        if i > 0:                                                                 # :311-321 into the draft's beam space
            tbs = torch.full((step_len[i], V), float("-inf"), dtype=torch.float32)
            tbs[hit] = bs.view(-1, V)
            bs = tbs.reshape(-1)
        probs = torch.softmax(bs, dim=-1)
        probs[torch.isnan(probs)] = 0
        dprobs = draft["step_probs"][i]
        d_ids = draft["step_seq_tokens"][i]                                       # :334-338
        p_i, q_i = probs[d_ids], dprobs[d_ids]
        rng.begin(P_ACCEPT, rnd, i, 0)
        r = rng.uniform(len(d_ids)) if rng.exact else rng.uniform_ids(np.arange(len(d_ids)))
        acc = (r <= p_i / q_i) if rng.exact else (r * q_i <= p_i)                 # same test; the device avoids the division
        acc_tokens = d_ids[acc]
        n_acc = int(acc.sum())
        if n_acc >= beam_size:                                                    # :341-350
            n_matches += 1
            rng.begin(P_PERM, rnd, i, 0)
            if rng.exact:
                sel = rng.subset(n_acc, beam_size)
            else:
                sel = rng.subset_ids(torch.nonzero(acc).reshape(-1).numpy(), beam_size)
            seq_tokens = torch.sort(acc_tokens[sel]).values
            pos_of = {int(d): k for k, d in enumerate(d_ids.tolist())}
            hit = torch.tensor([pos_of[int(y)] for y in seq_tokens.tolist()], dtype=torch.long)
            beam_scores = bs[seq_tokens]
            parents, toks = seq_tokens // V, seq_tokens % V
            trace.append({"accepted": n_acc, "ids": seq_tokens.tolist()})
        else:                                                                     # :351-369
            newp = torch.clamp(probs - dprobs, min=0)
            newp[acc_tokens] = 0
            if float(newp.sum()) == 0.0:
                newp = probs.clone()
                newp[acc_tokens] = 0
            rng.begin(P_RESID, rnd, i, 0)
            nxt = rng.multinomial(newp / newp.sum() if rng.exact else newp, beam_size - n_acc)
            seq_tokens = torch.sort(torch.cat((acc_tokens, nxt))).values
            beam_scores = bs[seq_tokens]
            parents, toks = seq_tokens // V, seq_tokens % V
            trace.append({"accepted": n_acc, "resampled": beam_size - n_acc, "ids": seq_tokens.tolist()})
            break

    new_seq = torch.cat((draft["step_beam_sequence"][n_matches][parents], toks[:, None]), dim=-1)   # :383
    blk_lo = n0 - step_len[0] + sum(step_len[:n_matches])
    blk_rows = packed.vis[blk_lo: blk_lo + step_len[n_matches]]
    base = int(packed.slots[blk_lo + step_len[n_matches] - 1]) + 1
    m = len(toks)
    vis = torch.cat((_pad_vis(blk_rows, base)[parents], torch.eye(m, dtype=torch.bool)), dim=1)
    pos_next = packed.pos[blk_lo] + 1
    nxt_t = StepInputs(ids=toks.clone(), pos=pos_next.repeat(m), slots=torch.arange(base, base + m), vis=vis)
    nxt_d = nxt_t
    if n_matches == draft_len and draft_len > 0:
        last = draft["step_inputs"][draft_len - 1]
        width = base + m
        nxt_d = StepInputs(ids=torch.cat((last.ids, toks)), pos=torch.cat((last.pos, nxt_t.pos)),
                           slots=torch.cat((last.slots, nxt_t.slots)), vis=torch.cat((_pad_vis(last.vis, width), vis), dim=0))
    return {"n_matches": n_matches, "beam_sequence": new_seq, "beam_scores": beam_scores, "target_inputs": nxt_t,
            "draft_inputs": nxt_d, "trace": trace}


def _final_sort(beam_seq: torch.Tensor, beam_scores: torch.Tensor):
    s = torch.sort(beam_scores, descending=True, stable=True)                     # :529-531
    return beam_seq[s.indices], s.values


@torch.no_grad()
def BSSD_sample(target: RefLlama, draft: RefLlama, input_ids, gamma: int, max_new_tokens: int, beam_size: int,
                draft_beam_size: int, fn: Optional[Callable], temperature: float, rng) -> Dict:
    """beamSD.py:458-542 with `generation_config.do_sample = True`."""
    ids = torch.as_tensor(np.asarray(input_ids), dtype=torch.long).reshape(-1)
    cur_len = len(ids)
    max_len = cur_len + max_new_tokens
    tin = din = _causal_inputs(ids)
    beam_scores = torch.zeros(1, dtype=torch.float32)
    beam_seq = ids[None, :].repeat(beam_size, 1)
    accept_steps: List[int] = []
    rounds = []
    while cur_len < max_len:
        rnd = len(accept_steps)
        draft_len = min(gamma, max_len - cur_len - 1)
        if draft_len == 0:
            rng.begin(P_STEP, rnd, 0, 0)
            o = one_step_sample(target, tin, beam_size, beam_scores, beam_seq, fn, temperature, rng)
            beam_seq, beam_scores = o["beam_sequence"], o["beam_scores"]
            break
        d = draft_beam_search_sample(draft, din, draft_len, draft_beam_size, beam_scores, beam_seq, fn, temperature, rng, rnd)
        t = target_beam_search(target, tin, d)
        v = verify_sample(tin, d, t, beam_size, beam_scores, beam_seq, fn, temperature, rng, rnd)
        rounds.append({"draft_len": draft_len, "n_matches": v["n_matches"], "step_len": list(d["step_len"]),
                       "draft_ids": [x.tolist() for x in d["step_seq_tokens"]], "verify": v["trace"]})
        beam_seq, beam_scores = v["beam_sequence"], v["beam_scores"]
        tin, din = v["target_inputs"], v["draft_inputs"]
        cur_len += v["n_matches"] + 1
        accept_steps.append(v["n_matches"])
    beam_seq, beam_scores = _final_sort(beam_seq, beam_scores)
    n_run, total = len(accept_steps), sum(accept_steps)
    return {"beam_sequence": beam_seq, "beam_scores": beam_scores, "n_run": n_run, "total_accept_steps": total,
            "total_accept_tokens": total * beam_size, "ave_accept_tokens": (total * beam_size / n_run) if n_run else 0.0,
            "rounds": rounds}


@torch.no_grad()
def target_generate_sample(model: RefLlama, input_ids, max_new_tokens: int, beam_size: int, fn, temperature: float, rng) -> Dict:
    """beamSD.py:544-595 with do_sample."""
    ids = torch.as_tensor(np.asarray(input_ids), dtype=torch.long).reshape(-1)
    inp = _causal_inputs(ids)
    beam_scores = torch.zeros(1, dtype=torch.float32)
    beam_seq = ids[None, :].repeat(beam_size, 1)
    for g in range(max_new_tokens):
        rng.begin(P_STEP, g, 0, 0)
        o = one_step_sample(model, inp, beam_size, beam_scores, beam_seq, fn, temperature, rng)
        inp, beam_scores, beam_seq = o["next_inputs"], o["beam_scores"], o["beam_sequence"]
    beam_seq, beam_scores = _final_sort(beam_seq, beam_scores)
    return {"beam_sequence": beam_seq, "beam_scores": beam_scores}
