"""Decision-level replay (test infrastructure): the fp32 engine judges every top-K / top-DK decision the bf16 engine made.

The bf16 engine runs BSSD for a lock-step batch with `trace_decisions=True`; `last_decisions()` returns, per user and round, the beams
the round started from, the draft's blocks, the target's picks at every verify step and n_matches.  For each of those decisions the
fp32 engine (pinned to the oracle bit for bit at the same dims, tests/test_fulldims_gpu.py) scores EVERY candidate of the decision in
the bf16 engine's own state -- same parents, same allowed children (beamSD.py:58-78,279-298) -- from one packed tree forward per round
and model.  A membership "candidate x is among the n best" is CLEAR when x's fp32 score is further from the n / n+1 boundary than

    bound = C_NOISE * max |bf16 score - fp32 score| over the chosen items of every user at that (model, depth)      (C_NOISE = 2)

(two items can only swap when their fp32 gap is at most the sum of their two bf16 errors; the error of an item the bf16 engine did NOT
keep cannot be measured -- its score is gone -- so the level comes from all kept items of the batch at the same depth, `evaluate`).  The assertion of the tests is: every clear
membership is the same in both engines.  Because the judge looks at the bf16 engine's own trajectory, one early near-tie does not
turn every later decision of that user into a "difference".
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from atspeed_amd.model import HipLlama, vis_bits_from_bool

C_NOISE = 2.0


class TreeJudge:
    """Full-vocabulary log-probabilities of arbitrary generated prefixes under one model, from ONE packed forward: the prompt once,
    every distinct prefix one token (tree mask: a node sees the prompt, its ancestors and itself)."""

    def __init__(self, model: HipLlama):
        self.m = model

    def run(self, prompt: np.ndarray, seqs: Sequence[Sequence[int]]):
        """-> (row, logp): row[tuple(prefix)] = row of `logp` (fp32 [n_nodes + 1, V], on the device) holding the next-token
        log-softmax after that prefix; row[()] = after the prompt alone."""
        P = len(prompt)
        row: Dict[Tuple[int, ...], int] = {(): 0}
        nodes: List[Tuple[int, ...]] = []
        for sq in seqs:
            for j in range(1, len(sq) + 1):
                pre = tuple(int(t) for t in sq[:j])
                if pre not in row:
                    row[pre] = 1 + len(nodes)
                    nodes.append(pre)
        T = P + len(nodes)
        assert T <= self.m.max_tokens and T <= self.m.max_slots and 1 + len(nodes) <= self.m.max_logit_rows, (T, len(nodes))
        ids = np.concatenate([np.asarray(prompt, np.int64), np.asarray([n[-1] for n in nodes], np.int64)]).astype(np.int32)
        pos = np.concatenate([np.arange(P), np.asarray([P + len(n) - 1 for n in nodes], np.int64)]).astype(np.int32)
        vis = np.zeros((T, T), dtype=bool)
        vis[:P, :P] = np.tril(np.ones((P, P), dtype=bool))
        for i, n in enumerate(nodes):
            r = P + i
            vis[r, :P] = True
            vis[r, r] = True
            for j in range(1, len(n)):
                vis[r, P + row[n[:j]] - 1] = True
        dev = self.m.device
        logits = self.m.forward_raw(torch.from_numpy(ids).to(dev), torch.from_numpy(pos).to(dev), torch.arange(T, dtype=torch.int32, device=dev),
                                    vis_bits_from_bool(torch.from_numpy(vis), self.m.max_slots).to(dev), T, 1 + len(nodes))
        return row, torch.log_softmax(logits.float(), dim=-1)

    @staticmethod
    def cum(row, logp, seq) -> float:
        """sum of the log-probabilities of seq's tokens (what beamSD.py:69-70 accumulates)"""
        s = 0.0
        for j, t in enumerate(seq):
            s += float(logp[row[tuple(seq[:j])], int(t)])
        return s


def _judge_one(base: torch.Tensor, rows: List[int], logp: torch.Tensor, allowed: List[torch.Tensor], chosen: List[Tuple[int, int, float]], n: int):
    """One decision, raw.  base [n_par] fp32 parent scores, rows = the parents' logp rows, allowed[p] = candidate token ids of parent p
    (ascending; one shared tensor per depth for the position mask, the trie's children per parent for the strict trie);
    chosen = (parent index, token, bf16 engine's score) of the items the bf16 engine picked; n = beams kept.
    -> dict(flat = fp32 score of every candidate (CPU), idx = candidates the bf16 engine chose, err = its score minus the fp32 score)"""
    parts, par, tok, off = [], [], [], [0]
    for p_i, (r, al) in enumerate(zip(rows, allowed)):
        parts.append(base[p_i] + logp[r][al])
        par.append(torch.full((al.numel(),), p_i, dtype=torch.long))
        tok.append(al.cpu())
        off.append(off[-1] + al.numel())
    flat = torch.cat(parts).cpu() if parts else torch.zeros(0)
    cand_par, cand_tok = (torch.cat(par), torch.cat(tok)) if parts else (torch.zeros(0, dtype=torch.long),) * 2
    pos = [{int(t): j for j, t in enumerate(al.tolist())} for al in allowed]
    idx = torch.tensor([off[p] + pos[p][t] for p, t, _ in chosen], dtype=torch.long)
    b_scores = torch.tensor([b for _, _, b in chosen], dtype=torch.float32)
    return dict(flat=flat, idx=idx, err=b_scores - flat[idx], n=n, cand_par=cand_par, cand_tok=cand_tok)


def evaluate(reports: List[dict]):
    """Second pass over the decisions of a whole batch.  The bf16 engine's score error is a property of (model, depth) -- an item's
    score is a sum of `depth + 1` log-probabilities -- not of the 20..40 items one decision happens to keep, so the noise level of a
    decision is the LARGEST |bf16 score - fp32 score| measured over every chosen item of every user at that (model, depth), and
    bound = C_NOISE x that.  Fills r["noise"], r["bound"], r["gap"], r["n_in"], r["n_clear"], r["n_same"], r["all_clear"], r["violations"]."""
    eps: Dict[Tuple[str, int], float] = {}
    for r in reports:
        if r["err"].numel():
            eps[r["key"]] = max(eps.get(r["key"], 0.0), float(r["err"].abs().max()))
    for r in reports:
        flat, idx, n, cpar, ctok = r["flat"], r["idx"], r["n"], r["cand_par"], r["cand_tok"]
        n_eff = min(n, int((flat > float("-inf")).sum()))
        if n_eff == 0:
            r.update(noise=0.0, own_noise=0.0, bound=0.0, gap=float("inf"), n_in=0, n_clear=0, n_same=0, all_clear=True, violations=[])
            continue
        top = torch.topk(flat, min(n_eff + 1, flat.numel())).values
        s_n = float(top[n_eff - 1])
        s_n1 = float(top[n_eff]) if top.numel() > n_eff else float("-inf")
        noise = eps.get(r["key"], 0.0)
        bound = C_NOISE * noise
        chosen_mask = torch.zeros_like(flat, dtype=torch.bool)
        chosen_mask[idx] = True
        in_f = flat >= s_n
        viol_a = in_f & ~chosen_mask & ((flat - s_n1) > bound)            # clearly among the n best, but the bf16 engine dropped it
        viol_b = chosen_mask & ~in_f & ((s_n - flat) > bound)             # clearly not among the n best, but the bf16 engine kept it
        clear_in = in_f & ((flat - s_n1) > bound)
        b_min = float((flat[idx] + r["err"]).min()) if idx.numel() else float("nan")
        viol = [dict(kind="dropped", parent=int(cpar[i]), tok=int(ctok[i]), score=float(flat[i]), boundary=s_n1, bound=bound,
                     implied_error=b_min - float(flat[i])) for i in torch.nonzero(viol_a).flatten().tolist()]
        viol += [dict(kind="kept", parent=int(cpar[i]), tok=int(ctok[i]), score=float(flat[i]), boundary=s_n, bound=bound)
                 for i in torch.nonzero(viol_b).flatten().tolist()]
        r.update(noise=noise, own_noise=float(r["err"].abs().max()) if r["err"].numel() else 0.0, bound=bound, gap=s_n - s_n1,
                 n_in=int(in_f.sum()), n_clear=int(clear_in.sum()), n_same=int((in_f & chosen_mask).sum()),
                 all_clear=(s_n - s_n1) > bound, violations=viol)
    return eps


def judge_user(prompt: np.ndarray, rounds: List[dict], jt: TreeJudge, jd: TreeJudge, allowed_of):
    """Replay one user's decision trace.  -> list of raw decision reports (dicts of _judge_one + `what`, `key`; thresholds are applied
    by `evaluate` over the whole batch) and the consistency of the
    acceptance tests (beamSD.py:371-380: step accepted iff every target pick is among the draft's candidates).
    `allowed_of(generated suffix as a tuple)` -> device LongTensor of the tokens the mask allows next, ascending."""
    reports = []
    accept_ok = True
    for rd in rounds:
        if rd["kind"] == "final":
            par = rd["parents"]["seq"]
            row, logp = jt.run(prompt, par)
            base = torch.tensor([TreeJudge.cum(row, logp, s) for s in par], dtype=torch.float32, device=logp.device)
            res = rd["result"]
            chosen = [(res["parent"][j], res["tok"][j], res["score"][j]) for j in range(len(res["tok"]))]
            # parent indices of the result block index the parents block's slots: map slot -> position among the valid rows
            slot_pos = {s: i for i, s in enumerate(rd["parents"]["index"])}
            chosen = [(slot_pos[p], t, b) for p, t, b in chosen]
            r = _judge_one(base, [row[tuple(s)] for s in par], logp, [allowed_of(tuple(s)) for s in par], chosen, rd["k"])
            r["what"], r["key"] = f"final step at depth {rd['gen0']}", ("target", rd["gen0"])
            reports.append(r)
            continue
        gen0, k, dk, dl, nm = rd["gen0"], rd["k"], rd["dk"], rd["draft_len"], rd["n_matches"]
        start = rd["start"]
        all_seqs = list(start["seq"]) + [s for b in rd["draft"] for s in b["seq"]] + [s for p in rd["picks"] for s in p["seq"]]
        rowT, logpT = jt.run(prompt, all_seqs)
        rowD, logpD = jd.run(prompt, all_seqs)
        dev = logpT.device
        f_start = [TreeJudge.cum(rowT, logpT, s) for s in start["seq"]]                 # fp32 target score of the round's beams
        # ---- the draft's steps (beamSD.py:108-179): parents = round beams, then the previous block; scores = parent + draft log-prob
        par_seq, par_score = start["seq"], f_start
        par_slot = {s: i for i, s in enumerate(start["index"])}
        for i in range(dl):
            blk = rd["draft"][i]
            chosen = [(par_slot[blk["parent"][j]], blk["tok"][j], blk["score"][j]) for j in range(len(blk["tok"]))]
            base = torch.tensor(par_score, dtype=torch.float32, device=dev)
            r = _judge_one(base, [rowD[tuple(s)] for s in par_seq], logpD, [allowed_of(tuple(s)) for s in par_seq], chosen, dk)
            r["what"], r["key"] = f"draft step {i} at depth {gen0 + i}", ("draft", gen0 + i)
            reports.append(r)
            nxt_score = [par_score[p] + float(logpD[rowD[tuple(par_seq[p])], t]) for p, t, _ in chosen]
            par_seq, par_score = blk["seq"], nxt_score
            par_slot = {s: j for j, s in enumerate(blk["index"])}
        # ---- the target's verify steps (beamSD.py:277-380): parents = round beams, then the previous step's picks
        par_seq, par_score = start["seq"], f_start
        for i in range(nm + 1):
            pk = rd["picks"][i]
            if i == 0:
                pos_of = {s: j for j, s in enumerate(start["index"])}
            else:                                       # picks' parents index draft block i (slot numbers): our parents are the previous picks
                prev = rd["picks"][i - 1]
                blk_seq = {s: tuple(q) for s, q in zip(rd["draft"][i - 1]["index"], rd["draft"][i - 1]["seq"])}
                prev_pos = {tuple(q): j for j, q in enumerate(prev["seq"])}
                pos_of = {s: prev_pos[q] for s, q in blk_seq.items() if q in prev_pos}
            chosen = [(pos_of[pk["parent"][j]], pk["tok"][j], pk["score"][j]) for j in range(len(pk["tok"]))]
            base = torch.tensor(par_score, dtype=torch.float32, device=dev)
            r = _judge_one(base, [rowT[tuple(s)] for s in par_seq], logpT, [allowed_of(tuple(s)) for s in par_seq], chosen, k)
            r["what"], r["key"] = f"verify step {i} at depth {gen0 + i}", ("target", gen0 + i)
            reports.append(r)
            # acceptance as the reference defines it, on the bf16 engine's own sets
            if i < dl:
                accepted = {tuple(s) for s in pk["seq"]} <= {tuple(s) for s in rd["draft"][i]["seq"]}
                accept_ok &= (accepted == (i < nm))
            nxt_score = [par_score[p] + float(logpT[rowT[tuple(par_seq[p])], t]) for p, t, _ in chosen]
            par_seq, par_score = pk["seq"], nxt_score
        accept_ok &= sorted(map(tuple, rd["picks"][nm]["seq"])) == sorted(map(tuple, rd["result"]["seq"]))
    return reports, accept_ok
