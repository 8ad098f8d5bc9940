"""Compatibility path for ARBITRARY `prefix_allowed_tokens_fn(batch_id, sentence) -> List[int]` callables.

The fast path compiles the mask into a device automaton (`generation_trie.ConstraintFSM`).  A callable that
cannot be compiled (any closure, e.g. the one `BaseDataset.get_prefix_allowed_tokens_fn` returns at
`code/data.py:96-104`) is served here the way the reference serves every mask: the function is called on the host
once per beam per step (`code/beamSD.py:60-64,286-291` through HF's `PrefixConstrainedLogitsProcessor`).  All
arithmetic still runs in libatspeed_hip — forwards (`atspeed_llama_forward`), the full-vocabulary normaliser
(`atspeed_lse_rows`), mask + expand + top-K (`atspeed_beam_expand_prune` over a per-step automaton whose node r
holds row r's allowed list) and the acceptance test (`atspeed_accept`); only the mask lists and the small beam
tables cross PCIe, with one synchronisation per step like the reference.  One user at a time.

Extra logits processors (`BSSD(..., logits_processor=LogitsProcessorList([...]))`, beamSD.py:469-478) are torch callables
`(input_ids [n, len], scores [n, V]) -> scores`, so a step with processors hands them the log-softmax rows as a device tensor
(`atspeed_log_softmax_rows`) -- the mask first, as HF's PrefixConstrainedLogitsProcessor does it (scores + (-inf outside the allowed
list)), then the caller's processors in order (HF appends custom processors after its own) -- and the library expands the rows they
return (`atspeed_beam_expand_prune_free`: row-wise top-k, then the K best (row, token) pairs; -inf entries are never picked).
The reference's post-top-k id filter runs whenever the processor list is non-empty (beamSD.py:80), mask or not.
Stage times (`draft/target/verify_time_cost`, the CSV columns inference.py:183-187 reads) are wall clock with a device
synchronisation at the end of each stage, as the reference's Timer measures them (beamSD.py:12-37).

Sampling mode (`generation_config.do_sample`, beamSD.py:65-75,293-321,332-369) with a host-side mask or processors (round 4): forwards and the
full-vocabulary log-softmax stay in the library; the tempered, masked rows of a step (<= DK x V fp32, 5 MB) come to the host, where the draws
are made from the SAME counter-based streams the device kernels use (`_HashRng`: (seed, purpose, round, step, model) -> sub-seed, one hash per
candidate id; scan.hip) -- so a callable that wraps a compilable constraint samples exactly what the device path samples for that seed
(tests/test_bssd_gpu.py).
"""
from __future__ import annotations

import ctypes as C
import time
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib, synth
from .model import HipLlama


class _Inputs:
    """Host-side description of one forward: tokens, positions, KV slots and visibility rows (bool [T, S])."""

    def __init__(self, ids, pos, slots, vis):
        self.ids = np.asarray(ids, np.int32)
        self.pos = np.asarray(pos, np.int32)
        self.slots = np.asarray(slots, np.int32)
        self.vis = np.asarray(vis, bool)


def _pad(v: np.ndarray, width: int) -> np.ndarray:
    if v.shape[1] >= width:
        return v[:, :width]
    return np.concatenate((v, np.zeros((v.shape[0], width - v.shape[1]), bool)), axis=1)


def _vis_bits(vis: np.ndarray, max_slots: int) -> np.ndarray:
    T, S = vis.shape
    full = np.zeros((T, max_slots), bool)
    full[:, :S] = vis
    return np.packbits(full.reshape(T, max_slots // 64, 64), axis=-1, bitorder="little").view(np.uint64).reshape(T, max_slots // 64).view(np.int64)


def _forward(model: HipLlama, inp: _Inputs, n_rows: int):
    """-> (logits [n_rows, ld] device view with row stride ld, lse [n_rows] device)."""
    dev = model.device
    lib = _lib.load()
    T = len(inp.ids)
    S = inp.vis.shape[1]
    with torch.cuda.device(dev):
        ids = torch.from_numpy(inp.ids).to(dev)
        pos = torch.from_numpy(inp.pos).to(dev)
        slots = torch.from_numpy(inp.slots).to(dev)
        bits = torch.from_numpy(_vis_bits(inp.vis, model.max_slots)).to(dev)
        ld = model.logits_ld
        logits = torch.empty(n_rows * ld, dtype=torch.float32, device=dev)
        _lib.check(lib.atspeed_llama_forward(model._handle, ids.data_ptr(), pos.data_ptr(), slots.data_ptr(), bits.data_ptr(),
                                             T, S, n_rows, logits.data_ptr(), _lib.stream_ptr(dev)))
        lse = torch.empty(n_rows, dtype=torch.float32, device=dev)
        _lib.check(lib.atspeed_lse_rows(logits.data_ptr(), n_rows, model.dims.vocab_size, ld, lse.data_ptr(), _lib.stream_ptr(dev)))
    return logits, lse


def _allowed_lists(fn: Callable, seqs: np.ndarray) -> List[List[int]]:
    out = []
    for r in range(seqs.shape[0]):
        al = fn(0, torch.from_numpy(seqs[r]))        # batch id 0: _num_beams is poked to the row count (beamSD.py:56,281)
        if len(al) == 0:                              # TypeError when fn returned None, like the HF processor
            raise ValueError("`prefix_allowed_tokens_fn` returned an empty list for batch ID 0."
                             "This means that the constraint is unsatisfiable. Please check your implementation"
                             "of `prefix_allowed_tokens_fn` ")
        out.append(sorted({int(t) for t in al}))
    return out


def _expand_prune(model: HipLlama, logits, lse, row_ids: Sequence[int], beam_scores: np.ndarray,
                  allowed: List[List[int]], k: int):
    """Mask + add beam scores + top-k on the device (beamSD.py:60-78).  `row_ids[r]` = logits row of candidate row r.
    Returns host arrays (score, parent(row index r), token, flat = r*V + token), only real beams (finite scores)."""
    lib = _lib.load()
    dev = model.device
    V = model.dims.vocab_size
    n = len(allowed)
    row_ptr = np.zeros(n + 1, np.int32)
    row_ptr[1:] = np.cumsum([len(a) for a in allowed])
    tok = np.asarray([t for a in allowed for t in a], np.int32)
    nxt = np.zeros(len(tok), np.int32)
    fsm = C.c_void_p()
    _lib.check(lib.atspeed_fsm_create(row_ptr.ctypes.data, tok.ctypes.data, nxt.ctypes.data, n, len(tok), V, C.byref(fsm)))
    try:
        with torch.cuda.device(dev):
            ld = model.logits_ld
            rows = torch.as_tensor(list(row_ids), dtype=torch.int64, device=dev)
            lg = logits.view(-1, ld)[rows].contiguous()               # gather the candidate rows (verify keeps only hit beams)
            ls = lse[rows].contiguous()
            bs = torch.from_numpy(np.asarray(beam_scores, np.float32)).to(dev)
            nd = torch.arange(n, dtype=torch.int32, device=dev)
            o_s = torch.empty(k, dtype=torch.float32, device=dev)
            o_p, o_t, o_n, o_f = (torch.empty(k, dtype=torch.int32, device=dev) for _ in range(4))
            _lib.check(lib.atspeed_beam_expand_prune(lg.data_ptr(), ld, ls.data_ptr(), bs.data_ptr(), nd.data_ptr(), n, fsm, k,
                                                     o_s.data_ptr(), o_p.data_ptr(), o_t.data_ptr(), o_n.data_ptr(), o_f.data_ptr(),
                                                     _lib.stream_ptr(dev)))
            s, p, t, f = (x.cpu().numpy() for x in (o_s, o_p, o_t, o_f))
    finally:
        lib.atspeed_fsm_destroy(fsm)
    keep = f >= 0
    return s[keep], p[keep].astype(np.int64), t[keep].astype(np.int64), f[keep].astype(np.int64)


def _expand_processed(model: HipLlama, logits, lse, row_ids: Sequence[int], beam_scores: np.ndarray, seqs: np.ndarray,
                      fn: Optional[Callable], procs: Sequence[Callable], k: int):
    """The expand of a step WITH extra logits processors (beamSD.py:58-78): log-softmax rows -> mask -> processors -> + beam scores ->
    top-k.  Returns host arrays like `_expand_prune`."""
    lib = _lib.load()
    dev = model.device
    V, ld = model.dims.vocab_size, model.logits_ld
    n = len(row_ids)
    with torch.cuda.device(dev):
        st = _lib.stream_ptr(dev)
        rows = torch.as_tensor(list(row_ids), dtype=torch.int64, device=dev)
        lg = logits.view(-1, ld)[rows].contiguous()
        ls = lse[rows].contiguous()
        scores = torch.empty(n, V, dtype=torch.float32, device=dev)
        _lib.check(lib.atspeed_log_softmax_rows(lg.data_ptr(), ld, ls.data_ptr(), n, V, scores.data_ptr(), V, st))
        ids = torch.from_numpy(np.ascontiguousarray(seqs)).to(dev)
        if fn is not None:                              # transformers PrefixConstrainedLogitsProcessor.__call__: scores + mask
            mask = torch.full_like(scores, float("-inf"))
            for r, al in enumerate(_allowed_lists(fn, seqs)):
                mask[r, torch.as_tensor(al, dtype=torch.long, device=dev)] = 0
            scores = scores + mask
        for proc in procs:
            scores = proc(ids, scores)
        scores = scores.to(torch.float32).contiguous()
        bs = torch.from_numpy(np.asarray(beam_scores, np.float32)).to(dev)
        zero = torch.zeros(n, dtype=torch.float32, device=dev)
        ws = torch.empty(n * _lib.MAX_BEAMS, dtype=torch.int32, device=dev)
        o_s = torch.empty(k, dtype=torch.float32, device=dev)
        o_p, o_t, o_f = (torch.empty(k, dtype=torch.int32, device=dev) for _ in range(3))
        _lib.check(lib.atspeed_beam_expand_prune_free(scores.data_ptr(), V, zero.data_ptr(), bs.data_ptr(), n, V, k, ws.data_ptr(),
                                                      o_s.data_ptr(), o_p.data_ptr(), o_t.data_ptr(), o_f.data_ptr(), st))
        s, p, t, f = (x.cpu().numpy() for x in (o_s, o_p, o_t, o_f))
    keep = f >= 0
    return s[keep], p[keep].astype(np.int64), t[keep].astype(np.int64), f[keep].astype(np.int64)


def _expand(model, logits, lse, row_ids, beam_scores, seqs, fn, procs, k):
    if procs:
        return _expand_processed(model, logits, lse, row_ids, beam_scores, seqs, fn, procs, k)
    return _expand_prune(model, logits, lse, row_ids, beam_scores, _allowed_lists(fn, seqs), k)


def _one_step(model: HipLlama, inp: _Inputs, k: int, beam_scores: np.ndarray, beam_seq: np.ndarray, fn: Optional[Callable],
              procs: Sequence[Callable] = ()) -> Dict:
    """one_step_beam_search (beamSD.py:40-106)."""
    n = len(beam_scores)
    logits, lse = _forward(model, inp, n)
    seqs = beam_seq[:1] if (n == 1 and k != 1) else beam_seq                       # :61-64
    s, p, t, f = _expand(model, logits, lse, range(n), beam_scores, seqs, fn, procs, k)
    keep = (t >= 32000) | (t == 2)                                                  # :80-86 (hard-coded Llama vocab / EOS; any processor switches it on)
    s, p, t, f = s[keep], p[keep], t[keep], f[keep]
    m = len(t)
    S = inp.vis.shape[1]
    vis = np.concatenate((inp.vis[-n:][p], np.eye(m, dtype=bool)), axis=1)          # :89
    nxt = _Inputs(t, np.full(m, inp.pos[-1] + 1), np.arange(S, S + m), vis)         # :91
    return dict(flat=f, scores=s, parents=p, tokens=t, seq=np.concatenate((beam_seq[p], t[:, None]), axis=1), next=nxt)


def _causal(ids: np.ndarray) -> _Inputs:
    n = len(ids)
    return _Inputs(ids, np.arange(n), np.arange(n), np.tril(np.ones((n, n), bool)))


def target_generate_host_mask(model: HipLlama, prompt: np.ndarray, max_new_tokens: int, fn: Optional[Callable],
                              procs: Sequence[Callable] = ()) -> Dict:
    k = int(model.generation_config.num_beams)
    inp = _causal(prompt)
    scores = np.zeros(1, np.float32)
    seq = np.repeat(prompt[None, :], k, axis=0)
    for _ in range(max_new_tokens):                                                 # beamSD.py:579-588
        o = _one_step(model, inp, k, scores, seq, fn, procs)
        inp, scores, seq = o["next"], o["scores"], o["seq"]
    return dict(beam_sequence=seq, beam_scores=scores)


class _Stage:
    """wall clock of a stage with a device synchronisation at its end (the reference's Timer, beamSD.py:12-37)"""

    def __init__(self, acc: Dict[str, float], key: str, dev):
        self.acc, self.key, self.dev = acc, key, dev

    def __enter__(self):
        self.t0 = time.time()

    def __exit__(self, *exc):
        torch.cuda.synchronize(self.dev)
        self.acc[self.key] += time.time() - self.t0


def bssd_host_mask(target: HipLlama, draft: HipLlama, prompt: np.ndarray, gamma: int, max_new_tokens: int, fn: Optional[Callable],
                   procs: Sequence[Callable] = ()) -> Dict:
    """BSSD (beamSD.py:458-542) with the mask function / logits processors on the host."""
    lib = _lib.load()
    cost = {"draft_time_cost": 0.0, "target_time_cost": 0.0, "verify_time_cost": 0.0}
    k, dk = int(target.generation_config.num_beams), int(draft.generation_config.num_beams)
    V = target.dims.vocab_size
    cur_len, max_len = len(prompt), len(prompt) + max_new_tokens
    tin = din = _causal(prompt)
    scores = np.zeros(1, np.float32)
    seq = np.repeat(prompt[None, :], k, axis=0)
    accept_steps: List[int] = []
    while cur_len < max_len:
        dl = min(gamma, max_len - cur_len - 1)                                      # :504
        if dl == 0:                                                                 # :505-509 (in no stage's sum: the reference breaks before :523-525)
            o = _one_step(target, tin, k, scores, seq, fn, procs)
            seq, scores = o["seq"], o["scores"]
            break
        # ---- draft (:108-179)
        steps, inp, d_scores, d_seq = [], din, scores, seq
        step_len, step_seq = [len(scores)], [seq]
        with _Stage(cost, "draft_time_cost", draft.device):
            for _ in range(dl):
                o = _one_step(draft, inp, dk, d_scores, d_seq, fn, procs)
                inp, d_scores, d_seq = o["next"], o["scores"], o["seq"]
                steps.append(o)
                step_len.append(len(d_scores))
                step_seq.append(d_seq)
        # ---- target: one forward over round inputs ++ every draft block (:190-232)
        blocks = [tin] + [o["next"] for o in steps]
        width = max(b.vis.shape[1] for b in blocks)
        packed = _Inputs(np.concatenate([b.ids for b in blocks]), np.concatenate([b.pos for b in blocks]),
                         np.concatenate([b.slots for b in blocks]), np.concatenate([_pad(b.vis, width) for b in blocks], axis=0))
        n_rows = sum(step_len)
        with _Stage(cost, "target_time_cost", target.device):
            logits, lse = _forward(target, packed, n_rows)
        t_verify = time.time()
        # ---- verify (:242-456, greedy)
        n0 = len(tin.ids)
        nm, lo, hi = 0, 0, step_len[0]
        hit = hit4 = None
        v_scores = scores
        for i in range(dl + 1):
            rows = list(range(lo, hi))
            if nm != dl:
                lo, hi = hi, hi + step_len[i + 1]
            seqs = step_seq[i]
            if i > 0:
                rows = [rows[h] for h in hit]
                seqs = seqs[hit]
                v_scores = v_scores[hit4]
            if i == 0 and len(rows) == 1 and k != 1:
                seqs = seqs[:1]
            s, p, t, f = _expand(target, logits, lse, rows, v_scores, seqs, fn, procs, k)
            v_scores = s
            parents = hit[p] if i > 0 else p
            flat = parents * V + t
            if nm == dl:
                break
            d_flat = steps[i]["flat"]
            kk, dd = len(flat), len(d_flat)
            with torch.cuda.device(target.device):                                  # acceptance on the device (:371-380)
                tf = torch.from_numpy(flat.astype(np.int32)).cuda()
                ts = torch.from_numpy(np.asarray(s, np.float32)).cuda()
                df = torch.from_numpy(d_flat.astype(np.int32)).cuda()
                h_out = torch.empty(kk, dtype=torch.int32, device="cuda")
                sb = torch.empty(kk, dtype=torch.float32, device="cuda")
                acc = torch.empty(1, dtype=torch.int32, device="cuda")
                _lib.check(lib.atspeed_accept(tf.data_ptr(), ts.data_ptr(), kk, df.data_ptr(), dd, h_out.data_ptr(), sb.data_ptr(),
                                              acc.data_ptr(), _lib.stream_ptr(target.device)))
                accepted = bool(acc.item()) and kk == k
                if accepted:
                    hit = h_out.cpu().numpy().astype(np.int64)
            if not accepted:
                break
            pos_of = {int(d): j for j, d in enumerate(d_flat)}
            hit4 = np.argsort(np.asarray([pos_of[int(y)] for y in flat]), kind="stable")
            nm += 1
        seq = np.concatenate((step_seq[nm][parents], t[:, None]), axis=1)           # :383
        scores = v_scores
        blk_lo = n0 - step_len[0] + sum(step_len[:nm])
        blk_rows = packed.vis[blk_lo: blk_lo + step_len[nm]]
        base = int(packed.slots[blk_lo + step_len[nm] - 1]) + 1
        m = len(t)
        vis = np.concatenate((_pad(blk_rows, base)[parents], np.eye(m, dtype=bool)), axis=1)
        tin = _Inputs(t, np.full(m, packed.pos[blk_lo] + 1), np.arange(base, base + m), vis)
        din = tin
        if nm == dl:                                                                # :402-416: the draft re-ingests its last block
            last = steps[dl - 1]["next"]
            din = _Inputs(np.concatenate((last.ids, tin.ids)), np.concatenate((last.pos, tin.pos)),
                          np.concatenate((last.slots, tin.slots)), np.concatenate((_pad(last.vis, base + m), vis), axis=0))
        cur_len += nm + 1
        accept_steps.append(nm)
        torch.cuda.synchronize(target.device)
        cost["verify_time_cost"] += time.time() - t_verify
    n_run, total = len(accept_steps), sum(accept_steps)
    return dict(beam_sequence=seq, beam_scores=scores, n_run=n_run, total_accept_steps=total, total_accept_tokens=total * k,
                ave_accept_tokens=total * k / n_run if n_run else 0.0, accept_steps=accept_steps, **cost)


# ---------------------------------------------------------------------------------------------- sampling mode on the host path
P_STEP, P_ACCEPT, P_PERM, P_RESID, P_BONUS = 1, 2, 3, 4, 5     # purposes of a random stream (scan.hip)


class _HashRng:
    """The device's counter-based generator (scan.hip / common.h ats_rng_sub): every draw is a pure function of
    (seed, purpose, round, step, model tag, element id).  A draw without replacement = top-n of log w + Gumbel(hash(id)) (Plackett-Luce, the
    law of torch.multinomial's sequential draws), a uniform = ((h >> 9) + 0.5) 2^-23, a random subset = the n smallest hashes."""

    def __init__(self, seed: int):
        self.seed, self.sub = int(seed) & 0xFFFFFFFF, 0

    def begin(self, purpose: int, rnd: int, step: int, model_tag: int = 0):
        ctr = (purpose & 0xFF) | ((rnd & 0xFF) << 8) | ((step & 0xFF) << 16) | ((model_tag & 0xFF) << 24)
        self.sub = int(synth.hash_u32(np.array([ctr], dtype=np.uint64), self.seed)[0])

    def _u01(self, ids: np.ndarray) -> np.ndarray:
        h = synth.hash_u32(ids.astype(np.uint64), self.sub)
        return ((h >> np.uint64(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)

    def multinomial_log(self, logw: np.ndarray, n: int) -> np.ndarray:
        ids = np.nonzero(np.isfinite(logw))[0]
        u = self._u01(ids)
        keys = logw[ids].astype(np.float32) + (-np.log(-np.log(u, dtype=np.float32), dtype=np.float32)).astype(np.float32)
        order = np.lexsort((ids, -keys))                 # key desc, id asc
        return ids[order[:n]].astype(np.int64)

    def uniform_ids(self, ids: np.ndarray) -> np.ndarray:
        return self._u01(ids)

    def subset_ids(self, ids: np.ndarray, n: int) -> np.ndarray:
        h = synth.hash_u32(ids.astype(np.uint64), self.sub)
        return np.lexsort((np.arange(len(ids)), h))[:n].astype(np.int64)


def _softmax(x: np.ndarray) -> np.ndarray:
    """softmax over a flat fp32 vector with -inf entries (all -inf -> zeros, as the reference zeroes its NaNs, beamSD.py:319-320)"""
    m = np.max(x) if x.size else -np.inf
    if not np.isfinite(m):
        return np.zeros_like(x, dtype=np.float32)
    e = np.exp((x - m).astype(np.float32), dtype=np.float32)
    return (e / e.sum(dtype=np.float32)).astype(np.float32)


def _tempered_rows(model: HipLlama, logits, lse, row_ids: Sequence[int], seqs: np.ndarray, fn: Optional[Callable], procs: Sequence[Callable],
                   temperature: float) -> np.ndarray:
    """log-softmax over the full vocabulary (:58,:285; library), prefix mask (:60-64,:286-291), processors, temperature warper (:65-66,
    :293-294) -> host fp32 [n, V]"""
    lib = _lib.load()
    dev = model.device
    V, ld = model.dims.vocab_size, model.logits_ld
    n = len(row_ids)
    with torch.cuda.device(dev):
        rows = torch.as_tensor(list(row_ids), dtype=torch.int64, device=dev)
        lg = logits.view(-1, ld)[rows].contiguous()
        ls = lse[rows].contiguous()
        scores = torch.empty(n, V, dtype=torch.float32, device=dev)
        _lib.check(lib.atspeed_log_softmax_rows(lg.data_ptr(), ld, ls.data_ptr(), n, V, scores.data_ptr(), V, _lib.stream_ptr(dev)))
        if fn is not None:
            mask = torch.full_like(scores, float("-inf"))
            for r, al in enumerate(_allowed_lists(fn, seqs)):
                mask[r, torch.as_tensor(al, dtype=torch.long, device=dev)] = 0
            scores = scores + mask
        if procs:
            ids = torch.from_numpy(np.ascontiguousarray(seqs)).to(dev)
            for proc in procs:
                scores = proc(ids, scores)
        out = (scores.to(torch.float32) / float(temperature)).cpu().numpy()
    return out


def _one_step_sample(model: HipLlama, inp: _Inputs, k: int, beam_scores: np.ndarray, beam_seq: np.ndarray, fn, procs, temperature: float,
                     rng: _HashRng) -> Dict:
    """one_step_beam_search with do_sample (beamSD.py:40-106, :65-75): k draws without replacement from softmax of the flattened scores."""
    n = len(beam_scores)
    V = model.dims.vocab_size
    logits, lse = _forward(model, inp, n)
    seqs = beam_seq[:1] if (n == 1 and k != 1) else beam_seq
    flat = (_tempered_rows(model, logits, lse, range(n), seqs, fn, procs, temperature) + np.asarray(beam_scores, np.float32)[:, None]).reshape(-1)
    idx = rng.multinomial_log(flat, k)
    s = flat[idx]
    p, t = idx // V, idx % V
    if fn is not None or procs:                                                     # :80-86 (any processor switches the id filter on)
        keep = ((t >= 32000) | (t == 2)) & np.isfinite(s)
        idx, s, p, t = idx[keep], s[keep], p[keep], t[keep]
    m = len(t)
    S = inp.vis.shape[1]
    vis = np.concatenate((inp.vis[-n:][p], np.eye(m, dtype=bool)), axis=1)
    nxt = _Inputs(t, np.full(m, inp.pos[-1] + 1), np.arange(S, S + m), vis)
    return dict(flat=idx, scores=s.astype(np.float32), parents=p, tokens=t, seq=np.concatenate((beam_seq[p], t[:, None]), axis=1), next=nxt,
                probs=_softmax(flat), dist=flat)


def _final_sort(seq: np.ndarray, scores: np.ndarray):
    o = np.argsort(-scores, kind="stable")                                          # beamSD.py:529-531, :589-591
    return seq[o], scores[o]


def target_generate_host_mask_sample(model: HipLlama, prompt: np.ndarray, max_new_tokens: int, fn, procs, temperature: float, seed: int) -> Dict:
    k = int(model.generation_config.num_beams)
    rng = _HashRng(seed)
    inp = _causal(prompt)
    scores = np.zeros(1, np.float32)
    seq = np.repeat(prompt[None, :], k, axis=0)
    for g in range(max_new_tokens):
        rng.begin(P_STEP, g, 0, 0)
        o = _one_step_sample(model, inp, k, scores, seq, fn, procs, temperature, rng)
        inp, scores, seq = o["next"], o["scores"], o["seq"]
    seq, scores = _final_sort(seq, scores)
    return dict(beam_sequence=seq, beam_scores=scores)


def bssd_host_mask_sample(target: HipLlama, draft: HipLlama, prompt: np.ndarray, gamma: int, max_new_tokens: int, fn, procs,
                          temperature: float, seed: int) -> Dict:
    """BSSD with `generation_config.do_sample` (beamSD.py:458-542; verify :293-321 distributions, :332-369 accept / resample, :303-309 bonus
    draw) and the mask / processors on the host.  One documented deviation, as on the device path: with no residual mass left the remaining
    draws come from the target distribution (the reference resamples uniformly over the whole vocabulary, -inf scores included)."""
    cost = {"draft_time_cost": 0.0, "target_time_cost": 0.0, "verify_time_cost": 0.0}
    k, dk = int(target.generation_config.num_beams), int(draft.generation_config.num_beams)
    V = target.dims.vocab_size
    rng = _HashRng(seed)
    cur_len, max_len = len(prompt), len(prompt) + max_new_tokens
    tin = din = _causal(prompt)
    scores = np.zeros(1, np.float32)
    seq = np.repeat(prompt[None, :], k, axis=0)
    accept_steps: List[int] = []
    while cur_len < max_len:
        rnd = len(accept_steps)
        dl = min(gamma, max_len - cur_len - 1)
        if dl == 0:
            rng.begin(P_STEP, rnd, 0, 0)
            o = _one_step_sample(target, tin, k, scores, seq, fn, procs, temperature, rng)
            seq, scores = o["seq"], o["scores"]
            break
        # ---- draft
        steps, inp, d_scores, d_seq = [], din, scores, seq
        step_len, step_seq = [len(scores)], [seq]
        with _Stage(cost, "draft_time_cost", draft.device):
            for i in range(dl):
                rng.begin(P_STEP, rnd, i, 1)
                o = _one_step_sample(draft, inp, dk, d_scores, d_seq, fn, procs, temperature, rng)
                inp, d_scores, d_seq = o["next"], o["scores"], o["seq"]
                steps.append(o)
                step_len.append(len(d_scores))
                step_seq.append(d_seq)
        # ---- target: one packed forward
        blocks = [tin] + [o["next"] for o in steps]
        width = max(b.vis.shape[1] for b in blocks)
        packed = _Inputs(np.concatenate([b.ids for b in blocks]), np.concatenate([b.pos for b in blocks]),
                         np.concatenate([b.slots for b in blocks]), np.concatenate([_pad(b.vis, width) for b in blocks], axis=0))
        n_rows = sum(step_len)
        with _Stage(cost, "target_time_cost", target.device):
            logits, lse = _forward(target, packed, n_rows)
        t_verify = time.time()
        # ---- verify
        n0 = len(tin.ids)
        nm, lo, hi = 0, 0, step_len[0]
        hit = None
        v_scores = scores
        for i in range(dl + 1):
            rows = list(range(lo, hi))
            if nm != dl:
                lo, hi = hi, hi + step_len[i + 1]
            seqs = step_seq[i]
            if i > 0:
                rows = [rows[h] for h in hit]
                seqs = seqs[hit]
            if i == 0 and len(rows) == 1 and k != 1:
                seqs = seqs[:1]
            bs = (_tempered_rows(target, logits, lse, rows, seqs, fn, procs, temperature) + np.asarray(v_scores, np.float32)[:, None])
            if i > 0:                                                               # :311-321: into the draft's beam space
                tbs = np.full((step_len[i], V), -np.inf, dtype=np.float32)
                tbs[hit] = bs
                bs = tbs
            bs = bs.reshape(-1)
            if nm == dl:                                                            # :303-309 bonus draw from the target
                rng.begin(P_BONUS, rnd, i, 0)
                nxt = rng.multinomial_log(bs, k)
                v_scores = bs[nxt]
                parents, t = nxt // V, nxt % V
                break
            probs = _softmax(bs)
            dprobs, d_ids = steps[i]["probs"], steps[i]["flat"]
            p_i, q_i = probs[d_ids], dprobs[d_ids]
            rng.begin(P_ACCEPT, rnd, i, 0)
            r = rng.uniform_ids(np.arange(len(d_ids)))
            acc = (r * q_i) <= p_i                                                  # r <= p / q without the division, as the device tests it
            acc_tokens = d_ids[acc]
            n_acc = int(acc.sum())
            if n_acc >= k:                                                          # :341-350
                nm += 1
                rng.begin(P_PERM, rnd, i, 0)
                sel = rng.subset_ids(np.nonzero(acc)[0], k)
                seq_tokens = np.sort(acc_tokens[sel])
                pos_of = {int(d): j for j, d in enumerate(d_ids.tolist())}
                hit = np.asarray([pos_of[int(y)] for y in seq_tokens.tolist()], dtype=np.int64)
                v_scores = bs[seq_tokens]
                parents, t = seq_tokens // V, seq_tokens % V
            else:                                                                   # :351-369 reject: resample the missing beams
                newp = np.clip(probs - dprobs, 0, None).astype(np.float32)
                newp[acc_tokens] = 0
                if float(newp.sum()) == 0.0:
                    newp = probs.copy()
                    newp[acc_tokens] = 0
                rng.begin(P_RESID, rnd, i, 0)
                with np.errstate(divide="ignore"):
                    nxt = rng.multinomial_log(np.log(newp, dtype=np.float32), k - n_acc)
                seq_tokens = np.sort(np.concatenate((acc_tokens, nxt)))
                v_scores = bs[seq_tokens]
                parents, t = seq_tokens // V, seq_tokens % V
                break
        seq = np.concatenate((step_seq[nm][parents], t[:, None]), axis=1)           # :383
        scores = np.asarray(v_scores, np.float32)
        blk_lo = n0 - step_len[0] + sum(step_len[:nm])
        blk_rows = packed.vis[blk_lo: blk_lo + step_len[nm]]
        base = int(packed.slots[blk_lo + step_len[nm] - 1]) + 1
        m = len(t)
        vis = np.concatenate((_pad(blk_rows, base)[parents], np.eye(m, dtype=bool)), axis=1)
        tin = _Inputs(t, np.full(m, packed.pos[blk_lo] + 1), np.arange(base, base + m), vis)
        din = tin
        if nm == dl:
            last = steps[dl - 1]["next"]
            din = _Inputs(np.concatenate((last.ids, tin.ids)), np.concatenate((last.pos, tin.pos)),
                          np.concatenate((last.slots, tin.slots)), np.concatenate((_pad(last.vis, base + m), vis), axis=0))
        cur_len += nm + 1
        accept_steps.append(nm)
        torch.cuda.synchronize(target.device)
        cost["verify_time_cost"] += time.time() - t_verify
    seq, scores = _final_sort(seq, scores)
    n_run, total = len(accept_steps), sum(accept_steps)
    return dict(beam_sequence=seq, beam_scores=scores, n_run=n_run, total_accept_steps=total, total_accept_tokens=total * k,
                ave_accept_tokens=total * k / n_run if n_run else 0.0, accept_steps=accept_steps, **cost)
