"""Drop-in surface of the reference's `code/beamSD.py` for the MI355X engine.

  BSSD(target_model, draft_model, inputs, gamma, max_new_tokens, logits_processor=None,
       prefix_allowed_tokens_fn=None) -> Dict            <- beamSD.py:458-542
  target_generate(model, inputs, max_new_tokens, logits_processor=None,
       prefix_allowed_tokens_fn=None) -> Dict            <- beamSD.py:544-595
  Timer(func="", sync_cuda=True, syn_device=0)            <- beamSD.py:12-37
  beam_sd_generate = BSSD (name used by BASELINE.json's north star)

Same positional order and the same result keys (`beam_sequence, beam_scores, n_run,
total_accept_steps, total_accept_tokens, ave_accept_tokens, draft_time_cost,
target_time_cost, verify_time_cost, time_cost`; consumed at `code/inference.py:179-187`).
The models are `HipLlama` objects; the whole loop runs in libatspeed_hip
(`atspeed_bssd_generate`) with one host read-back per verification round instead of the
reference's per-beam mask calls and `.tolist()` syncs (beamSD.py:62-64,371-372).

Mask functions that can `compile()` (PositionSetConstraint, SuffixTrieConstraint, prefix_allowed_tokens_fn(trie)) run
as a device automaton; any other callable is served by `hostmask.py` the way the reference does it (one host call per
beam per step, all arithmetic still in HIP).  `generation_config.do_sample` selects the sampling branch
(beamSD.py:65-75,293-321,332-369) with counter-based draws (`seed=` or torch's generator picks the stream).
No mask at all (`prefix_allowed_tokens_fn=None`, legal in the reference: beamSD.py:460-481) runs on the device too: every token is a
candidate, the id filter of :80-86 is off as in the reference.  Extra `logits_processor` entries (the reference always passes None,
inference.py:175-176) are torch callables and are served by the host path: each step's log-softmax rows go through them between the
library's forward and its expand + top-K.  Sampling with a host-side mask or processors (round 4) draws on the host from the device's
counter-based streams (hostmask.py): a callable wrapping a compilable constraint samples exactly what the device path samples for that seed.
"""
from __future__ import annotations

import ctypes as C
import time
import weakref
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .generation_trie import ConstraintFSM, free_constraint
from .model import HipLlama


class Timer:
    """Context manager / decorator; as a decorator it injects `result["time_cost"]`."""

    def __init__(self, func="", sync_cuda=True, syn_device=0):
        self.func = func
        self.sync_cuda = sync_cuda
        self.syn_device = syn_device

    def _sync(self):
        if self.sync_cuda and torch.cuda.is_available():
            torch.cuda.synchronize(self.syn_device)

    def __enter__(self):
        self.start = time.time()
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self._sync()
        self.time_cost = time.time() - self.start

    def __call__(self, func):
        def wrapper(*args, syn_device=None, **kwargs):
            if syn_device is not None:
                self.syn_device = syn_device
            self.start = time.time()
            result = func(*args, **kwargs)
            self._sync()
            self.time_cost = time.time() - self.start
            result["time_cost"] = self.time_cost
            return result
        wrapper.__name__ = getattr(func, "__name__", "wrapped")
        wrapper.__doc__ = func.__doc__
        return wrapper


# ---------------------------------------------------------------- device handles (cached)
class _DeviceFSM:
    """Device copy of a ConstraintFSM's CSR arrays (shared by every prompt; only the start node differs)."""
    _cache: Dict[int, "_DeviceFSM"] = {}

    def __init__(self, fsm: ConstraintFSM, vocab_size: int):
        lib = _lib.load()
        self.arrays = (np.ascontiguousarray(fsm.row_ptr, np.int32), np.ascontiguousarray(fsm.tok, np.int32),
                       np.ascontiguousarray(fsm.nxt, np.int32))
        h = C.c_void_p()
        if fsm.free:
            _lib.check(lib.atspeed_fsm_create_free(vocab_size, C.byref(h)))
        else:
            _lib.check(lib.atspeed_fsm_create(self.arrays[0].ctypes.data, self.arrays[1].ctypes.data, self.arrays[2].ctypes.data,
                                              fsm.n_nodes, len(self.arrays[1]), vocab_size, C.byref(h)))
        if fsm.id_filter is not None and not fsm.free:
            _lib.check(lib.atspeed_fsm_set_id_filter(h, int(fsm.id_filter[0]), int(fsm.id_filter[1])))
        self.handle = h
        self.src = fsm.row_ptr       # keeps id() stable while cached

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            try:
                _lib.load().atspeed_fsm_destroy(h)
            except Exception:
                pass
            self.handle = None

    @classmethod
    def get(cls, fsm: ConstraintFSM, vocab_size: int) -> "_DeviceFSM":
        key = (id(fsm.row_ptr), id(fsm.tok), vocab_size, fsm.id_filter)
        d = cls._cache.get(key)
        if d is None or d.src is not fsm.row_ptr:
            d = cls(fsm, vocab_size)
            cls._cache[key] = d
        return d


class _Decoder:
    """One user stream (private KV arenas + beam state) of a (target, draft) pair.  The cache refers to its models WEAKLY and
    drops a model's decoders when the model is collected, so `del model` really frees the weights and the ~0.27 GB of KV
    arenas per lane (a `--run_beam_sizes` sweep would otherwise keep every beam size's models resident)."""
    _cache: Dict[tuple, "_Decoder"] = {}
    _watched: set = set()

    def __init__(self, target: HipLlama, draft: Optional[HipLlama], max_prompt: int):
        lib = _lib.load()
        h = C.c_void_p()
        with torch.cuda.device(target.device):
            _lib.check(lib.atspeed_decoder_create(target._handle, draft._handle if draft is not None else None,
                                                  max_prompt, C.byref(h)))
        self.handle, self.max_prompt = h, max_prompt
        self.models = (weakref.ref(target), weakref.ref(draft) if draft is not None else None)
        for m in (target, draft):
            if m is not None and id(m) not in _Decoder._watched:
                _Decoder._watched.add(id(m))
                weakref.finalize(m, _evict_model, id(m))

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            try:
                _lib.load().atspeed_decoder_destroy(h)
            except Exception:
                pass
            self.handle = None

    def serves(self, target, draft) -> bool:
        return self.models[0]() is target and (self.models[1]() if self.models[1] is not None else None) is draft

    @classmethod
    def get(cls, target: HipLlama, draft: Optional[HipLlama], prompt_len: int, lane: int = 0) -> "_Decoder":
        """One decoder (= one user stream: private KV + activations) per (model pair, lane)."""
        key = (id(target), id(draft) if draft is not None else None, lane)
        d = cls._cache.get(key)
        if d is None or d.max_prompt < prompt_len or not d.serves(target, draft):
            d = cls(target, draft, max(prompt_len, min(target.max_tokens, 512)))
            cls._cache[key] = d
        return d


def _evict_model(model_id: int) -> None:
    """weakref.finalize callback: the model with this id() is gone, so are the decoders that used it."""
    _Decoder._watched.discard(model_id)
    for k in [k for k in _Decoder._cache if model_id in k[:2]]:
        del _Decoder._cache[k]


def release_decoders(*models) -> int:
    """Free the cached per-user decoders (KV arenas: ~270 MB each at Llama-7B dims, 512 slots) of the given models, or all of
    them when called without arguments; returns how many were freed.  They are re-created on demand."""
    ids = {id(m) for m in models}
    keys = [k for k in _Decoder._cache if not ids or k[0] in ids or k[1] in ids]
    for k in keys:
        del _Decoder._cache[k]
    return len(keys)


def _compile_constraint(fn, prompt):
    if fn is None:               # no mask: every token is a candidate (beamSD.py:469-478 builds an empty processor list)
        return free_constraint()
    return fn.compile(prompt)


def _host_path(logits_processor, prefix_allowed_tokens_fn) -> bool:
    """True when a step must call back into Python: extra logits processors, or a mask callable that cannot compile() itself."""
    if logits_processor is not None and len(logits_processor) != 0:
        return True
    return prefix_allowed_tokens_fn is not None and not hasattr(prefix_allowed_tokens_fn, "compile")


def _prompt_lists(prompts):
    """Token lists of all prompts with ONE device-to-host copy (a `.tolist()` per user is a synchronising copy each)."""
    if len(prompts) == 1:
        return [prompts[0].tolist()]
    flat = torch.cat([p.reshape(-1) for p in prompts]).tolist()
    out, s0 = [], 0
    for p in prompts:
        out.append(flat[s0: s0 + p.numel()])
        s0 += p.numel()
    return out


def _check_models(*models):
    for m in models:
        if not isinstance(m, HipLlama):
            raise TypeError("models must be atspeed_amd.HipLlama (use HipLlama.from_hf(model) for an HF module)")


def _sampling(model, seed):
    """(do_sample, temperature, seed) of a call.  The reference samples when `target_model.generation_config.do_sample`
    is set (beamSD.py:479-481) with its temperature warper; draws there come from torch's global generator, here from a
    counter-based stream whose 32-bit seed is drawn from that generator (so `torch.manual_seed` makes a call repeatable)
    unless `seed` is given."""
    gc = model.generation_config
    if not getattr(gc, "do_sample", False):
        return False, 1.0, 0
    if seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    temp = getattr(gc, "temperature", None)
    return True, 1.0 if temp is None else float(temp), int(seed) & 0xFFFFFFFF


def _set_sampling(decs, mode, per_user: bool = True):
    do, temp, seed = mode
    lib = _lib.load()
    for u, d in enumerate(decs):
        _lib.check(lib.atspeed_decoder_set_sampling(d.handle, 1 if do else 0, temp, (seed + (u if per_user else 0)) & 0xFFFFFFFF))


def _set_trace(decs, on: bool):
    lib = _lib.load()
    for d in decs:
        if getattr(d, "trace_on", False) != bool(on):
            _lib.check(lib.atspeed_decoder_set_trace(d.handle, 1 if on else 0))
            d.trace_on = bool(on)


def _prompt_row(inputs) -> torch.Tensor:
    ids = inputs["input_ids"]
    if ids.dim() == 2:
        ids = ids[0]            # the reference reads batch row 0 only (beamSD.py:57,203,224)
    return ids


def _batch_buffers(prompts, k: int, max_new_tokens: int, dev):
    """One allocation / conversion for a whole lock-step batch instead of three small kernels per user: the prompts as ONE int32 buffer,
    the K x L token block and the K scores of every user as slices of two tensors."""
    lens = [int(p.numel()) for p in prompts]
    flat = torch.cat([p.reshape(-1) for p in prompts]).to(torch.int32)          # two kernels for the whole batch
    starts = [0]
    for n_tok in lens:
        starts.append(starts[-1] + n_tok)
    ids32 = [flat[s0: s0 + n_tok] for s0, n_tok in zip(starts, lens)]
    toks = torch.empty(len(prompts), k, max_new_tokens, dtype=torch.int32, device=dev)
    scores = torch.empty(len(prompts), k, dtype=torch.float32, device=dev)
    return ids32, toks, scores, (flat, starts)


def _batch_results(keep, toks: torch.Tensor, scores: torch.Tensor, k: int):
    """`beam_sequence` [k, P + L] int64 of every user (prompt ++ suffix per beam, beamSD.py:87,383) as views of ONE buffer written by one
    launch (atspeed_assemble_sequences) instead of a torch.cat per user."""
    flat, starts = keep
    n, L = toks.shape[0], toks.shape[2]
    off = np.asarray(starts, dtype=np.int64)
    # under the MODEL's device like every other library call: the library stages `off` through the ring of the thread's current device
    with torch.cuda.device(toks.device):
        out = torch.empty(k * (int(off[-1]) + n * L), dtype=torch.int64, device=toks.device)
        _lib.check(_lib.load().atspeed_assemble_sequences(flat.data_ptr(), off.ctypes.data, toks.data_ptr(), n, k, L, out.data_ptr(),
                                                          _lib.stream_ptr(toks.device)))
    res = []
    for i in range(n):
        o0, P = k * (int(off[i]) + i * L), int(off[i + 1] - off[i])
        res.append({"beam_sequence": out[o0: o0 + k * (P + L)].view(k, P + L), "beam_scores": scores[i]})
    return res


def _result(prompt: torch.Tensor, toks: torch.Tensor, scores: torch.Tensor, k: int) -> Dict:
    seq = torch.cat((prompt.to(torch.int64)[None, :].repeat(k, 1), toks.to(torch.int64)), dim=1)
    return {"beam_sequence": seq, "beam_scores": scores}


@Timer()
@torch.no_grad()
def BSSD(target_model, draft_model, inputs: Dict, gamma: int, max_new_tokens: int,
         logits_processor=None, prefix_allowed_tokens_fn=None, seed=None, trace_decisions: bool = False) -> Dict:
    _check_models(target_model, draft_model)
    mode = _sampling(target_model, seed)
    lib = _lib.load()
    dev = target_model.device
    prompt = _prompt_row(inputs).to(dev)
    P = int(prompt.numel())
    k = int(target_model.generation_config.num_beams)                 # beamSD.py:482
    dk = int(draft_model.generation_config.num_beams)                 # beamSD.py:483
    if _host_path(logits_processor, prefix_allowed_tokens_fn):
        # arbitrary Python callables (a mask closure, extra logits processors): served like the reference does, one host call per
        # beam per step (hostmask.py); stage times are wall clock with a device sync, like the reference's Timer
        from .hostmask import bssd_host_mask, bssd_host_mask_sample
        if mode[0]:                                   # do_sample with a host-side mask / processors: the draws happen on the host, from the device's streams
            if mode[1] <= 0.0:
                raise ValueError("sampling needs a temperature > 0")
            r = bssd_host_mask_sample(target_model, draft_model, prompt.cpu().numpy().astype(np.int64), int(gamma), int(max_new_tokens),
                                      prefix_allowed_tokens_fn, list(logits_processor or ()), mode[1], mode[2])
        else:
            r = bssd_host_mask(target_model, draft_model, prompt.cpu().numpy().astype(np.int64), int(gamma), int(max_new_tokens),
                               prefix_allowed_tokens_fn, list(logits_processor or ()))
        out = {"beam_sequence": torch.from_numpy(r["beam_sequence"]).to(dev), "beam_scores": torch.from_numpy(r["beam_scores"]).to(dev)}
        out.update({kk: r[kk] for kk in ("n_run", "total_accept_steps", "total_accept_tokens", "ave_accept_tokens", "accept_steps",
                                         "draft_time_cost", "target_time_cost", "verify_time_cost")})
        out["n_valid"] = int(len(r["beam_scores"]))
        return out
    # the mask functions look at the prompt (position of "Response:", data.py:97-102): one D2H copy per
    # user, where the reference does one per beam per step (generation_trie.py:94, data.py:98)
    fsm = _compile_constraint(prefix_allowed_tokens_fn, prompt.tolist())
    dfsm = _DeviceFSM.get(fsm, target_model.dims.vocab_size)
    dec = _Decoder.get(target_model, draft_model, P)
    _set_sampling([dec], mode)
    _set_trace([dec], trace_decisions)
    with torch.cuda.device(dev):
        ids32 = prompt.to(torch.int32).contiguous()
        toks = torch.empty(k, max_new_tokens, dtype=torch.int32, device=dev)
        scores = torch.empty(k, dtype=torch.float32, device=dev)
        stats = _lib.GenStats()
        _lib.check(lib.atspeed_bssd_generate(dec.handle, ids32.data_ptr(), P, dfsm.handle, fsm.start, int(gamma),
                                             int(max_new_tokens), k, dk, toks.data_ptr(), scores.data_ptr(),
                                             C.byref(stats), _lib.stream_ptr(dev)))
    out = _result(prompt, toks, scores, k)
    n_run = int(stats.n_run)
    total = int(stats.total_accept_steps)
    out.update({
        "n_run": n_run,                                              # beamSD.py:527-541
        "total_accept_steps": total,
        "total_accept_tokens": total * k,
        "ave_accept_tokens": total * k / n_run if n_run else 0.0,
        "draft_time_cost": stats.draft_ms * 1e-3,
        "target_time_cost": stats.target_ms * 1e-3,
        "verify_time_cost": stats.verify_ms * 1e-3,
        "device_time_cost": stats.total_ms * 1e-3,
        "accept_steps": [int(stats.accept_steps[i]) for i in range(min(n_run, _lib.MAX_NEW_TOKENS))],
        "n_valid": int(stats.n_valid),
        "n_target_forwards": int(stats.n_target_forwards),
        "n_draft_forwards": int(stats.n_draft_forwards),
    })
    return out


beam_sd_generate = BSSD
MAX_USERS_PER_CALL = 256


@torch.no_grad()
def BSSD_batch(target_model, draft_model, inputs_list, gamma: int, max_new_tokens: int,
               prefix_allowed_tokens_fn=None, seed=None, trace_decisions: bool = False):
    """BSSD for several independent users at once (one result dict per user, same keys as BSSD).

    The reference decodes users strictly one after another (inference.py:162-176).  Here the users advance in
    lock step: each has its own decoder (private KV caches and beam state) and every draft step / target
    verification of a round is ONE forward over the tokens of all users (`atspeed_bssd_generate_batch`), so the
    weights are streamed once per forward instead of once per user.  Token ids, n_matches and draft candidates
    are identical to calling BSSD() per user (scores agree to fp32 rounding: the GEMM tiling depends on the batch)."""
    _check_models(target_model, draft_model)
    mode = _sampling(target_model, seed)             # user u of the call draws from stream seed + u
    if len(inputs_list) > MAX_USERS_PER_CALL:            # the library batches up to 256 users per forward
        outs = []
        for i in range(0, len(inputs_list), MAX_USERS_PER_CALL):
            outs += BSSD_batch(target_model, draft_model, inputs_list[i:i + MAX_USERS_PER_CALL], gamma, max_new_tokens,
                               prefix_allowed_tokens_fn, seed=(mode[2] + i) if mode[0] else None, trace_decisions=trace_decisions)
        return outs
    lib = _lib.load()
    dev = target_model.device
    n = len(inputs_list)
    k = int(target_model.generation_config.num_beams)
    dk = int(draft_model.generation_config.num_beams)
    t0 = time.time()
    prompts = [_prompt_row(inp).to(dev) for inp in inputs_list]
    fsms = [_compile_constraint(prefix_allowed_tokens_fn, ids) for ids in _prompt_lists(prompts)]
    dfsm = _DeviceFSM.get(fsms[0], target_model.dims.vocab_size)
    for f in fsms[1:]:
        if f.row_ptr is not fsms[0].row_ptr:
            raise ValueError("BSSD_batch needs one shared constraint automaton (only the start node may differ per user)")
    decs = [_Decoder.get(target_model, draft_model, int(p.numel()), lane=i) for i, p in enumerate(prompts)]
    _set_sampling(decs, mode)
    _set_trace(decs, trace_decisions)
    with torch.cuda.device(dev):
        ids32, toks, scores, _keep = _batch_buffers(prompts, k, max_new_tokens, dev)
        stats = (_lib.GenStats * n)()
        arr_p = (C.c_void_p * n)
        _lib.check(lib.atspeed_bssd_generate_batch(
            arr_p(*[d.handle for d in decs]), n, arr_p(*[t.data_ptr() for t in ids32]),
            (C.c_int32 * n)(*[int(p.numel()) for p in prompts]), dfsm.handle, (C.c_int32 * n)(*[f.start for f in fsms]),
            int(gamma), int(max_new_tokens), k, dk, arr_p(*[t.data_ptr() for t in toks]),
            arr_p(*[t.data_ptr() for t in scores]), stats, _lib.stream_ptr(dev)))
    wall = time.time() - t0
    outs = []
    results = _batch_results(_keep, toks, scores, k)
    for i in range(n):
        st = stats[i]
        out = results[i]
        n_run, total = int(st.n_run), int(st.total_accept_steps)
        out.update({"n_run": n_run, "total_accept_steps": total, "total_accept_tokens": total * k,
                    "ave_accept_tokens": total * k / n_run if n_run else 0.0,
                    "draft_time_cost": st.draft_ms * 1e-3, "target_time_cost": st.target_ms * 1e-3,
                    "verify_time_cost": st.verify_ms * 1e-3, "device_time_cost": st.total_ms * 1e-3,
                    "time_cost": wall / n, "n_valid": int(st.n_valid), "status": int(st.status),
                    "accept_steps": [int(st.accept_steps[j]) for j in range(min(n_run, _lib.MAX_NEW_TOKENS))],
                    "n_target_forwards": int(st.n_target_forwards), "n_draft_forwards": int(st.n_draft_forwards)})
        outs.append(out)
    return outs


@Timer()
@torch.no_grad()
def target_generate(model, inputs: Dict, max_new_tokens: int, logits_processor=None,
                    prefix_allowed_tokens_fn=None, seed=None) -> Dict:
    _check_models(model)
    mode = _sampling(model, seed)
    lib = _lib.load()
    dev = model.device
    prompt = _prompt_row(inputs).to(dev)
    P = int(prompt.numel())
    k = int(model.generation_config.num_beams)                        # beamSD.py:553
    if _host_path(logits_processor, prefix_allowed_tokens_fn):
        from .hostmask import target_generate_host_mask, target_generate_host_mask_sample
        if mode[0]:
            if mode[1] <= 0.0:
                raise ValueError("sampling needs a temperature > 0")
            r = target_generate_host_mask_sample(model, prompt.cpu().numpy().astype(np.int64), int(max_new_tokens), prefix_allowed_tokens_fn,
                                                 list(logits_processor or ()), mode[1], mode[2])
        else:
            r = target_generate_host_mask(model, prompt.cpu().numpy().astype(np.int64), int(max_new_tokens), prefix_allowed_tokens_fn,
                                          list(logits_processor or ()))
        return {"beam_sequence": torch.from_numpy(r["beam_sequence"]).to(dev), "beam_scores": torch.from_numpy(r["beam_scores"]).to(dev),
                "n_valid": int(len(r["beam_scores"]))}
    fsm = _compile_constraint(prefix_allowed_tokens_fn, prompt.tolist())
    dfsm = _DeviceFSM.get(fsm, model.dims.vocab_size)
    dec = _Decoder.get(model, None, P)
    _set_sampling([dec], mode)
    with torch.cuda.device(dev):
        ids32 = prompt.to(torch.int32).contiguous()
        toks = torch.empty(k, max_new_tokens, dtype=torch.int32, device=dev)
        scores = torch.empty(k, dtype=torch.float32, device=dev)
        stats = _lib.GenStats()
        _lib.check(lib.atspeed_target_generate(dec.handle, ids32.data_ptr(), P, dfsm.handle, fsm.start,
                                               int(max_new_tokens), k, toks.data_ptr(), scores.data_ptr(),
                                               C.byref(stats), _lib.stream_ptr(dev)))
    out = _result(prompt, toks, scores, k)
    out.update({"n_valid": int(stats.n_valid), "device_time_cost": stats.total_ms * 1e-3})
    return out


def target_generate_batch(model, inputs_list, max_new_tokens: int, prefix_allowed_tokens_fn=None, seed=None):
    """`target_generate` for several independent users in lock step (one result dict per user): position g of every
    user's constrained beam search is ONE forward.  This is the loop `code/generate_teacher_data.py:211-244` runs over
    a whole training set with HF `generate`; token ids equal per-user `target_generate` calls."""
    _check_models(model)
    mode = _sampling(model, seed)
    if len(inputs_list) > MAX_USERS_PER_CALL:
        outs = []
        for i in range(0, len(inputs_list), MAX_USERS_PER_CALL):
            outs += target_generate_batch(model, inputs_list[i:i + MAX_USERS_PER_CALL], max_new_tokens, prefix_allowed_tokens_fn,
                                          seed=(mode[2] + i) if mode[0] else None)
        return outs
    lib = _lib.load()
    dev = model.device
    n = len(inputs_list)
    k = int(model.generation_config.num_beams)
    t0 = time.time()
    prompts = [_prompt_row(inp).to(dev) for inp in inputs_list]
    fsms = [_compile_constraint(prefix_allowed_tokens_fn, ids) for ids in _prompt_lists(prompts)]
    dfsm = _DeviceFSM.get(fsms[0], model.dims.vocab_size)
    for f in fsms[1:]:
        if f.row_ptr is not fsms[0].row_ptr:
            raise ValueError("target_generate_batch needs one shared constraint automaton (only the start node may differ per user)")
    decs = [_Decoder.get(model, None, int(p.numel()), lane=i) for i, p in enumerate(prompts)]
    _set_sampling(decs, mode)
    with torch.cuda.device(dev):
        ids32, toks, scores, _keep = _batch_buffers(prompts, k, max_new_tokens, dev)
        stats = (_lib.GenStats * n)()
        arr_p = (C.c_void_p * n)
        _lib.check(lib.atspeed_target_generate_batch(
            arr_p(*[d.handle for d in decs]), n, arr_p(*[t.data_ptr() for t in ids32]),
            (C.c_int32 * n)(*[int(p.numel()) for p in prompts]), dfsm.handle, (C.c_int32 * n)(*[f.start for f in fsms]),
            int(max_new_tokens), k, arr_p(*[t.data_ptr() for t in toks]), arr_p(*[t.data_ptr() for t in scores]), stats,
            _lib.stream_ptr(dev)))
    wall = time.time() - t0
    outs = []
    results = _batch_results(_keep, toks, scores, k)
    for i in range(n):
        out = results[i]
        out.update({"n_valid": int(stats[i].n_valid), "status": int(stats[i].status), "device_time_cost": stats[i].total_ms * 1e-3,
                    "time_cost": wall / n})
        outs.append(out)
    return outs


def last_trace(target_model, draft_model):
    """Per-round trace of the last BSSD call on this model pair (parity tests):
    list of dict(draft_len, n_matches, n_beams, draft_ids=[draft_len][dk])."""
    lib = _lib.load()
    dec = _Decoder._cache[(id(target_model), id(draft_model) if draft_model is not None else None, 0)]
    n = lib.atspeed_decoder_trace(dec.handle, None, 0)
    buf = (C.c_int32 * max(n, 1))()
    lib.atspeed_decoder_trace(dec.handle, buf, n)
    dk = int(draft_model.generation_config.num_beams)
    rounds, i = [], 0
    while i < n:
        dl, nm, nb = buf[i], buf[i + 1], buf[i + 2]
        i += 3
        ids = [[int(buf[i + s * dk + j]) for j in range(dk)] for s in range(dl)]
        i += dl * dk
        rounds.append(dict(draft_len=int(dl), n_matches=int(nm), n_beams=int(nb), draft_ids=ids))
    return rounds


def last_decisions(target_model, draft_model, lane: int = 0):
    """Decision trace of the last BSSD / BSSD_batch call made with `trace_decisions=True` for the user in `lane`
    (`atspeed_decoder_decisions`): one dict per round with the beams the round started from, the draft's blocks, the target's picks of
    every verify step (the decisions of beamSD.py:297-298,323-328) and n_matches; a final single step (beamSD.py:505-509) is a round
    of kind "final".  Sequences are the generated suffixes (lists of token ids), scores fp32."""
    lib = _lib.load()
    dec = _Decoder._cache[(id(target_model), id(draft_model) if draft_model is not None else None, lane)]
    n = int(lib.atspeed_decoder_decisions(dec.handle, None, 0))
    buf = np.zeros(max(n, 1), dtype=np.int32)
    lib.atspeed_decoder_decisions(dec.handle, buf.ctypes.data, n)
    B, L, G1 = _lib.MAX_BEAMS, _lib.MAX_NEW_TOKENS, _lib.MAX_GAMMA + 1
    bw = 5 * B + B * L

    def block(words, n_valid, seq_len):
        sc = words[:B].view(np.float32)
        parent, tok, flat = words[2 * B: 3 * B], words[3 * B: 4 * B], words[4 * B: 5 * B]
        seq = words[5 * B:].reshape(B, L)
        rows = [j for j in range(n_valid) if flat[j] >= 0]
        return dict(score=[float(sc[j]) for j in rows], seq=[[int(t) for t in seq[j, :seq_len]] for j in rows],
                    parent=[int(parent[j]) for j in rows], tok=[int(tok[j]) for j in rows], index=rows)

    rounds, i = [], 0
    while i < n:
        kind, nb, dl, nm, gen0, k, dk, nblk = (int(x) for x in buf[i: i + 8])
        i += 8
        blocks = [buf[i + b * bw: i + (b + 1) * bw] for b in range(nblk)]
        i += nblk * bw
        if kind == 1:
            rounds.append(dict(kind="final", gen0=gen0, k=k, parents=block(blocks[0], nb, gen0), result=block(blocks[1], k, gen0 + 1)))
            continue
        vt = buf[i: i + G1 * 3 * B].reshape(G1, 3, B)
        i += G1 * 3 * B
        start = block(blocks[0], nb, gen0)
        draft = [block(blocks[s], dk, gen0 + s) for s in range(1, dl + 1)]
        picks = []
        for s in range(nm + 1):
            src_full = blocks[s][5 * B:].reshape(B, L)
            sc, par, tok = vt[s, 0].view(np.float32), vt[s, 1], vt[s, 2]
            rows = [j for j in range(k) if tok[j] >= 0]
            picks.append(dict(score=[float(sc[j]) for j in rows], parent=[int(par[j]) for j in rows], tok=[int(tok[j]) for j in rows],
                              seq=[[int(t) for t in src_full[par[j], :gen0 + s]] + [int(tok[j])] for j in rows]))
        rounds.append(dict(kind="verify", gen0=gen0, k=k, dk=dk, draft_len=dl, n_matches=nm, start=start, draft=draft, picks=picks,
                           result=block(blocks[dl + 1], k, gen0 + nm + 1)))
    return rounds
