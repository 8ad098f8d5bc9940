#!/usr/bin/env python3
"""Headline benchmark of the beam-speculative-decoding hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Metric (BASELINE.json): recommended items/sec + mean accepted length, Llama-68M draft /
Llama-7B target, K=20 beams, DK=40 draft beams, gamma=4, L=4 code tokens, Beauty vocabulary
(V=32859), bf16, synthetic hash-PRNG weights and prompts (no tokenizer/checkpoints offline).
A "step" is one pass of the hot path over one batch of inputs: a lock-step batch of `--streams`
users (default 256), each running its complete BSSD() (draft steps + packed target verification +
verify rounds), prompts already resident in HBM; `value` = users * K beams / wall time.
Users are independent, so N GPUs shard the user list (weak scaling: --steps batches per GPU)
and exchange only one all-gather of counters at the end.

Extra objects on the JSON line (N = 1; all driver-clocked, all bounded so the default run stays within minutes):
  roofline     — the dominant kernel (the target forward's gate_up projection GEMM), hipEvent-bracketed on its
                 launch stream: algorithmic flops / launch time vs the 2.5 PF dense bf16 MFMA peak when users are
                 batched; verify_scan: the verify step's scan vs 8 TB/s.
  configs      — BASELINE configs 3 and 5 as sub-passes with their own items/s, ms_per_step and dominant-kernel roofline:
                 games_256 / games_256_trie (Games V=33014, 256 users per lock-step batch, position-set mask / strict item
                 trie), fp16 (the reference's own dtype on the engine's fp16 flavour) and fp8 (e4m3 W8A8 target projections, roofline
                 against 5 PF, accepted-length drift vs bf16).
  aligned_weight_brackets — the same users and kernels with draft / target weights that agree (accept length 3 and ~1.1), each
                 with the fp32 ENGINE's accepted length on the same weights and >= 32 users next to the bf16 engine's.
  speedup_curve — the reference's own figure of merit (inference.py:179: speedup = target_generate time / BSSD time) on this engine at
                 1 / 4 / 16 / 64 / 256 users per lock-step batch: items/s and ms to the last result of both decoders, speedup, and per point
                 the fraction of max(weight stream at 8 TB/s, matrix flops at 2.5 PF) over the forwards the call really ran (engine log);
                 the aligned brackets carry the same curve at accept length ~1.1 and 3.  latency_curve = its BSSD leg.
  single_user_stream — the reference's one-user-at-a-time loop (HBM-bound projections).
  cpu_baseline — the oracle (oracle/beamsd_ref.py, torch-CPU fp32) timed on the host cores on a
                 bounded sample of the same workload with the same weights (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from atspeed_amd import synth                      # noqa: E402
from atspeed_amd.beamSD import BSSD, BSSD_batch, release_decoders, target_generate, target_generate_batch    # noqa: E402
from atspeed_amd.dist import Counters, aggregate, all_gather_counters   # noqa: E402
from atspeed_amd.generation_trie import PositionSetConstraint   # noqa: E402
from atspeed_amd.model import HipLlama             # noqa: E402

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy)
MFMA_PEAK_TFLOPS = 2500.0    # same guide: ~2.5 PF dense bf16
MFMA_FP8_PEAK_TFLOPS = 5000.0    # same guide: ~5 PF dense fp8 (block-scaled v_mfma_scale_f32_*_f8f6f4; the non-scaled fp8 forms run at the bf16 rate)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed steps per GPU; a step = one lock-step batch of --streams users")
    ap.add_argument("--warmup", type=int, default=1, help="untimed steps (batches) before the timed region")
    ap.add_argument("--target-layers", type=int, default=32, help="32 = Llama-7B (the metric's config)")
    ap.add_argument("--beam", type=int, default=20)
    ap.add_argument("--draft-beam", type=int, default=40)
    ap.add_argument("--gamma", type=int, default=4)
    ap.add_argument("--new-tokens", type=int, default=4)
    ap.add_argument("--seed", type=int, default=2025)
    ap.add_argument("--streams", type=int, default=256, help="users decoded in lock step per GPU (one batched forward per draft step / verification); 1 = the reference's one-user-at-a-time loop")
    ap.add_argument("--target-fp8", action="store_true", help="BASELINE config 5 as the HEADLINE pass: fp8 (e4m3 W8A8) target projections in the batched forwards")
    ap.add_argument("--single-stream-users", type=int, default=6, help="extra untimed-for-value pass: users decoded one at a time (the reference's loop)")
    ap.add_argument("--aligned-resid-scale", type=str, default="3e-6,3e-5",
                    help="extra brackets (SURVEY.md 8d 'oracle-draft'): same shapes and kernels, draft/target weights aligned through a "
                         "shared bigram table with the layers' residual contributions scaled by each factor of this comma list "
                         "(3e-6: every draft step accepted, 3e-5: about one step per verification); empty string skips the pass")
    ap.add_argument("--one-user-fp8-users", type=int, default=24, help="users per bracket decoded ONE AT A TIME with the bf16 and then the W8A8 target (config 5 at the reference's batch-1 shape)")
    ap.add_argument("--aligned-fp32-users", type=int, default=64, help="users per aligned bracket the fp32 ENGINE also decodes (accepted length next to the bf16 engine's)")
    ap.add_argument("--dataset", choices=("beauty", "games"), default="beauty", help="vocabulary / prompt-length shape of the headline pass (games = BASELINE config 3)")
    ap.add_argument("--mask", choices=("position", "trie"), default="position",
                    help="position = the per-position allowed sets inference.py installs; trie = strict item trie on the generated suffix")
    ap.add_argument("--do-sample", action="store_true", help="sampling-mode beam-SD (generation_config.do_sample) instead of the greedy headline")
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--cpu-baseline-users", type=int, default=4, help="users the CPU oracle decodes (about 7 s each on a one-GPU box's 16 allotted CPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sub-steps", type=int, default=2, help="timed batches of each sub-pass (configs 3 / 5, aligned brackets); independent of --steps so the line stays bounded")
    ap.add_argument("--no-configs", action="store_true", help="skip the config 3 / config 5 sub-passes")
    ap.add_argument("--no-latency-curve", action="store_true")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ roofline.traffic provenance
TRAFFIC_FILE = os.path.join("profiles", "r06_pmc_traffic.json")
KERNEL_SOURCES = ("gemm.hip", "common.h")          # the GEMM kernels and the device helpers they use (internal.h holds only host-side declarations for them)


def kernel_sha(root: str = ROOT) -> str:
    """Identity of the GEMM kernels' source: sha256 over gemm.hip + common.h (first 16 hex digits)."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(root, "atspeed_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def traffic_from_profiles(kind: str, path: str = None, sha: str = None):
    """(HBM bytes per launch, provenance) from the committed PMC summary.  The counters need their own rocprofv3 --pmc passes (FETCH_SIZE
    and WRITE_SIZE do not fit one pass), so this number is NOT measured by the run that prints it: the line carries `traffic_source`, and
    a summary recorded for OTHER kernel sources than the ones in this tree (`kernel_sha`) is refused -- the number would be stale."""
    p = path or os.path.join(ROOT, TRAFFIC_FILE)
    if not os.path.exists(p):
        return None, None
    try:
        with open(p) as f:
            d = json.load(f)
    except Exception:
        return None, None
    v = d.get(kind, {}).get("hbm_bytes_per_launch")
    if v is None:
        return None, None
    have, want = d.get("kernel_sha"), (sha or kernel_sha())
    rel = os.path.relpath(p, ROOT)
    if have != want:
        return None, f"{rel} refused: recorded for kernel sources {have}, this tree has {want} (re-run tools/profile_round.sh)"
    return v, (f"{rel} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload at commit {d.get('commit', '?')}, avg M "
               f"{d.get(kind, {}).get('avg_m', '?')}, same kernel sources {have}; collected offline, not in this run)")


# ------------------------------------------------------------------------------------------------ CPU oracle leg
def cpu_baseline(target, draft, prompts, fn, args, n_users=None):
    """Oracle (CPU restatement of the reference) on the same weights/prompts; bounded sample.  Returns (object, oracle outputs, oracle models)."""
    from oracle import beamsd_ref as R, use_allotted_cpu_threads
    from oracle.llama_ref import RefLlama
    cores = use_allotted_cpu_threads()              # the CPUs this process really has (cgroup quota), not the host's core count
    t0 = time.perf_counter()
    rt = RefLlama(target.dims, target.export_state_dict(), max_slots=512)
    rd = RefLlama(draft.dims, draft.export_state_dict(), max_slots=512)
    setup = time.perf_counter() - t0
    n = min(args.cpu_baseline_users if n_users is None else n_users, len(prompts))
    t0 = time.perf_counter()
    acc = runs = 0
    outs = []
    for u in range(n):
        o = R.BSSD(rt, rd, prompts[u], args.gamma, args.new_tokens, args.beam, args.draft_beam, fn)
        acc += o["total_accept_steps"]
        runs += o["n_run"]
        outs.append(o)
    dt = time.perf_counter() - t0
    return dict(value=n * args.beam / dt, unit="items/s", cores=cores, kind="port",
                sample=f"{n} user(s) of the same workload (same weights, fp32 on CPU), {dt:.1f}s after {setup:.1f}s weight export",
                mean_accept_len=(acc / runs if runs else 0.0)), outs, (rt, rd)


def oracle_scores_of(ref_model, prompt, seqs):
    """Beam scores the fp32 oracle gives to arbitrary generated sequences: sum of the full-vocabulary log-probabilities of their tokens
    (what beamSD.py:58,69-70 accumulate), from ONE packed forward (prompt once, every sequence a branch under a tree mask)."""
    P, L, n = len(prompt), len(seqs[0]), len(seqs)
    ids = list(int(t) for t in prompt) + [int(t) for sq in seqs for t in sq[:-1]]
    T = len(ids)
    pos = list(range(P)) + [P + j for _ in seqs for j in range(L - 1)]
    vis = torch.zeros(T, T, dtype=torch.bool)
    vis[:P, :P] = torch.tril(torch.ones(P, P, dtype=torch.bool))
    for i in range(n):
        lo = P + i * (L - 1)
        vis[lo: lo + L - 1, :P] = True
        vis[lo: lo + L - 1, lo: lo + L - 1] = torch.tril(torch.ones(L - 1, L - 1, dtype=torch.bool))
    logp = torch.log_softmax(ref_model.forward(ids, pos, list(range(T)), vis, n_logit_rows=T - P + 1), dim=-1)   # row 0 = last prompt token
    out = []
    for i, sq in enumerate(seqs):
        rows = [0] + [1 + i * (L - 1) + j for j in range(L - 1)]
        out.append(float(sum(logp[r, int(t)] for r, t in zip(rows, sq))))
    return out


def disagreement_report(gpu_out, ref_out, P, ref_target, prompt):
    """Where the bf16 engine's ranking of a user differs from the fp32 oracle's: per rank, the oracle's own score of both items.  The
    ASSERTED form of this statement is tests/test_decisions_gpu.py (every decision of 64 users replayed by the fp32 engine, margin-gated);
    this is the same evidence for the headline user, next to the timing."""
    g_items = [tuple(x) for x in gpu_out["beam_sequence"][:, P:].cpu().tolist()]
    r_items = [tuple(x) for x in ref_out["beam_sequence"][:, P:].tolist()]
    g_scores = [float(x) for x in gpu_out["beam_scores"].cpu().tolist()]
    r_scores = [float(x) for x in ref_out["beam_scores"].tolist()]
    ranks = [i for i, (a, b) in enumerate(zip(g_items, r_items)) if a != b]
    rep = dict(top_k_overlap=len(set(g_items) & set(r_items)) / max(1, len(r_items)), ranks_that_differ=ranks)
    if not ranks:
        return rep
    ours = sorted({g_items[i] for i in ranks})
    sc = dict(zip(ours, oracle_scores_of(ref_target, prompt, [list(x) for x in ours])))
    rows = [dict(rank=i, oracle_score_of_oracle_item=r_scores[i], oracle_score_of_gpu_item=sc[g_items[i]],
                 gap=r_scores[i] - sc[g_items[i]], bf16_noise_on_gpu_item=abs(g_scores[i] - sc[g_items[i]])) for i in ranks]
    rep.update(per_rank=rows, max_gap=max(abs(r["gap"]) for r in rows), max_bf16_noise=max(r["bf16_noise_on_gpu_item"] for r in rows),
               oracle_kth_score_margin=(r_scores[-2] - r_scores[-1]) if len(r_scores) > 1 else None,
               asserted_by="tests/test_decisions_gpu.py (64 users, every top-K / top-DK decision judged by the fp32 engine; clear margins identical)",
               note="gap = how much worse (by the fp32 oracle's own arithmetic) the item the bf16 engine put at this rank is than the oracle's item there")
    return rep



# ------------------------------------------------------------------------------------------------ the ONE line the driver parses
LINE_BUDGET_BYTES = 6144        # the driver keeps an 8 KB tail of stdout: r03's 14 KB line parsed, r04's 41 KB line did not (BENCH_r04.parsed = null)
DETAIL_FILE = "bench_detail.json"


def _num(x, sig=5):
    """numbers of the compact line: 5 significant digits (floats), ints as they are"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _num(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, sig) for v in x]
    return x


def _pick(d, keys):
    return {k: d.get(k) for k in keys if d is not None and k in d} if d else None


def _curve_rows(curve, bracket):
    """numbers-only rows of a speedup curve: users, bracket, ms of both decoders, speedup, roofline fraction of both"""
    rows = []
    for p_ in curve or []:
        b, t = p_.get("bssd", {}), p_.get("target_generate", {})
        rows.append(dict(users=p_.get("users_per_batch"), bracket=bracket, bssd_ms=b.get("ms_to_last_result"), tg_ms=t.get("ms_to_last_result"),
                         speedup=p_.get("speedup"), bssd_frac=(b.get("roofline") or {}).get("frac"), tg_frac=(t.get("roofline") or {}).get("frac")))
    return rows


ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "avg_launch_us", "launches", "avg_m",
                 "algorithmic_bytes_per_launch", "algorithmic_flops_per_launch", "peak_measured", "frac_of_measured")


def compact_line(detail: dict) -> dict:
    """The final stdout line: the contract's keys, `roofline`, `cpu_baseline`, `measured_peaks` and a NUMBERS-ONLY summary of every sub-pass
    (inference.py:152-156,183-189 emit a CSV row of numbers, not a report).  Everything else -- `what` strings, per-point forward logs, the
    per-rank disagreement rows -- lives in DETAIL_FILE.  tests/test_bench_contract.py holds this to LINE_BUDGET_BYTES."""
    d = detail
    line = {k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                  "vs_baseline", "dtype", "data")}
    cfg = d.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "users_per_step", "users_per_gpu", "mean_prompt_len", "parallelism", "kernel_sha"))
    line["mean_accept_len"] = d.get("mean_accept_len")
    if d.get("decoding"):
        line["decoding"] = d["decoding"]
    rf = d.get("roofline") or {}
    r = _pick(rf, ROOFLINE_KEYS)
    if r and isinstance(r.get("kernel"), str):
        r["kernel"] = r["kernel"].split(" (SK =")[0][:120]
    line["roofline"] = r
    line["measured_peaks"] = _pick(d.get("measured_peaks"), ("mfma_bf16_tflops", "hbm_read_gbs"))
    line["cpu_baseline"] = _pick(d.get("cpu_baseline"), ("value", "unit", "cores", "kind", "sample", "mean_accept_len", "gpu_accept_len_same_users",
                                                         "top_k_overlap_with_gpu_bf16"))
    line["per_user"] = d.get("per_user")
    if d.get("n_gpus", 1) > 1:
        line["per_rank"] = [_pick(r_, ("rank", "n_users", "n_run", "accept_steps", "elapsed_ms")) for r_ in d.get("per_rank") or []]
    vs = d.get("verify_scan")
    line["verify_scan"] = _pick(vs, ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_us", "bytes_per_launch"))
    cf = {}
    for name, c in (d.get("configs") or {}).items():
        cf[name] = dict(items_per_s=c.get("items_per_s"), ms_per_step=c.get("ms_per_step"), mean_accept_len=c.get("mean_accept_len"),
                        frac=(c.get("roofline") or {}).get("frac"), bound=(c.get("roofline") or {}).get("bound"))
        dr = c.get("accepted_length_drift_vs_bf16")
        if dr:
            cf[name]["accept_drift_vs_bf16"] = dr.get("drift_steps")
    line["configs"] = cf or None
    rows = _curve_rows(d.get("speedup_curve"), "accept0")
    br = []
    for a in d.get("aligned_weight_brackets") or []:
        tag = f"resid{a.get('resid_scale'):g}"
        rows += _curve_rows(a.get("speedup_curve"), tag)
        f32 = a.get("fp32_engine") or {}
        br.append(dict(bracket=tag, items_per_s=a.get("items_per_s"), mean_accept_len=a.get("mean_accept_len"),
                       fp32_engine_accept_len=f32.get("mean_accept_len"), users_equal_accept_steps=f32.get("users_with_equal_accept_steps"),
                       fp32_users=f32.get("users")))
    line["aligned_weight_brackets"] = br or None
    line["speedup_curve"] = rows or None          # speedup = plain beam search ms / beam-SD ms (inference.py:179)
    su = d.get("single_user_stream")
    if su:
        line["single_user_stream"] = dict(_pick(su, ("users", "items_per_s", "ms_per_user", "weights_stream_frac_of_hbm_peak")),
                                          frac=(su.get("roofline") or {}).get("frac"), avg_launch_us=(su.get("roofline") or {}).get("avg_launch_us"))
    ou = ((d.get("configs") or {}).get("fp8") or {}).get("one_user")
    if ou or any(a.get("one_user") for a in d.get("aligned_weight_brackets") or []):
        # config 5 at the reference's batch-1 shape: ms per user with the bf16 / the W8A8 target, by acceptance bracket
        rows8 = [dict(bracket="accept0", bf16_ms=ou.get("bf16_ms_per_user"), fp8_ms=ou.get("ms_per_user"), fp8_frac=ou.get("weights_stream_frac_of_hbm_peak"),
                      not_fp8=ou.get("projections_not_in_fp8"))] if ou else []
        for a in d.get("aligned_weight_brackets") or []:
            o1 = a.get("one_user")
            if o1:
                rows8.append(dict(bracket=f"resid{a.get('resid_scale'):g}", bf16_ms=o1.get("bf16_ms_per_user"), fp8_ms=o1.get("fp8_ms_per_user"),
                                  fp8_frac=o1.get("fp8_weights_stream_frac_of_hbm_peak"), accept_bf16=o1.get("bf16_mean_accept_len"),
                                  accept_fp8=o1.get("fp8_mean_accept_len"), not_fp8=o1.get("projections_not_in_fp8")))
        o16 = ((d.get("configs") or {}).get("fp8_fp16") or {}).get("one_user")
        if o16:                                   # the reference's dtype combination: fp16 models, W8A8 target (inference.py:75-91)
            rows8.append(dict(bracket="accept0_fp16", fp8_ms=o16.get("ms_per_user"), fp8_frac=o16.get("weights_stream_frac_of_hbm_peak")))
        line["one_user_fp8"] = rows8
    errs = d.get("sub_pass_errors")
    line["sub_pass_errors"] = {k: str(v)[:120] for k, v in errs.items()} if errs else None
    line["detail"] = d.get("detail_file")
    return _num(line)


def emit(detail: dict, out_dir: str = None):
    """Write the full report to DETAIL_FILE (under gpurun_out/ when that exists, so it comes back from the GPU box) and print the compact
    line as the LAST line of stdout."""
    out_dir = out_dir or (os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else ROOT)
    path = os.path.join(out_dir, DETAIL_FILE)
    try:
        with open(path, "w") as f:
            json.dump(detail, f)
        detail["detail_file"] = os.path.relpath(path, ROOT)
    except OSError as e:
        detail["detail_file"] = f"not written: {e}"
    text = json.dumps(compact_line(detail), separators=(",", ":"))
    if len(text) > LINE_BUDGET_BYTES:            # never let a sub-pass summary cost the headline: drop the widest optional objects
        slim = compact_line(detail)
        for k in ("speedup_curve", "aligned_weight_brackets", "configs", "verify_scan", "single_user_stream", "per_user"):
            slim.pop(k, None)
            text = json.dumps(slim, separators=(",", ":"))
            if len(text) <= LINE_BUDGET_BYTES:
                break
    print(text, flush=True)
    return text


# ------------------------------------------------------------------------------------------------ launch plumbing
def launcher_command(n_gpus: int, argv, port: int = 0):
    """The torch.distributed.run command line that starts `n_gpus` ranks of this script (one process per GPU, RCCL rendezvous on
    127.0.0.1).  What `python bench.py --gpus N` runs when it was not started under torchrun itself."""
    port = port or 29500 + (os.getpid() % 2000)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def resolve_world(args, env=os.environ):
    """-> ("run", rank, local_rank, world) or ("spawn", n).  The rank count comes from --gpus; under torchrun WORLD_SIZE must agree with it
    (a line that says n_gpus = 1 for a --gpus 8 request would be a silent lie).  Nothing here touches the GPU."""
    have_env = "WORLD_SIZE" in env and "RANK" in env
    if not have_env:
        return ("spawn", args.gpus) if args.gpus > 1 else ("run", 0, 0, 1)
    world = int(env["WORLD_SIZE"])
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start exactly --gpus ranks "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus} ...)")
    return ("run", int(env["RANK"]), int(env.get("LOCAL_RANK", "0")), world)


def per_rank_report(per_rank, beam: int):
    """Straggler visibility of an N > 1 line: what every rank decoded and how long it took (the job's time is the slowest rank's)."""
    return [dict(rank=i, n_users=c.n_users, n_run=c.n_run, accept_steps=c.accept_steps, elapsed_ms=c.elapsed_ns * 1e-6,
                 items_per_s=(c.n_users * beam / (c.elapsed_ns * 1e-9)) if c.elapsed_ns else 0.0) for i, c in enumerate(per_rank)]


# ------------------------------------------------------------------------------------------------ workload pieces
def make_mask(vocab, kind: str):
    if kind == "trie":
        from atspeed_amd.generation_trie import SuffixTrieConstraint, Trie
        return SuffixTrieConstraint(Trie([[1] + [int(t) for t in it] + [2] for it in synth.synthetic_items(vocab)]), synth.RESPONSE_SEP, 1)
    return PositionSetConstraint(vocab.allowed_tokens(), synth.RESPONSE_SEP)


def make_prompts(n, first, seed, dataset, dev):
    plens = synth.prompt_lengths(first + n, seed, mean_hist=7.33 if dataset == "beauty" else 5.98)
    prompts = [synth.synthetic_prompt(int(plens[first + u]), synth.tensor_seed(seed, f"user{first + u}")) for u in range(n)]
    return prompts, [{"input_ids": torch.from_numpy(p)[None].to(dev)} for p in prompts]   # resident in HBM before timing


def run_users(target, draft, dprompts, lo, hi, streams, fn, args):
    """users [lo, hi) in lock-step batches of `streams` (1 = plain per-user BSSD calls, the reference's loop)"""
    res = []
    if streams <= 1:
        for u in range(lo, hi):
            res.append(BSSD(target, draft, dprompts[u], args.gamma, args.new_tokens, prefix_allowed_tokens_fn=fn))
        return res
    for g in range(lo, hi, streams):
        res += BSSD_batch(target, draft, dprompts[g:min(hi, g + streams)], args.gamma, args.new_tokens, prefix_allowed_tokens_fn=fn)
    return res


def one_user_loop(target, draft, dprompts, first, n, fn, args, dev, bytes_per_weight):
    """The reference's own loop (inference.py:162-176): `n` users decoded strictly one at a time.  -> ms per user, accepted length, and the
    weight stream of the target's forwards against 8 TB/s (the roofline of a forward at 20-230 tokens: its weights read once)."""
    grp = dprompts[first: first + n]
    for p_ in grp[:2]:
        BSSD(target, draft, p_, args.gamma, args.new_tokens, prefix_allowed_tokens_fn=fn)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    outs = [BSSD(target, draft, p_, args.gamma, args.new_tokens, prefix_allowed_tokens_fn=fn) for p_ in grp]
    torch.cuda.synchronize(dev)
    sec = (time.perf_counter() - t0) / max(1, len(grp))
    tf = sum(o["n_target_forwards"] for o in outs) / max(1, len(grp))
    d = target.dims
    w_layers = d.n_layers * (4 * d.hidden * d.hidden + 3 * d.hidden * d.ffn)
    stream_bytes = tf * (w_layers * bytes_per_weight + d.hidden * d.vocab_size * 2)          # the lm_head stays 16-bit
    return dict(users=len(grp), ms_per_user=1e3 * sec, items_per_s=args.beam / sec,
                mean_accept_len=sum(o["total_accept_steps"] for o in outs) / max(1, sum(o["n_run"] for o in outs)),
                target_forwards_per_user=tf, weights_stream_gbs=stream_bytes / sec / 1e9,
                weights_stream_frac_of_hbm_peak=stream_bytes / sec / 1e9 / HBM_PEAK_GBS), outs


def gemm_roofline(target, prof, prof_big, fp8: bool, measured, streams: int, with_traffic: bool):
    """Roofline object of the GEMM kind with the largest total time.  One user at a time (M ~ 100-230 tokens) a launch is a single pass
    over the weights: HBM-bound.  With lock-step batching M = tokens of all users (thousands): arithmetic intensity is far above the ridge
    (2.5 PF / 8 TB/s = 312 flop/B) and the bound is the MFMA peak of the arithmetic type."""
    kind = max(prof, key=lambda k: prof[k]["ms"])
    one_kernel = prof_big[kind]["count"] > 0          # launches of >= 1024 tokens: exactly the 256x256 ring kernel
    pk = prof_big[kind] if one_kernel else prof[kind]
    N, K = target.gemm_shape(kind)
    avg_m = pk["rows"] / max(1, pk["count"])
    n_out = N // 2 if kind == "gate_up" else N
    out_b = 4 if kind == "lm_head" else 2
    f8 = fp8 and kind != "lm_head"
    in_b = 1 if f8 else 2                             # operand bytes: e4m3 or bf16
    alg_bytes = N * K * in_b + avg_m * K * in_b + avg_m * n_out * out_b
    alg_flops = 2.0 * avg_m * N * K
    avg_ms = pk["ms"] / max(1, pk["count"])
    gemm_ms_total = sum(v["ms"] for v in prof.values())
    intensity = alg_flops / alg_bytes
    if intensity >= MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
        achieved = alg_flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        bound, peak, unit = "mfma", (MFMA_FP8_PEAK_TFLOPS if f8 else MFMA_PEAK_TFLOPS), "TFLOP/s"
    else:
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        bound, peak, unit = "hbm", HBM_PEAK_GBS, "GB/s"
    # qkv: 5 = RoPE + KV scatter in the epilogue (head_dim 128 targets), as the engine's counter says
    epi = {"qkv": 5 if target.rope_fused_launches() > 0 else 0, "o_proj": 2, "gate_up": 3, "down": 2, "lm_head": 1}[kind]
    ring = (f"gemm_ring_mx_kernel<{epi}, 8>" if (f8 and os.environ.get("ATSPEED_FP8_MX", "1") != "0")
            else f"gemm_ring_kernel<{4 if kind == 'lm_head' else epi}, 8, {'true' if f8 else 'false'}, false, 4, SK> (SK = false | true: the same kernel without / with "
                 f"its split-K tail, chosen per launch by the tile count; rocprofv3 lists the two instantiations separately)")
    kname = (f"{ring} [{kind}] N={N} K={K} avg_M={avg_m:.0f} (launches of >= 1024 tokens)"
             if one_kernel else f"projection GEMM [{kind}] N={N} K={K} avg_M={avg_m:.0f} (all launches)")
    # PMC traffic was collected on the bf16 headline workload: it says nothing about the fp8 kernels or other batch shapes
    traffic, traffic_source = traffic_from_profiles(kind) if (with_traffic and one_kernel and not fp8 and streams == 256) else (None, None)
    m_peak = None
    if measured:
        m_peak = (measured["mfma_bf16_tflops"] if not f8 else None) if bound == "mfma" else measured["hbm_read_gbs"]
    return dict(bound=bound, kernel=kname, achieved=achieved, peak=peak, unit=unit, frac=achieved / peak,
                peak_measured=m_peak, frac_of_measured=(achieved / m_peak) if m_peak else None,
                traffic=traffic, traffic_source=traffic_source, traffic_over_algorithmic=(traffic / alg_bytes) if traffic else None,
                avg_launch_us=avg_ms * 1e3, launches=pk["count"], avg_m=avg_m,
                algorithmic_bytes_per_launch=alg_bytes, algorithmic_flops_per_launch=alg_flops, arithmetic_intensity=intensity,
                gemm_ms_share={k: v["ms"] / gemm_ms_total for k, v in prof.items()} if gemm_ms_total else {},
                all_gemms_tflops=(sum(2.0 * v["rows"] * target.gemm_shape(k)[0] * target.gemm_shape(k)[1] for k, v in prof.items())
                                  / (gemm_ms_total * 1e-3) / 1e12) if gemm_ms_total else 0.0)


def timed_pass(target, draft, dprompts, n_warm_batches, n_batches, streams, fn, args, dev, profile=True):
    """`n_warm_batches` untimed then `n_batches` timed lock-step batches of `streams` users from `dprompts`.  -> dict(outs, elapsed, prof, prof_big)"""
    ups = max(1, streams)
    n_warm, n_timed = n_warm_batches * ups, n_batches * ups
    assert n_warm + n_timed <= len(dprompts)
    run_users(target, draft, dprompts, 0, n_warm, streams, fn, args)
    if streams > 1 and n_warm == 0:                  # create every decoder / grow the batch buffers outside the timed region
        BSSD_batch(target, draft, dprompts[:streams], args.gamma, args.new_tokens, prefix_allowed_tokens_fn=fn)
    if profile:
        target.profile(1)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    outs = run_users(target, draft, dprompts, n_warm, n_warm + n_timed, streams, fn, args)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    prof_big = target.profile_big() if profile else None
    prof = target.profile(0) if profile else None
    return dict(outs=outs, elapsed=elapsed, prof=prof, prof_big=prof_big, n_timed=n_timed)


def pass_summary(r, args, streams):
    outs, n = r["outs"], r["n_timed"]
    runs = sum(o["n_run"] for o in outs)
    return dict(items_per_s=n * args.beam / r["elapsed"], ms_per_step=1e3 * r["elapsed"] / max(1, n // max(1, streams)), users=n,
                mean_accept_len=sum(o["total_accept_steps"] for o in outs) / max(1, runs),
                target_forwards_per_user=sum(o["n_target_forwards"] for o in outs) / n, n_run_per_user=runs / n)


def forwards_roofline(logs, elapsed_s, bytes_per_weight=2):
    """A whole decode call against its roofline.  `logs` = [(dims, [(tokens, logit rows), ...])] per model: every forward the call ran, as
    the engine logged it.  A forward can finish no sooner than max(its weights streamed once at 8 TB/s, its matrix flops at 2.5 PF): one
    user's forwards (20-230 tokens) sit on the first term, 256 users' on the second, and the band in between is where both matter."""
    hbm_s = mfma_s = ideal_s = 0.0
    n_fwd = tokens = 0
    n_hbm_bound = 0
    for dims, log in logs:
        w_layers = dims.n_layers * (4 * dims.hidden * dims.hidden + 3 * dims.hidden * dims.ffn)
        w_head = dims.hidden * dims.vocab_size
        for t_, r_ in log:
            b = (w_layers + w_head) * bytes_per_weight / (HBM_PEAK_GBS * 1e9)
            f = 2.0 * (w_layers * t_ + w_head * r_) / (MFMA_PEAK_TFLOPS * 1e12)
            hbm_s += b; mfma_s += f; ideal_s += max(b, f)
            n_hbm_bound += 1 if b >= f else 0
            n_fwd += 1; tokens += t_
    return dict(bound="hbm" if n_hbm_bound * 2 >= n_fwd else "mfma", ideal_ms=1e3 * ideal_s, measured_ms=1e3 * elapsed_s,
                frac=(ideal_s / elapsed_s) if elapsed_s > 0 else 0.0, forwards=n_fwd, forwards_hbm_bound=n_hbm_bound,
                tokens_per_forward=(tokens / n_fwd) if n_fwd else 0.0, weight_stream_ms=1e3 * hbm_s, mfma_ms=1e3 * mfma_s,
                what="sum over the call's forwards (target + draft) of max(weight bytes / 8 TB/s, matrix flops / 2.5 PF dense bf16) over the measured time")


def speedup_curve(target, draft, dprompts, first, sizes, fn, args, dev):
    """The reference's own figure of merit (inference.py:179: speedup = target_generate time / BSSD time, CSV columns :152-156) on THIS engine, per
    lock-step batch size: BSSD(_batch) and target_generate(_batch) on the same users, same weights, same box, each point with its roofline
    fraction (forwards_roofline) and the per-kind GEMM launch times of the BSSD call."""
    curve = []
    for s in sizes:
        grp = dprompts[first: first + s]
        if len(grp) < s:
            continue
        if s == 1:
            bssd = lambda: [BSSD(target, draft, grp[0], args.gamma, args.new_tokens, prefix_allowed_tokens_fn=fn)]
            tg = lambda: [target_generate(target, grp[0], args.new_tokens, prefix_allowed_tokens_fn=fn)]
        else:
            bssd = lambda: BSSD_batch(target, draft, grp, args.gamma, args.new_tokens, prefix_allowed_tokens_fn=fn)
            tg = lambda: target_generate_batch(target, grp, args.new_tokens, prefix_allowed_tokens_fn=fn)
        reps = 3 if s <= 16 else 2
        point = dict(users_per_batch=s)
        for name, call in (("bssd", bssd), ("target_generate", tg)):
            call(); call()                                            # decoders created, recurring shapes settled
            torch.cuda.synchronize(dev)
            tc = time.perf_counter()
            for _ in range(reps):
                outs = call()
            torch.cuda.synchronize(dev)
            sec = (time.perf_counter() - tc) / reps
            # one more, untimed, call with the forward log and the GEMM brackets on (the brackets add events to the stream)
            target.forward_log(1); draft.forward_log(1); target.profile(1)
            call()
            torch.cuda.synchronize(dev)
            prof = target.profile(0)
            tlog, dlog = target.forward_log(0), draft.forward_log(0)
            d = dict(ms_to_last_result=1e3 * sec, ms_per_user=1e3 * sec / s, items_per_s=s * args.beam / sec,
                     roofline=forwards_roofline([(target.dims, tlog), (draft.dims, dlog)], sec),
                     target_forwards=len(tlog), target_tokens_per_forward=[t_ for t_, _ in tlog], draft_forwards=len(dlog),
                     per_kind_us={k: 1e3 * v["ms"] / max(1, v["count"]) for k, v in prof.items()})
            if name == "bssd":
                d["mean_accept_len"] = sum(o["total_accept_steps"] for o in outs) / max(1, sum(o["n_run"] for o in outs))
            point[name] = d
        point["speedup"] = point["target_generate"]["ms_to_last_result"] / point["bssd"]["ms_to_last_result"]   # inference.py:179
        curve.append(point)
    return curve


def main():
    args = parse()
    mode = resolve_world(args)
    if mode[0] == "spawn":
        # one process per GPU, started BEFORE anything in this process initialises the GPU (torch.cuda.device_count() does not, on this
        # image); the parent only relays the children's output and exit code -- it never execs over itself
        import subprocess
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} requested but only {n_dev} HIP device(s) are visible")
        raise SystemExit(subprocess.run(launcher_command(args.gpus, sys.argv[1:])).returncode)
    _, rank, local_rank, world = mode
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    # rehearsal switches (NOT for measurements): ATSPEED_BENCH_BACKEND=gloo + ATSPEED_BENCH_SHARE_GPU=1 let two ranks run the N > 1 code
    # path on a ONE-GPU box (both on cuda:0, counters gathered over gloo) -- RCCL itself refuses two ranks on one device
    backend = os.environ.get("ATSPEED_BENCH_BACKEND", "nccl")
    share_gpu = os.environ.get("ATSPEED_BENCH_SHARE_GPU", "0") == "1"
    dev_index = 0 if (world == 1 or share_gpu) else local_rank
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        dist.init_process_group(backend)      # "nccl" = RCCL over xGMI
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    vocab = synth.BEAUTY if args.dataset == "beauty" else synth.GAMES
    V = vocab.vocab_size
    tdims = synth.llama_7b(V, args.target_layers)
    ddims = synth.llama_68m(V)
    kw = dict(max_slots=512, max_tokens=512, max_logit_rows=384, device=dev)

    def build_pair(tdims_, ddims_, dtype=torch.bfloat16, resid_scale=None, **extra):
        rs = 1.0 if resid_scale is None else resid_scale
        k2 = dict(kw, **extra)
        d = HipLlama.from_synthetic(ddims_, args.seed + 1, std=0.02, head_std=0.02, dtype=dtype, num_beams=args.draft_beam, resid_scale=rs, **k2)
        t = HipLlama.from_synthetic(tdims_, args.seed, std=0.02, head_std=0.02, dtype=dtype, num_beams=args.beam, resid_scale=rs,
                                    align_to=(d if resid_scale is not None else None), **k2)
        if args.do_sample:
            for m in (t, d):
                m.generation_config.do_sample = True
                m.generation_config.temperature = args.temperature
        return t, d

    target, draft = build_pair(tdims, ddims)
    if args.target_fp8:
        target.enable_fp8()
    if args.do_sample:
        torch.manual_seed(args.seed)
    fn = make_mask(vocab, args.mask)

    # a STEP is one lock-step batch: `--streams` users decoded together (one pass of the hot path over one batch of inputs)
    ups = max(1, args.streams)
    n_warm, n_timed = args.warmup * ups, args.steps * ups
    n_local = n_warm + n_timed
    prompts, dprompts = make_prompts(n_local, rank * n_local, args.seed, args.dataset, dev)   # contiguous user shard per rank

    run_users(target, draft, dprompts, 0, n_warm, args.streams, fn, args)
    if args.streams > 1:                         # create every decoder / grow the batch buffers outside the timed region
        BSSD_batch(target, draft, dprompts[:args.streams], args.gamma, args.new_tokens, prefix_allowed_tokens_fn=fn)
    target.profile(1)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    n_run = acc = 0
    stage = np.zeros(3)
    n_tf = n_df = 0
    outs = []
    for o in run_users(target, draft, dprompts, n_warm, n_local, args.streams, fn, args):
        n_run += o["n_run"]; acc += o["total_accept_steps"]
        stage += (o["draft_time_cost"], o["target_time_cost"], o["verify_time_cost"])
        n_tf += o["n_target_forwards"]; n_df += o["n_draft_forwards"]
        outs.append(o)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof_big = target.profile_big()
    prof = target.profile(0)
    solo = world == 1                 # auxiliary passes belong to the N = 1 line only
    sub_steps = max(1, min(args.sub_steps, args.steps))
    sub_warm = 1 if (args.warmup + args.steps) > sub_steps else 0       # sub-passes reuse the main pass's users: one warm batch if there is room

    # ---- measured peaks of THIS box: register-only bf16 MFMA loop on random operands, read-only HBM stream over 2 GiB
    measured = None
    if rank == 0:
        import ctypes as C
        from atspeed_amd import _lib
        lib, tf, gbs = _lib.load(), C.c_double(), C.c_double()
        buf = torch.empty(2 << 30, dtype=torch.uint8, device=dev).fill_(1)
        scratch = torch.empty(4 << 20, dtype=torch.uint8, device=dev)
        _lib.check(lib.atspeed_probe_mfma_bf16(4000, scratch.data_ptr(), scratch.numel(), _lib.stream_ptr(dev), C.byref(tf)))
        _lib.check(lib.atspeed_probe_hbm_read(buf.data_ptr(), buf.numel(), 8, scratch.data_ptr(), _lib.stream_ptr(dev), C.byref(gbs)))
        measured = dict(mfma_bf16_tflops=tf.value, hbm_read_gbs=gbs.value,
                        how="atspeed_probe_mfma_bf16 (16x16x32, random operands, 8 waves/CU) and atspeed_probe_hbm_read (2 GiB, 8 passes, best of 2 access shapes x 5 grids)")
        del buf, scratch

    sub_errors = {}

    device_dead = []

    def guarded(name, fn_):
        """An auxiliary pass must never cost the headline: an ORDINARY failure is recorded on the line (`sub_pass_errors`) and the run goes on.
        A DEVICE error (the library's ERR_HIP status, a HIP error out of torch, a synchronize that fails) poisons the context: every later GPU
        pass is skipped, the line is still printed with what was measured before, and the process exits non-zero -- no in-process retry."""
        if device_dead:
            sub_errors[name] = f"skipped: device error in {device_dead[0]}"
            return None
        try:
            return fn_()
        except Exception as e:                                        # noqa: BLE001
            sub_errors[name] = f"{type(e).__name__}: {e}"[:400]
            from atspeed_amd import _lib as _l
            hip = (isinstance(e, _l.AtSpeedError) and e.status == _l.ERR_HIP) or "HIP error" in str(e) or "hipError" in str(e)
            try:
                torch.cuda.synchronize(dev)
            except Exception:                                         # noqa: BLE001
                hip = True
            if hip:
                device_dead.append(name)
            return None

    # ---- speedup curve (the reference's figure of merit, inference.py:179): BSSD against plain beam search on the same engine, users and
    # box, by users per lock-step batch, each point with its roofline fraction; `latency_curve` is its BSSD leg (kept under that name)
    sizes = [s_ for s_ in (1, 4, 16, 64, 256) if s_ <= args.streams and s_ <= n_timed]

    def speedup_pass(t_, d_, first):
        return speedup_curve(t_, d_, dprompts, first, sizes, fn, args, dev)

    speedup = guarded("speedup_curve", lambda: speedup_pass(target, draft, n_warm)) if (solo and not args.no_latency_curve and args.streams > 1) else None
    curve = [dict(users_per_batch=p_["users_per_batch"], **{k_: p_["bssd"][k_] for k_ in ("ms_to_last_result", "items_per_s", "ms_per_user")})
             for p_ in speedup] if speedup else None
    if speedup:
        release_decoders(target)                      # the plain-beam-search decoders of the curve (a KV arena per lane)

    # ---- the reference's own regime, for the record (not part of `value`): a few users strictly one at a time.
    # Here every projection is one pass over the weights (M ~ 20-230 tokens): HBM-bound.
    def single_user_pass():
        n1 = min(args.single_stream_users, n_timed)
        for u in range(n_warm, n_warm + min(3, n1)):       # warm-up
            BSSD(target, draft, dprompts[u], args.gamma, args.new_tokens, prefix_allowed_tokens_fn=fn)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for u in range(n_warm, n_warm + n1):
            BSSD(target, draft, dprompts[u], args.gamma, args.new_tokens, prefix_allowed_tokens_fn=fn)
        torch.cuda.synchronize(dev)
        dt1 = time.perf_counter() - t1
        target.profile(1)                                           # GEMM brackets from a second, untimed pass
        for u in range(n_warm, n_warm + min(3, n1)):
            BSSD(target, draft, dprompts[u], args.gamma, args.new_tokens, prefix_allowed_tokens_fn=fn)
        torch.cuda.synchronize(dev)
        p1 = target.profile(0)
        k1 = max(p1, key=lambda k: p1[k]["ms"])
        N1, K1 = target.gemm_shape(k1)
        m1 = p1[k1]["rows"] / max(1, p1[k1]["count"])
        b1 = N1 * K1 * 2 + m1 * K1 * 2 + m1 * (N1 // 2 if k1 == "gate_up" else N1) * (4 if k1 == "lm_head" else 2)
        us1 = 1e3 * p1[k1]["ms"] / max(1, p1[k1]["count"])
        # the whole forward against the stream of its weights: target forwards per user x W_t bytes / time
        tf_user = sum(o["n_target_forwards"] for o in outs[:n1]) / n1
        single = dict(users=n1, items_per_s=n1 * args.beam / dt1, ms_per_user=1e3 * dt1 / n1,
                      weights_stream_gbs=tf_user * tdims.n_params_streamed() * 2 / (dt1 / n1) / 1e9,
                      weights_stream_frac_of_hbm_peak=tf_user * tdims.n_params_streamed() * 2 / (dt1 / n1) / 1e9 / HBM_PEAK_GBS,
                      roofline=dict(bound="hbm", kernel=f"small-M projection [{k1}] N={N1} K={K1} avg_M={m1:.0f}",
                                    achieved=b1 / (us1 * 1e-6) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                    frac=b1 / (us1 * 1e-6) / 1e9 / HBM_PEAK_GBS, avg_launch_us=us1,
                                    per_kind_us={k: 1e3 * v["ms"] / max(1, v["count"]) for k, v in p1.items()}))
        return single

    single = guarded("single_user_pass", single_user_pass) if (solo and args.streams > 1 and args.single_stream_users > 0) else None

    # ---- high-acceptance brackets: identical shapes / kernels / users, weights aligned so the draft's beams are mostly the target's
    # (the natural bracket above accepts ~0 steps because the random weights are unrelated).  Next to each: the fp32 ENGINE (bit-exact
    # against the CPU oracle at these dims, tests/test_fulldims_gpu.py) on the same weight values and users -- the "reference's" accepted length.
    scales = [float(x) for x in args.aligned_resid_scale.split(",") if x.strip()]

    def aligned_pass():
        fp8_drift = None
        release_decoders(target, draft)                # the main pass's per-user KV arenas: room for the other model pairs
        aligned = []
        for rs in scales:
            target_a, draft_a = build_pair(tdims, ddims, resid_scale=rs)
            if args.target_fp8:
                target_a.enable_fp8()
            r = timed_pass(target_a, draft_a, dprompts, sub_warm, sub_steps, args.streams, fn, args, dev, profile=False)
            br = dict(resid_scale=rs, **pass_summary(r, args, ups))
            if not args.no_latency_curve and not args.do_sample:
                br["speedup_curve"] = speedup_pass(target_a, draft_a, sub_warm * ups)
            nf = min(args.aligned_fp32_users, r["n_timed"])
            if nf > 0 and not args.do_sample:
                t32, d32 = build_pair(tdims, ddims, dtype=torch.float32, resid_scale=rs, round_to_bf16=True, max_logit_rows=448)
                tf0 = time.perf_counter()
                fo = [BSSD(t32, d32, dprompts[sub_warm * ups + u], args.gamma, args.new_tokens, prefix_allowed_tokens_fn=fn) for u in range(nf)]
                torch.cuda.synchronize(dev)
                go = r["outs"][:nf]
                br["fp32_engine"] = dict(
                    users=nf, mean_accept_len=sum(o["total_accept_steps"] for o in fo) / max(1, sum(o["n_run"] for o in fo)),
                    bf16_mean_accept_len_same_users=sum(o["total_accept_steps"] for o in go) / max(1, sum(o["n_run"] for o in go)),
                    users_with_equal_accept_steps=sum(1 for a, b in zip(fo, go) if a["accept_steps"] == b["accept_steps"]),
                    users_with_equal_item_sets=sum(1 for a, b, p in zip(fo, go, prompts[sub_warm * ups:]) if
                                                   {tuple(x) for x in a["beam_sequence"][:, len(p):].tolist()} == {tuple(x) for x in b["beam_sequence"][:, len(p):].tolist()}),
                    seconds=time.perf_counter() - tf0,
                    what="the fp32 engine (exact-fp32 MFMA, one user at a time) on the same bf16-valued weights and users; it equals the CPU oracle bit for bit at these dims")
                release_decoders(t32, d32)
                del t32, d32
            if not args.target_fp8 and not args.do_sample and not args.no_configs and args.one_user_fp8_users > 0:
                # config 5 in the reference's own regime (one user per call, inference.py:86-91,162-176): the same users one at a time, bf16
                # target then W8A8 target (weight-streaming kernels), with the accepted-length drift between the two
                n1 = min(args.one_user_fp8_users, r["n_timed"])
                b1, _ = one_user_loop(target_a, draft_a, dprompts, sub_warm * ups, n1, fn, args, dev, 2)
                target_a.enable_fp8()
                target_a.fp8_counters(reset=True)
                f1, _ = one_user_loop(target_a, draft_a, dprompts, sub_warm * ups, n1, fn, args, dev, 1)
                c1 = target_a.fp8_counters()
                br["one_user"] = dict(users=n1, bf16_ms_per_user=b1["ms_per_user"], fp8_ms_per_user=f1["ms_per_user"], bf16_mean_accept_len=b1["mean_accept_len"],
                                      fp8_mean_accept_len=f1["mean_accept_len"], drift_steps=f1["mean_accept_len"] - b1["mean_accept_len"],
                                      fp8_weights_stream_frac_of_hbm_peak=f1["weights_stream_frac_of_hbm_peak"],
                                      bf16_weights_stream_frac_of_hbm_peak=b1["weights_stream_frac_of_hbm_peak"],
                                      projections_not_in_fp8=sum(v["other"] for v in c1.values()))
            if rs == max(scales) and not args.target_fp8 and not args.do_sample and not args.no_configs:
                # config 5's accepted-length drift: the SAME aligned pair and users with the target's projections in fp8
                target_a.enable_fp8()
                r8 = timed_pass(target_a, draft_a, dprompts, sub_warm, sub_steps, args.streams, fn, args, dev, profile=False)
                s8 = pass_summary(r8, args, ups)
                fp8_drift = dict(resid_scale=rs, users=r8["n_timed"], bf16_mean_accept_len=br["mean_accept_len"], fp8_mean_accept_len=s8["mean_accept_len"],
                                 drift_steps=s8["mean_accept_len"] - br["mean_accept_len"], fp8_items_per_s=s8["items_per_s"])
            aligned.append(br)
            release_decoders(target_a, draft_a)
            del target_a, draft_a
        return aligned, fp8_drift

    aligned, fp8_drift = (guarded("aligned_weight_brackets", aligned_pass) or (None, None)) if (solo and scales) else (None, None)

    # ---- BASELINE configs 3 and 5 as bounded sub-passes of the same line (driver-clocked)
    def configs_pass():
        configs = {}
        if args.dataset == "beauty" and not args.target_fp8:
            gv = synth.GAMES
            tg_, dg_ = build_pair(synth.llama_7b(gv.vocab_size, args.target_layers), synth.llama_68m(gv.vocab_size))
            _, gprompts = make_prompts((1 + sub_steps) * ups, 0, args.seed, "games", dev)
            for name, mk in (("games_256", "position"), ("games_256_trie", "trie")):
                gfn = make_mask(gv, mk)
                r = timed_pass(tg_, dg_, gprompts, 1, sub_steps, args.streams, gfn, args, dev)
                configs[name] = dict(workload=f"Games V={gv.vocab_size}, Llama-68M / Llama-7B({args.target_layers}L), K={args.beam}, DK={args.draft_beam}, {args.streams} users per "
                                              f"lock-step batch, {'position-set mask' if mk == 'position' else 'strict item trie (17 332 items)'}, bf16",
                                     steps=sub_steps, **pass_summary(r, args, ups),
                                     roofline=gemm_roofline(tg_, r["prof"], r["prof_big"], False, measured, args.streams, False))
            release_decoders(tg_, dg_)
            del tg_, dg_
        if not args.target_fp8:
            # the reference's own dtype (inference.py:75-100 loads both models in fp16): the engine's fp16 flavour on the headline workload
            release_decoders(target, draft)
            t16, d16 = build_pair(tdims, ddims, dtype=torch.float16)
            r = timed_pass(t16, d16, dprompts, sub_warm, sub_steps, args.streams, fn, args, dev)
            configs["fp16"] = dict(workload=f"{args.dataset.capitalize()} V={V}, Llama-68M / Llama-7B({args.target_layers}L) in fp16 (ATSPEED_F16: the bf16 engine's kernels "
                                            f"compiled for IEEE half, v_mfma_f32_16x16x32_f16), K={args.beam}, {args.streams} users per lock-step batch",
                                   steps=sub_steps, dtype="fp16", **pass_summary(r, args, ups),
                                   roofline=gemm_roofline(t16, r["prof"], r["prof_big"], False, measured, args.streams, False))
            # ... and the reference's actual COMBINATION (inference.py:75-91: fp16 checkpoints, target loaded 8-bit): W8A8 projections on the fp16 flavour
            release_decoders(t16, d16)
            t16.enable_fp8()
            t16.fp8_counters(reset=True)
            r = timed_pass(t16, d16, dprompts, sub_warm, sub_steps, args.streams, fn, args, dev)
            cnt16 = t16.fp8_counters()
            one16 = None
            if args.one_user_fp8_users > 0:
                one16, _ = one_user_loop(t16, d16, dprompts, n_warm, min(args.one_user_fp8_users, n_timed), fn, args, dev, 1)
            configs["fp8_fp16"] = dict(workload=f"{args.dataset.capitalize()} V={V}, fp16 draft / fp16 Llama-7B({args.target_layers}L) target with e4m3 W8A8 projections (the reference's "
                                                f"dtype combination, inference.py:75-91), K={args.beam}, {args.streams} users per lock-step batch",
                                       steps=sub_steps, dtype="fp8-e4m3 (W8A8 target projections, fp16 elsewhere)", **pass_summary(r, args, ups),
                                       projection_launches={k: v for k, v in cnt16.items()},
                                       roofline=gemm_roofline(t16, r["prof"], r["prof_big"], True, measured, args.streams, False),
                                       one_user=one16,
                                       parity="pinned to the build's W8A8 oracle on the fp16 weight values, tests/test_fp8_gpu.py[*fp16]")
            release_decoders(t16, d16)
            del t16, d16
        if not args.target_fp8:
            release_decoders(target, draft)
            target.enable_fp8()                        # from here on the headline target runs its batched projections in fp8
            target.fp8_counters(reset=True)
            r = timed_pass(target, draft, dprompts, sub_warm, sub_steps, args.streams, fn, args, dev)
            cnt = target.fp8_counters()
            one_user_fp8 = None
            if args.one_user_fp8_users > 0:                 # the headline weights (zero-acceptance bracket: four target forwards per user), one user per call
                target.fp8_counters(reset=True)
                one_user_fp8, _ = one_user_loop(target, draft, dprompts, n_warm, min(args.one_user_fp8_users, n_timed), fn, args, dev, 1)
                c8 = target.fp8_counters()
                target.profile(1)                           # GEMM brackets from a second, untimed pass (512 event pairs per user cost ~20 % of the loop)
                one_user_loop(target, draft, dprompts, n_warm, min(3, n_timed), fn, args, dev, 1)
                p8 = target.profile(0)
                one_user_fp8.update(projections_not_in_fp8=sum(v["other"] for v in c8.values()),
                                    per_kind_us={k: 1e3 * v["ms"] / max(1, v["count"]) for k, v in p8.items()},
                                    bf16_ms_per_user=(single or {}).get("ms_per_user"),
                                    roofline="target forwards x (6.48 G e4m3 layer weights + the bf16 lm_head) / time vs 8 TB/s")
            configs["fp8"] = dict(workload=f"{args.dataset.capitalize()} V={V}, Llama-7B({args.target_layers}L) target verify in fp8 (e4m3 W8A8 projections on the block-scaled MFMA, "
                                           f"bf16 elsewhere), K={args.beam}, {args.streams} users per lock-step batch",
                                  steps=sub_steps, dtype="fp8-e4m3 (W8A8 target projections, bf16 elsewhere)", **pass_summary(r, args, ups),
                                  projection_launches={k: v for k, v in cnt.items()},
                                  roofline=gemm_roofline(target, r["prof"], r["prof_big"], True, measured, args.streams, False),
                                  accepted_length_drift_vs_bf16=fp8_drift,
                                  one_user=one_user_fp8,
                                  parity="unpinned against the reference (its 8-bit target is bitsandbytes LLM.int8, absent offline); pinned to the build's W8A8 oracle, tests/test_fp8_gpu.py")
        return configs

    configs = guarded("configs", configs_pass) if (solo and not args.no_configs and not args.do_sample and args.streams > 1) else None

    # ---- the verify step's scan (full-vocabulary log-sum-exp over the packed logit rows of one lock-step round, the HBM-bound
    # kernel of beamSD.py:285): rows = users x (1 + 3*DK) at V fp32 logits, timed alone with events on its launch stream
    def scan_pass():
        from atspeed_amd import _lib
        lib = _lib.load()
        rows = max(1, args.streams) * (1 + (args.new_tokens - 1) * args.draft_beam)
        ld = target.logits_ld
        lg = torch.randn(rows, ld, dtype=torch.float32, device=dev)
        lse = torch.empty(rows, dtype=torch.float32, device=dev)
        st = _lib.stream_ptr(dev)
        run_lse = lambda: _lib.check(lib.atspeed_lse_rows(lg.data_ptr(), rows, V, ld, lse.data_ptr(), st))
        for _ in range(3):
            run_lse()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run_lse()
        e1.record()
        torch.cuda.synchronize(dev)
        us = e0.elapsed_time(e1) * 1e3 / 20
        scan = dict(kernel="lse_rows_kernel", rows=rows, bytes_per_launch=rows * V * 4, avg_launch_us=us, bound="hbm",
                    achieved=rows * V * 4 / (us * 1e-6) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=rows * V * 4 / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    where="standalone launch of atspeed_lse_rows on a synthetic buffer of one verification round's shape: the kernel the one-user path "
                          "and the C-ABI keep.  The lock-step decode loop no longer runs it (see in_situ)")
        del lg, lse
        # in situ: the batched forwards take the normaliser out of the lm_head GEMM's epilogue and write only the logit tiles that
        # hold a token of the constraint automaton, so the verify step's former HBM-bound pass (write + re-read of every logit) is gone
        lm = prof["lm_head"]
        tiles_all = (V + 255) // 256
        tiles_kept = len({int(t) // 256 for t in fn.compile(prompts[0].tolist()).tok})
        avg_rows = lm["rows"] / max(1, lm["count"])
        scan["in_situ"] = dict(kernel="gemm_ring_kernel<4 (EPI_F32_LSE), 8, false> + lse_combine_kernel (lm_head with the fused full-vocabulary normaliser)",
                               launches=lm["count"], avg_rows=avg_rows, avg_launch_us=1e3 * lm["ms"] / max(1, lm["count"]),
                               logit_tiles_written=tiles_kept, logit_tiles_total=tiles_all,
                               bytes_written_per_launch=avg_rows * (tiles_kept * 256 * 4 + tiles_all * 8),
                               bytes_avoided_per_launch=avg_rows * ((tiles_all - tiles_kept) * 256 * 4 + V * 4),
                               note="avoided = logit tiles never written + the LSE pass's re-read of every logit (what lse_rows_kernel streamed)")
        if measured:
            scan["peak_measured"], scan["frac_of_measured"] = measured["hbm_read_gbs"], scan["achieved"] / measured["hbm_read_gbs"]
        return scan

    scan = guarded("verify_scan", scan_pass) if rank == 0 else None

    per_rank = all_gather_counters(Counters(n_timed, n_run, acc, int(elapsed * 1e9)), dev if backend == "nccl" else "cpu")   # the path's single collective
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    agg = aggregate(per_rank, args.beam)
    users, t_max, value, mean_accept = agg["users"], agg["elapsed_s"], agg["items_per_s"], agg["mean_accept_len"]
    total_runs = sum(c.n_run for c in per_rank)

    roofline = gemm_roofline(target, prof, prof_big, args.target_fp8, measured, args.streams, True)
    roofline["target_forward"] = dict(avg_ms_per_user=1e3 * stage[1] / max(1, n_tf), weight_bytes=tdims.n_params_streamed() * 2)

    line = {
        "metric": "recommended items/sec (K=20 beams per user), mean accepted length alongside",
        "value": value, "unit": "items/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp8-e4m3 (W8A8 target projections, bf16 elsewhere)" if args.target_fp8 else "bf16", "data": "synthetic (hash-PRNG weights, Beauty-shaped vocabulary and prompts)",
        "config": {"workload": f"{args.dataset.capitalize()} V={V}, Llama-68M draft / Llama-7B({args.target_layers}L) target, K={args.beam}, DK={args.draft_beam}, "
                               f"gamma={args.gamma}, L={args.new_tokens}, {args.streams} user(s) per lock-step batch per GPU, "
                               f"{'position-set mask' if args.mask == 'position' else 'strict item trie'}",
                   "users_per_step": ups, "users_per_gpu": n_timed, "streams": args.streams,
                   "mean_prompt_len": float(np.mean([len(p) for p in prompts[n_warm:]])),
                   "parallelism": f"user-shard x{world}" + ("" if backend == "nccl" and not share_gpu else f" (REHEARSAL: backend {backend}, ranks share one GPU: {share_gpu})"),
                   "operand_layout": "packed (row pairs per 128-byte line)" if getattr(target, "weights_packed", False) else "row-major",
                   "kernel_sha": kernel_sha()},
        "mean_accept_len": mean_accept,
        "accept_note": "unrelated random draft/target weights accept ~0 draft steps: worst-case bracket (3 verify rounds + 1 final step = 4 target forwards per user)",
        "per_user": {"n_run": total_runs / users, "target_forwards": n_tf / n_timed, "draft_forwards": n_df / n_timed,
                     "draft_ms": 1e3 * stage[0] / n_timed, "target_ms": 1e3 * stage[1] / n_timed, "verify_ms": 1e3 * stage[2] / n_timed},
        "per_rank": per_rank_report(per_rank, args.beam),
        "roofline": roofline,
        "measured_peaks": measured,
        "verify_scan": scan,
        "configs": configs,                   # BASELINE configs 3 (Games, 256 users) and 5 (fp8 verify), bounded sub-passes
        "speedup_curve": speedup,             # BSSD vs target_generate on the same engine (inference.py:179), by users per batch, with roofline fractions
        "latency_curve": curve,
        "single_user_stream": single,
        "aligned_weight_brackets": aligned,   # same users, shapes and kernels as `value`; only the weights' agreement differs
        "sub_pass_errors": sub_errors or None,   # an auxiliary pass that raised (the headline above does not depend on any of them)
    }
    if args.do_sample:
        line["decoding"] = f"sampling (temperature {args.temperature})"
    def cpu_pass():
        # (after the fp8 sub-pass the headline target also carries fp8 copies; its bf16 weights, which export_state_dict reads, are untouched)
        cb, ref_outs, (ref_t, _ref_d) = cpu_baseline(target, draft, prompts[n_warm:], fn, args)
        # next to the timing: the bf16 engine's items vs the fp32 oracle's on identical weights, every rank at which they disagree scored
        # by the oracle itself (the asserted form over 64 users is tests/test_decisions_gpu.py)
        P0 = len(prompts[n_warm])
        rep = disagreement_report(outs[0], ref_outs[0], P0, ref_t, prompts[n_warm])
        cb["top_k_overlap_with_gpu_bf16"] = rep["top_k_overlap"]
        cb["bf16_vs_fp32_disagreements"] = rep
        cb["gpu_accept_len_same_users"] = float(sum(o["total_accept_steps"] for o in outs[:len(ref_outs)])) / max(
            1, sum(o["n_run"] for o in outs[:len(ref_outs)]))
        return cb

    line["cpu_baseline"] = guarded("cpu_baseline", cpu_pass) if (world == 1 and not args.no_cpu_baseline and not args.do_sample) else None
    line["sub_pass_errors"] = sub_errors or None
    emit(line)
    if device_dead:
        raise SystemExit(f"bench.py: device error in sub-pass {device_dead[0]} (line printed above with sub_pass_errors); exiting non-zero")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
