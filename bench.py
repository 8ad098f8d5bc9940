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

Extra objects on the JSON line:
  roofline     — the dominant kernel (the target forward's gate_up projection GEMM), hipEvent-bracketed on its
                 launch stream: algorithmic flops / launch time vs the 2.5 PF dense bf16 MFMA peak when users are
                 batched (HBM bytes vs 8 TB/s with --streams 1); verify_scan: the verify step's scan vs 8 TB/s.
  cpu_baseline — the oracle (oracle/beamsd_ref.py, torch-CPU fp32) timed on the host cores on a
                 bounded sample of the same workload with the same weights (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
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
from atspeed_amd.beamSD import BSSD, BSSD_batch, release_decoders    # noqa: E402
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
    ap.add_argument("--target-fp8", action="store_true", help="BASELINE config 5: fp8 (e4m3 W8A8) target projections in the batched forwards")
    ap.add_argument("--single-stream-users", type=int, default=6, help="extra untimed-for-value pass: users decoded one at a time (the reference's loop)")
    ap.add_argument("--aligned-resid-scale", type=str, default="3e-6,3e-5",
                    help="extra brackets (SURVEY.md 8d 'oracle-draft'): same shapes and kernels, draft/target weights aligned through a "
                         "shared bigram table with the layers' residual contributions scaled by each factor of this comma list "
                         "(3e-6: every draft step accepted, 3e-5: about one step per verification); empty string skips the pass")
    ap.add_argument("--dataset", choices=("beauty", "games"), default="beauty", help="vocabulary / prompt-length shape (games = BASELINE config 3)")
    ap.add_argument("--mask", choices=("position", "trie"), default="position",
                    help="position = the per-position allowed sets inference.py installs; trie = strict item trie on the generated suffix")
    ap.add_argument("--do-sample", action="store_true", help="sampling-mode beam-SD (generation_config.do_sample) instead of the greedy headline")
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--cpu-baseline-users", type=int, default=1)
    ap.add_argument("--aligned-oracle-users", type=int, default=1, help="users per aligned-weight bracket that the CPU oracle also decodes (accepted length next to the GPU's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


TRAFFIC_FILE = os.path.join("profiles", "pmc_traffic.json")


def traffic_from_profiles(kind: str):
    """(HBM bytes per launch, provenance) from the committed PMC summary, if present.  The counters need their own rocprofv3 --pmc
    passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), so this number is NOT measured by the run that prints it: the line
    carries `traffic_source` next to it."""
    p = os.path.join(ROOT, TRAFFIC_FILE)
    if not os.path.exists(p):
        return None, None
    try:
        with open(p) as f:
            d = json.load(f)
        v = d.get(kind, {}).get("hbm_bytes_per_launch")
        src = f"{TRAFFIC_FILE} (offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this workload; not measured in this run)"
        return v, (src if v is not None else None)
    except Exception:
        return None, None


def cpu_baseline(target, draft, prompts, fn, args, n_users=None):
    """Oracle (CPU restatement of the reference) on the same weights/prompts; bounded sample.  Returns (object, oracle outputs, oracle models)."""
    from oracle import beamsd_ref as R
    from oracle.llama_ref import RefLlama
    cores = torch.get_num_threads()
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
    """Where the bf16 engine's ranking of a user differs from the fp32 oracle's: per rank, the oracle's own score of both items.
    If the gaps are within the bf16 noise (|gpu score - oracle score of the same item|) the disagreements are near-ties of nearly
    flat random-init logits, not errors -- this prints the evidence instead of asserting it in a comment."""
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
               note="gap = how much worse (by the fp32 oracle's own arithmetic) the item the bf16 engine put at this rank is than the oracle's item there")
    return rep


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
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")       # RCCL over xGMI
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    vocab = synth.BEAUTY if args.dataset == "beauty" else synth.GAMES
    V = vocab.vocab_size
    tdims = synth.llama_7b(V, args.target_layers)
    ddims = synth.llama_68m(V)
    kw = dict(max_slots=512, max_tokens=512, max_logit_rows=384, device=dev)
    target = HipLlama.from_synthetic(tdims, args.seed, std=0.02, head_std=0.02, dtype=torch.bfloat16, num_beams=args.beam, **kw)
    draft = HipLlama.from_synthetic(ddims, args.seed + 1, std=0.02, head_std=0.02, dtype=torch.bfloat16, num_beams=args.draft_beam, **kw)
    if args.target_fp8:
        target.enable_fp8()
    if args.do_sample:
        for m in (target, draft):
            m.generation_config.do_sample = True
            m.generation_config.temperature = args.temperature
        torch.manual_seed(args.seed)
    if args.mask == "trie":
        from atspeed_amd.generation_trie import SuffixTrieConstraint, Trie
        fn = SuffixTrieConstraint(Trie([[1] + [int(t) for t in it] + [2] for it in synth.synthetic_items(vocab)]), synth.RESPONSE_SEP, 1)
    else:
        fn = PositionSetConstraint(vocab.allowed_tokens(), synth.RESPONSE_SEP)

    # a STEP is one lock-step batch: `--streams` users decoded together (one pass of the hot path over one batch of inputs)
    ups = max(1, args.streams)
    n_warm, n_timed = args.warmup * ups, args.steps * ups
    n_local = n_warm + n_timed
    first = rank * n_local                                   # contiguous user shard per rank
    plens = synth.prompt_lengths(world * n_local, args.seed, mean_hist=7.33 if args.dataset == "beauty" else 5.98)
    prompts = [synth.synthetic_prompt(int(plens[first + u]), synth.tensor_seed(args.seed, f"user{first + u}")) for u in range(n_local)]
    dprompts = [{"input_ids": torch.from_numpy(p)[None].to(dev)} for p in prompts]   # resident in HBM before timing

    def run_users(lo, hi):
        """users [lo, hi): groups of --streams interleaved lanes (1 = plain per-user BSSD calls)"""
        res = []
        if args.streams <= 1:
            for u in range(lo, hi):
                res.append(BSSD(target, draft, dprompts[u], args.gamma, args.new_tokens, prefix_allowed_tokens_fn=fn))
            return res
        for g in range(lo, hi, args.streams):
            res += BSSD_batch(target, draft, dprompts[g:min(hi, g + args.streams)], args.gamma, args.new_tokens,
                              prefix_allowed_tokens_fn=fn)
        return res

    run_users(0, n_warm)
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
    for o in run_users(n_warm, n_local):
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

    # ---- the reference's own regime, for the record (not part of `value`): a few users strictly one at a time.
    # Here every projection is one pass over the weights (M ~ 20-230 tokens): HBM-bound.
    single = None
    if world == 1 and args.streams > 1 and args.single_stream_users > 0:      # auxiliary passes belong to the N=1 line only
        n1 = min(args.single_stream_users, n_timed)
        for u in range(n_warm, n_warm + min(3, n1)):       # warm-up: recurring forward shapes get their hipGraphs
            BSSD(target, draft, dprompts[u], args.gamma, args.new_tokens, prefix_allowed_tokens_fn=fn)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for u in range(n_warm, n_warm + n1):
            BSSD(target, draft, dprompts[u], args.gamma, args.new_tokens, prefix_allowed_tokens_fn=fn)
        torch.cuda.synchronize(dev)
        dt1 = time.perf_counter() - t1
        target.profile(1)                                           # GEMM brackets from a second, untimed pass (profiling bypasses the graphs)
        for u in range(n_warm, n_warm + min(3, n1)):
            BSSD(target, draft, dprompts[u], args.gamma, args.new_tokens, prefix_allowed_tokens_fn=fn)
        torch.cuda.synchronize(dev)
        p1 = target.profile(0)
        k1 = max(p1, key=lambda k: p1[k]["ms"])
        N1, K1 = target.gemm_shape(k1)
        m1 = p1[k1]["rows"] / max(1, p1[k1]["count"])
        b1 = N1 * K1 * 2 + m1 * K1 * 2 + m1 * (N1 // 2 if k1 == "gate_up" else N1) * (4 if k1 == "lm_head" else 2)
        us1 = 1e3 * p1[k1]["ms"] / max(1, p1[k1]["count"])
        single = dict(users=n1, items_per_s=n1 * args.beam / dt1, ms_per_user=1e3 * dt1 / n1,
                      roofline=dict(bound="hbm", kernel=f"small-M projection [{k1}] N={N1} K={K1} avg_M={m1:.0f} (split-K ring GEMM + slab reduce)",
                                    achieved=b1 / (us1 * 1e-6) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                    frac=b1 / (us1 * 1e-6) / 1e9 / HBM_PEAK_GBS, avg_launch_us=us1))

    # ---- high-acceptance bracket: identical shapes / kernels / users, weights aligned so the draft's beams are mostly
    # the target's (the natural bracket above accepts ~0 steps because the random weights are unrelated).
    aligned = None
    scales = [float(x) for x in args.aligned_resid_scale.split(",") if x.strip()]
    if world == 1 and scales:
        release_decoders(target, draft)                # the main pass's per-user KV arenas: room for the second model pair
        aligned = []
        grp = max(1, args.streams)
        for rs in scales:
            draft_a = HipLlama.from_synthetic(ddims, args.seed + 1, std=0.02, head_std=0.02, dtype=torch.bfloat16, num_beams=args.draft_beam,
                                              resid_scale=rs, **kw)
            target_a = HipLlama.from_synthetic(tdims, args.seed, std=0.02, head_std=0.02, dtype=torch.bfloat16, num_beams=args.beam,
                                               resid_scale=rs, align_to=draft_a, **kw)
            if args.target_fp8:
                target_a.enable_fp8()
            if args.do_sample:
                for m in (target_a, draft_a):
                    m.generation_config.do_sample = True
                    m.generation_config.temperature = args.temperature

            def run_aligned(lo, hi):
                res = []
                for g in range(lo, hi, grp):
                    res += BSSD_batch(target_a, draft_a, dprompts[g:min(hi, g + grp)], args.gamma, args.new_tokens, prefix_allowed_tokens_fn=fn)
                return res
            run_aligned(0, min(n_local, max(n_warm, grp)))
            torch.cuda.synchronize(dev)
            ta = time.perf_counter()
            ro = run_aligned(n_warm, n_local)
            torch.cuda.synchronize(dev)
            dta = time.perf_counter() - ta
            br = dict(resid_scale=rs, items_per_s=n_timed * args.beam / dta, ms_per_user=1e3 * dta / n_timed,
                      mean_accept_len=sum(o["total_accept_steps"] for o in ro) / max(1, sum(o["n_run"] for o in ro)),
                      n_run_per_user=sum(o["n_run"] for o in ro) / n_timed,
                      target_forwards_per_user=sum(o["n_target_forwards"] for o in ro) / n_timed)
            if rank == 0 and not args.no_cpu_baseline and not args.do_sample and args.aligned_oracle_users > 0:
                # the accepted length of the CPU oracle (fp32) on EXACTLY these aligned weights and users, next to the engine's (bf16):
                # the non-zero data points behind "mean accepted length >= the reference's"
                cb_a, ref_a, _models = cpu_baseline(target_a, draft_a, prompts[n_warm:], fn, args, n_users=args.aligned_oracle_users)
                na = len(ref_a)
                del _models
                br["oracle"] = dict(users=na, mean_accept_len=cb_a["mean_accept_len"], accept_steps=[[r["n_matches"] for r in o["rounds"]] for o in ref_a],
                                    gpu_mean_accept_len_same_users=sum(o["total_accept_steps"] for o in ro[:na]) / max(1, sum(o["n_run"] for o in ro[:na])),
                                    gpu_accept_steps_same_users=[o["accept_steps"] for o in ro[:na]], items_per_s=cb_a["value"], sample=cb_a["sample"])
            aligned.append(br)
            release_decoders(target_a, draft_a)
            del target_a, draft_a, run_aligned

    # ---- the verify step's scan (full-vocabulary log-sum-exp over the packed logit rows of one lock-step round, the HBM-bound
    # kernel of beamSD.py:285): rows = users x (1 + 3*DK) at V fp32 logits, timed alone with events on its launch stream
    scan = None
    if rank == 0:
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
        # in situ: since round 2 the batched forwards take the normaliser out of the lm_head GEMM's epilogue and write only the logit tiles that
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
        if scan is not None:
            scan["peak_measured"], scan["frac_of_measured"] = gbs.value, scan["achieved"] / gbs.value

    per_rank = all_gather_counters(Counters(n_timed, n_run, acc, int(elapsed * 1e9)), dev)   # the path's single collective
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    agg = aggregate(per_rank, args.beam)
    users, t_max, value, mean_accept = agg["users"], agg["elapsed_s"], agg["items_per_s"], agg["mean_accept_len"]
    total_runs = sum(c.n_run for c in per_rank)

    # ---- roofline of the dominant GEMM kind.  One user at a time (M ~ 100-230 tokens) the launch is a single pass over
    # the weights: HBM-bound.  With lock-step batching M = tokens of all users (thousands): arithmetic intensity is far
    # above the ridge (2.5 PF / 8 TB/s = 312 flop/B) and the bound is the bf16 MFMA peak.
    kind = max(prof, key=lambda k: prof[k]["ms"])
    one_kernel = prof_big[kind]["count"] > 0          # launches of >= 1024 tokens: exactly the 256x256 ring kernel
    pk = prof_big[kind] if one_kernel else prof[kind]
    N, K = target.gemm_shape(kind)
    avg_m = pk["rows"] / max(1, pk["count"])
    n_out = N // 2 if kind == "gate_up" else N
    out_b = 4 if kind == "lm_head" else 2
    in_b = 1 if (args.target_fp8 and kind != "lm_head") else 2        # operand bytes: e4m3 or bf16
    alg_bytes = N * K * in_b + avg_m * K * in_b + avg_m * n_out * out_b
    alg_flops = 2.0 * avg_m * N * K
    avg_ms = pk["ms"] / max(1, pk["count"])
    gemm_ms_total = sum(v["ms"] for v in prof.values())
    intensity = alg_flops / alg_bytes
    if intensity >= MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
        achieved = alg_flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        # the dense MFMA peak of the arithmetic type (guide: ~2.5 PF bf16, ~5 PF fp8 through the block-scaled MFMA forms)
        bound, peak, unit = "mfma", (MFMA_FP8_PEAK_TFLOPS if (args.target_fp8 and kind != "lm_head") else MFMA_PEAK_TFLOPS), "TFLOP/s"
    else:
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        bound, peak, unit = "hbm", HBM_PEAK_GBS, "GB/s"
    # qkv: 5 = RoPE + KV scatter in the epilogue (head_dim 128 targets), as the engine's counter says
    epi = {"qkv": 5 if target.rope_fused_launches() > 0 else 0, "o_proj": 2, "gate_up": 3, "down": 2, "lm_head": 1}[kind]
    ring = (f"gemm_ring_mx_kernel<{epi}, 8>" if (args.target_fp8 and kind != "lm_head" and os.environ.get("ATSPEED_FP8_MX", "1") != "0")
            else f"gemm_ring_kernel<{4 if kind == 'lm_head' else epi}, 8, {'true' if (args.target_fp8 and kind != 'lm_head') else 'false'}, false, 4>")
    kname = (f"{ring} [{kind}] N={N} K={K} avg_M={avg_m:.0f} (launches of >= 1024 tokens)"
             if one_kernel else f"projection GEMM [{kind}] N={N} K={K} avg_M={avg_m:.0f} (all launches)")
    # PMC traffic was collected on the bf16 headline workload: it says nothing about the fp8 kernels or other batch shapes
    traffic, traffic_source = traffic_from_profiles(kind) if (one_kernel and not args.target_fp8 and args.streams == 256) else (None, None)
    roofline = dict(bound=bound, kernel=kname,
                    achieved=achieved, peak=peak, unit=unit, frac=achieved / peak,
                    peak_measured=(measured["mfma_bf16_tflops"] if bound == "mfma" else measured["hbm_read_gbs"]) if measured else None,
                    frac_of_measured=(achieved / (measured["mfma_bf16_tflops"] if bound == "mfma" else measured["hbm_read_gbs"])) if measured else None,
                    traffic=traffic, traffic_source=traffic_source, avg_launch_us=avg_ms * 1e3, launches=pk["count"],
                    algorithmic_bytes_per_launch=alg_bytes, algorithmic_flops_per_launch=alg_flops,
                    arithmetic_intensity=intensity,
                    target_forward=dict(avg_ms_per_user=1e3 * stage[1] / max(1, n_tf),
                                        weight_bytes=tdims.n_params_streamed() * 2,
                                        gemm_ms_share={k: v["ms"] / gemm_ms_total for k, v in prof.items()} if gemm_ms_total else {},
                                        all_gemms_tflops=(sum(2.0 * v["rows"] * target.gemm_shape(k)[0] * target.gemm_shape(k)[1] for k, v in prof.items())
                                                          / (gemm_ms_total * 1e-3) / 1e12) if gemm_ms_total else 0.0))

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
                   "parallelism": f"user-shard x{world}",
                   "operand_layout": "packed (row pairs per 128-byte line)" if getattr(target, "weights_packed", False) else "row-major"},
        "mean_accept_len": mean_accept,
        "accept_note": "unrelated random draft/target weights accept ~0 draft steps: worst-case bracket (3 verify rounds + 1 final step = 4 target forwards per user)",
        "per_user": {"n_run": total_runs / users, "target_forwards": n_tf / n_timed, "draft_forwards": n_df / n_timed,
                     "draft_ms": 1e3 * stage[0] / n_timed, "target_ms": 1e3 * stage[1] / n_timed, "verify_ms": 1e3 * stage[2] / n_timed},
        "roofline": roofline,
        "measured_peaks": measured,
        "verify_scan": scan,
        "single_user_stream": single,
        "aligned_weight_brackets": aligned,   # same users, shapes and kernels as `value`; only the weights' agreement differs
    }
    if args.do_sample:
        line["decoding"] = f"sampling (temperature {args.temperature})"
    if world == 1 and not args.no_cpu_baseline and not args.do_sample:
        cb, ref_outs, (ref_t, _ref_d) = cpu_baseline(target, draft, prompts[n_warm:], fn, args)
        line["cpu_baseline"] = cb
        # next to the timing: the bf16 engine's items vs the fp32 oracle's on identical weights.  NOT the parity test (tests/test_fulldims_gpu.py
        # runs the fp32 engine at these dims against the oracle bit for bit): here every rank at which bf16 and fp32 disagree is scored by the
        # oracle itself, so that "near-ties of flat random-init logits" is a measured statement
        P0 = len(prompts[n_warm])
        rep = disagreement_report(outs[0], ref_outs[0], P0, ref_t, prompts[n_warm])
        line["cpu_baseline"]["top_k_overlap_with_gpu_bf16"] = rep["top_k_overlap"]
        line["cpu_baseline"]["bf16_vs_fp32_disagreements"] = rep
        line["cpu_baseline"]["gpu_accept_len_same_users"] = float(sum(o["total_accept_steps"] for o in outs[:len(ref_outs)])) / max(
            1, sum(o["n_run"] for o in outs[:len(ref_outs)]))
        del ref_t, _ref_d
    else:
        line["cpu_baseline"] = None
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
