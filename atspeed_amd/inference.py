"""`python -m atspeed_amd.inference` — the driver of `code/inference.py` for the beam-SD path on MI355X.

Same job and flag names as the reference script for this path (`--dataset --data_path --index_file --max_his_len
--gamma --draft_beam_size --run_beam_sizes --L --R --seed`; `code/utils.py:41-52,134-147`): load the test split, decode
users [L, R) with BSSD, write the mean timing / acceptance columns of `timing_mean_*.csv` (inference.py:146-156,189)
and, which the reference defines but never wires in, the ranking metrics of the returned items.  Under
`torch.distributed.run` each rank takes a contiguous shard of the users; ranks exchange one all-gather of counters
and one of metric sums.

Weights: `--target_ckpt/--draft_ckpt` point at HF Llama checkpoints when they exist on the machine; without them the
models are the synthetic hash-PRNG Llama-7B / Llama-68M pair of bench.py (`--aligned` makes the draft agree with the
target so that acceptance is non-trivial).  A tokenizer (`--tokenizer`, any HF path) is used when given; otherwise
prompts are laid out as code-token ids (`harness.CodeTokenEncoder`).
"""
from __future__ import annotations

import argparse
import ast
import json
import os
import sys

import torch


def parse(argv=None):
    ap = argparse.ArgumentParser(description="AtSpeed_inference (MI355X)")
    ap.add_argument("--seed", type=int, default=2025)
    ap.add_argument("--data_path", type=str, required=True)
    ap.add_argument("--dataset", type=str, default="games")
    ap.add_argument("--index_file", type=str, default=".LCRec-1e-3lr.json")
    ap.add_argument("--max_his_len", type=int, default=20)
    ap.add_argument("--add_prefix", action="store_true")
    ap.add_argument("--his_sep", type=str, default=", ")
    ap.add_argument("--gamma", type=int, default=4)
    ap.add_argument("--run_beam_sizes", type=str, default="[20]")
    ap.add_argument("--draft_beam_size", type=int, default=40)
    ap.add_argument("--L", type=int, default=0)
    ap.add_argument("--R", type=int, default=None)
    ap.add_argument("--users_per_batch", type=int, default=128, help="users decoded in lock step (1 = the reference's loop)")
    ap.add_argument("--strict_trie", action="store_true", help="strict item trie instead of the position-set mask")
    ap.add_argument("--decoder", choices=("bssd", "beam"), default="bssd",
                    help="bssd = beam speculative decoding (the reference's method); beam = plain constrained beam search of the target in lock "
                         "step: identical items, and on MI355X the faster of the two once a batch of users makes the target forward MFMA-bound "
                         "(bench.py speedup_curve)")
    ap.add_argument("--target_ckpt", type=str, default=None)
    ap.add_argument("--draft_ckpt", type=str, default=None)
    ap.add_argument("--tokenizer", type=str, default=None)
    ap.add_argument("--target_layers", type=int, default=32)
    ap.add_argument("--aligned", type=float, default=None, metavar="RESID_SCALE", help="synthetic weights: align draft and target (see bench.py)")
    ap.add_argument("--target_fp8", action="store_true",
                    help="W8A8 (e4m3) target projections -- the place of the reference's `load_in_8bit` target (inference.py:86-91); works on bf16 and "
                         "fp16 targets (an fp16 checkpoint under --dtype auto keeps its type), not with --dtype fp32")
    ap.add_argument("--dtype", choices=("auto", "fp16", "bf16", "fp32"), default="auto",
                    help="engine arithmetic.  auto: a checkpoint runs in the type it is stored in (fp16 -- what the reference loads, inference.py:75-100 -- "
                         "takes the engine's fp16 flavour and keeps every weight bit; bf16 and fp32 likewise), synthetic weights are bf16")
    ap.add_argument("--baseline", action="store_true", help="also time target_generate per user (speedup / overhead columns)")
    ap.add_argument("--output_dir", type=str, default="AnaResult")
    args = ap.parse_args(argv)
    if args.target_fp8 and args.dtype == "fp32":
        ap.error("--target_fp8 makes its e4m3 copies from 16-bit weights: use --dtype auto, fp16 or bf16")
    return args


def load_models(args, vocab_size: int, beam: int, dev, max_prompt: int = 0):
    from . import synth
    from .harness import capacity_for
    from .model import HipLlama
    # KV arenas, token buffers and logit rows sized from the longest prompt of the users this rank decodes (the reference cuts prompts at
    # cutoff_len = 512 tokens, code/utils.py:119: 512 + 3 x 40 draft tokens no longer fit the library's default 512 slots)
    kw = dict(capacity_for(max_prompt, beam, args.draft_beam_size, args.gamma, 4), device=dev)
    want = {"auto": None, "fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[args.dtype]
    if args.target_ckpt and args.draft_ckpt:
        from transformers import AutoModelForCausalLM
        # torch_dtype="auto": the checkpoint's own type (no rounding on load); from_hf(dtype=None) then picks the engine flavour of that type
        load = lambda path: AutoModelForCausalLM.from_pretrained(path, torch_dtype="auto" if want is None else want)
        kw_hf = {k: v for k, v in kw.items() if k != "device"}
        tgt = HipLlama.from_hf(load(args.target_ckpt), want, dev, num_beams=beam, **kw_hf)
        drf = HipLlama.from_hf(load(args.draft_ckpt), want, dev, num_beams=args.draft_beam_size, **kw_hf)
    else:
        rs = 1.0 if args.aligned is None else args.aligned
        syn = torch.bfloat16 if want is None else want
        drf = HipLlama.from_synthetic(synth.llama_68m(vocab_size), args.seed + 1, dtype=syn, num_beams=args.draft_beam_size, resid_scale=rs, **kw)
        tgt = HipLlama.from_synthetic(synth.llama_7b(vocab_size, args.target_layers), args.seed, dtype=syn, num_beams=beam, resid_scale=rs,
                                      align_to=drf if args.aligned is not None else None, **kw)
    if args.target_fp8:
        tgt.enable_fp8()
    return tgt, drf


def main(argv=None):
    args = parse(argv)
    from .beamSD import release_decoders
    from .dist import Counters, aggregate, all_gather_counters, shard_range
    from .harness import SeqRecTestData, longest_prompt, reduce_metrics, run_inference

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("atspeed_amd.inference needs a HIP device (no CPU path)")
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    data = SeqRecTestData.load(args.data_path, args.dataset, args.index_file, max_his_len=args.max_his_len, add_prefix=args.add_prefix,
                               his_sep=args.his_sep)
    tok = None
    if args.tokenizer:
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(args.tokenizer)
    stop_r = min(len(data), args.R) if args.R is not None else len(data)
    lo, hi = shard_range(stop_r - args.L, rank, world)
    fn = data.strict_trie_fn() if args.strict_trie else data.get_prefix_allowed_tokens_fn()
    summary = []
    max_prompt = longest_prompt(data, args.L + lo, args.L + hi, tok)
    for beam in ast.literal_eval(args.run_beam_sizes):          # the reference eval()s this flag (inference.py:151); a literal list is all it needs
        tgt, drf = load_models(args, data.index.vocab_size, beam, dev, max_prompt)
        res = run_inference(tgt, drf, data, args.gamma, 4, args.L + lo, args.L + hi, args.users_per_batch, fn, tok, args.baseline, dev, args.decoder)
        c = res.counters()
        per_rank = all_gather_counters(Counters(len(res.rows), int(sum(r["n_run"] for r in res.rows)),
                                                int(sum(r["total_accept_steps"] for r in res.rows)), int(res.wall_s * 1e9)), dev)
        metrics = reduce_metrics(res, data.index)
        if rank == 0:
            agg = aggregate(per_rank, beam)
            row = {"dataset": args.dataset, "beam_size": beam, "draft_beam_size": args.draft_beam_size, "gamma": args.gamma,
                   "users": agg["users"], "items_per_s": agg["items_per_s"], "mean_accept_len": agg["mean_accept_len"],
                   "timing_mean_rank0": res.timing_mean(), "metrics": metrics}
            summary.append(row)
            os.makedirs(os.path.join(args.output_dir, args.dataset), exist_ok=True)
            name = f"timing_mean_B{beam}-{args.draft_beam_size}_{args.L}-{stop_r}_seed{args.seed}.json"
            with open(os.path.join(args.output_dir, args.dataset, name), "w") as f:
                json.dump(row, f, indent=1)
            print(json.dumps(row))
        release_decoders(tgt, drf)        # per-lane KV arenas of this beam size (the cache only holds weak references to the models)
        del tgt, drf, res
    if world > 1:
        torch.distributed.destroy_process_group()
    return summary


if __name__ == "__main__":
    main(sys.argv[1:])
