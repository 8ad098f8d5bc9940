"""Child process of tests/test_env_switches_gpu.py: the A/B switches of the library are read once per process, so each configuration gets a fresh
one.  Decodes the same users with the bf16 engine (one user per call and a 24-user lock-step batch) and with the W8A8 target, prints one JSON
line: item ids, scores, rounds and accepted steps per user.   usage: python -m tests.env_switch_worker   (the switches come in the environment)"""
import json
import sys

import torch


def main():
    from atspeed_amd import synth
    from atspeed_amd.beamSD import BSSD, BSSD_batch, release_decoders
    from atspeed_amd.generation_trie import PositionSetConstraint
    from atspeed_amd.model import HipLlama
    V = synth.BEAUTY.vocab_size
    kw = dict(max_slots=512, max_tokens=512, max_logit_rows=448, device="cuda:0")
    # hidden 2048 / ffn 5632 / 16 heads x 128: every 16-bit and fp8 kernel family applies (weight-streaming, split forms, ring, 32-rows-per-wave
    # attention); draft and target aligned (3e-5: about one accepted step) with peaked heads, so that a kernel switch moves scores, not decisions
    ddims = synth.LlamaDims(V, 256, 2, 4, 704)
    tdims = synth.LlamaDims(V, 2048, 2, 16, 5632)
    drf = HipLlama.from_synthetic(ddims, 52, std=0.03, head_std=0.2, dtype=torch.bfloat16, num_beams=40, resid_scale=3e-5, **kw)
    tgt = HipLlama.from_synthetic(tdims, 51, std=0.03, head_std=0.2, dtype=torch.bfloat16, num_beams=20, resid_scale=3e-5, align_to=drf, **kw)
    fn = PositionSetConstraint(synth.BEAUTY.allowed_tokens(), synth.RESPONSE_SEP)
    prompts = [{"input_ids": torch.from_numpy(synth.synthetic_prompt(60 + 7 * (u % 9), 900 + u))[None].cuda()} for u in range(24)]
    out = {}

    def pack(o, P):
        return dict(items=o["beam_sequence"][:, P:].cpu().tolist(), scores=[float(x) for x in o["beam_scores"].cpu().tolist()], n_run=int(o["n_run"]),
                    accept=int(o["total_accept_steps"]))

    def run(tag):
        out[tag + "_one"] = [pack(BSSD(tgt, drf, p, 4, 4, prefix_allowed_tokens_fn=fn), p["input_ids"].shape[1]) for p in prompts[:3]]
        out[tag + "_batch"] = [pack(o, p["input_ids"].shape[1]) for o, p in zip(BSSD_batch(tgt, drf, prompts, 4, 4, prefix_allowed_tokens_fn=fn), prompts)]

    run("bf16")
    release_decoders(tgt, drf)
    tgt.enable_fp8()
    tgt.fp8_counters(reset=True)
    run("fp8")
    out["fp8_counters"] = tgt.fp8_counters()
    torch.cuda.synchronize()
    print("ENVSWITCH " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
    sys.exit(0)
