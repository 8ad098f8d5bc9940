#!/usr/bin/env python3
"""CPU experiment behind tests/test_fulldims_gpu.py's "decidable" recipe: at the FULL Llama-7B(32L) / Llama-68M dims, which
synthetic weight recipe gives every top-k decision of the oracle's BSSD a margin far above fp32 summation noise WITH the 32 layers mattering?

Candidates of a decision are (beam, token) pairs scored s_beam + logp_beam[token]; a step sorts 20-40 winners out of up to 40 x 256 of them, so a
user makes ~10 decisions x 40 adjacent gaps, twelve users ~5000 gaps.  With Gaussian logits (any head_std: margins and noise scale together) the
smallest of 5000 gaps sits at the noise level -- the valve of round 4.  Heavy-tailed logits (each code token's head row scaled by
exp(beta * z_t), z_t ~ N(0, 1): a few tokens dominate, as in a trained recommender) space the winners out.

Weights here come from torch.randn (fast); the recipe's statistics, not these exact weights, are what carries over to the device-generated
model.  usage: python tools/margin_search.py [users] [beta,...] [resid_scale] [layers] [K/DK,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from atspeed_amd import synth
from atspeed_amd.generation_trie import PositionSetConstraint
from oracle import beamsd_ref as R
from oracle.llama_ref import RefLlama

users = int(sys.argv[1]) if len(sys.argv) > 1 else 4
betas = [float(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1.5").split(",")]
rs = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
layers = int(sys.argv[4]) if len(sys.argv) > 4 else 32
beams = [tuple(int(v) for v in kd.split("/")) for kd in (sys.argv[5] if len(sys.argv) > 5 else "20/40").split(",")]   # K/DK pairs
torch.set_num_threads(8)
V = synth.BEAUTY.vocab_size


def make(dims, seed, resid_scale):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape, kind in synth.weight_specs(dims):
        if kind == "norm":
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            s = 0.02 * (resid_scale if ("o_proj" in name or "down_proj" in name) else 1.0)
            sd[name] = torch.randn(shape, generator=g) * s
    return sd


t0 = time.time()
tdims, ddims = synth.llama_7b(V, layers), synth.llama_68m(V)
tsd, dsd = make(tdims, 1, rs), make(ddims, 2, rs)
print(f"weights in {time.time() - t0:.0f}s", flush=True)
fn = PositionSetConstraint(synth.BEAUTY.allowed_tokens(), synth.RESPONSE_SEP)
z = torch.from_numpy(synth.hash_normal(V, 777, 1.0))
head_t, head_d = tsd["lm_head.weight"].clone(), dsd["lm_head.weight"].clone()
PROMPTS = (108, 70, 66, 84, 96, 78, 120, 150, 186, 72, 102, 132)
for beta in betas:
    a = torch.ones(V)
    a[32000:] = torch.exp(beta * z[32000:])
    tsd["lm_head.weight"] = head_t * a[:, None]
    dsd["lm_head.weight"] = head_d * a[:, None]
    rt, rd = RefLlama(tdims, tsd, max_slots=512), RefLlama(ddims, dsd, max_slots=512)
    for K, DK in beams:
        mins = []
        for u in range(users):
            P = PROMPTS[u % len(PROMPTS)]
            prompt = synth.synthetic_prompt(P, synth.tensor_seed(2025, f"user{u}"))
            R.MARGINS = []
            t1 = time.time()
            ref = R.BSSD(rt, rd, prompt, 4, 4, K, DK, fn)
            m, R.MARGINS = sorted(R.MARGINS), None
            mins.append(m[0])
            # fp32 summation noise: the same oracle with another thread count (another blocking of every matmul's sum)
            torch.set_num_threads(3)
            ref2 = R.BSSD(rt, rd, prompt, 4, 4, K, DK, fn)
            torch.set_num_threads(8)
            same = ref2["beam_sequence"].tolist() == ref["beam_sequence"].tolist()
            noise = float((ref2["beam_scores"] - ref["beam_scores"]).abs().max())
            print(f"   noise proxy (8 vs 3 threads): items equal {same}, max |score diff| {noise:.3e}, min margin / noise {m[0] / max(noise, 1e-12):.1f}", flush=True)
            print(f"beta {beta} resid {rs} K {K} DK {DK} user {u}: n_run {ref['n_run']} accept {ref['total_accept_steps']} decisions {len(m)} min margin {m[0]:.3e} "
                  f"next {m[1]:.3e} {m[2]:.3e} median {m[len(m) // 2]:.3e} best score {float(ref['beam_scores'][0]):.3f} worst {float(ref['beam_scores'][-1]):.3f} ({time.time() - t1:.0f}s)", flush=True)
        print(f"== beta {beta} resid {rs} K {K} DK {DK}: min over users {min(mins):.3e}, users below 1e-4: {sum(x < 1e-4 for x in mins)} of {users}", flush=True)
