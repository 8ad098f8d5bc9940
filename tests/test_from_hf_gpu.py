"""The HF entry (`HipLlama.from_hf`, what INTEGRATION.md tells a maintainer to call in place of the reference's
`LlamaForCausalLM.from_pretrained(..., torch_dtype=torch.float16)`, code/inference.py:75-100): a random-init transformers
LlamaForCausalLM at the Llama-68M dims in fp16 and in bf16 goes through `from_hf`; the engine's logits on a tree mask with a KV cache
are compared with the HF module's own (fp32 arithmetic on the same weight values), then one BSSD call against the oracle.
dtype policy: a checkpoint runs in its own type -- fp16 (what the reference loads) on the engine's fp16 flavour with every weight bit kept,
bf16 on bf16, either in fp32 or fp16 -> bf16 BY VALUE on request; never reinterpreted."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import atspeed_amd
from atspeed_amd import synth
from atspeed_amd.beamSD import BSSD, target_generate
from atspeed_amd.model import HipLlama, vis_bits_from_bool

LOGIT_TOL = 1e-3          # north star: fp32 logits within 1e-3
BF16_MAX_TOL, BF16_MEAN_TOL = 0.04, 0.006      # of max|logit| (tests/test_fulldims_gpu.py)
F16_MAX_TOL, F16_MEAN_TOL = BF16_MAX_TOL / 4, BF16_MEAN_TOL / 4       # fp16 carries 3 more significand bits than bf16 (<= a quarter)
V = synth.BEAUTY.vocab_size


def _hf(dtype, seed, layers=2, kv_heads=12):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(vocab_size=V, hidden_size=768, intermediate_size=3072, num_hidden_layers=layers, num_attention_heads=12,
                      num_key_value_heads=kv_heads, rms_norm_eps=1e-6, attn_implementation="eager", tie_word_embeddings=False,
                      max_position_embeddings=2048)
    torch.manual_seed(seed)
    m = LlamaForCausalLM(cfg).eval()
    with torch.no_grad():
        for p in m.parameters():                       # HF's init is N(0, 0.02): widen it so logits are not flat
            if p.dim() == 2:
                p.mul_(2.0)
    # like from_pretrained(torch_dtype=...): parameters in the checkpoint dtype, the rotary inv_freq buffer stays the fp32 it was computed in
    # (a blanket .to(fp16) would round it to 11 bits and change every rotation angle)
    inv = m.model.rotary_emb.inv_freq.clone()
    m = m.to(dtype)
    m.model.rotary_emb.inv_freq = inv
    if hasattr(m.model.rotary_emb, "original_inv_freq"):
        m.model.rotary_emb.original_inv_freq = inv.clone()
    return m


def _hf_logits(hf, P, B, g):
    """prompt forward (causal) then B tree tokens over the cached prompt with one hidden slot: what beamSD.py:52,221 ask of the model"""
    from transformers.cache_utils import DynamicCache
    ref = hf.float()
    ids0 = torch.randint(3, 32000, (P,), generator=g)
    ids1 = torch.randint(32000, V, (B,), generator=g)
    neg = torch.finfo(torch.float32).min
    vis1 = torch.cat((torch.ones(B, P, dtype=torch.bool), torch.eye(B, dtype=torch.bool)), 1)
    vis1[:, 5] = False
    with torch.no_grad():
        cache = DynamicCache(config=ref.config)
        m0 = (torch.tril(torch.ones(P, P)) == 0) * neg
        o0 = ref(input_ids=ids0[None], attention_mask=m0[None, None], position_ids=torch.arange(P)[None], past_key_values=cache, use_cache=True)
        o1 = ref(input_ids=ids1[None], attention_mask=((~vis1) * neg)[None, None].float(), position_ids=torch.full((1, B), P),
                 past_key_values=o0.past_key_values, use_cache=True)
    return ids0, ids1, vis1, o0.logits[0], o1.logits[0]


def _engine_logits(m, ids0, ids1, vis1):
    P, B = len(ids0), len(ids1)
    i32 = lambda t: t.to(torch.int32).cuda()
    causal = torch.tril(torch.ones(P, P, dtype=torch.bool))
    l0 = m.forward_raw(i32(ids0), i32(torch.arange(P)), i32(torch.arange(P)), vis_bits_from_bool(causal, m.max_slots).cuda(), P, P)
    l1 = m.forward_raw(i32(ids1), i32(torch.full((B,), P)), i32(torch.arange(P, P + B)), vis_bits_from_bool(vis1, m.max_slots).cuda(), P + B, B)
    return l0.float().cpu(), l1.float().cpu()


@pytest.mark.parametrize("ckpt_dtype", [torch.float16, torch.bfloat16], ids=["fp16_checkpoint", "bf16_checkpoint"])
def test_from_hf_converts_by_value_and_matches_the_hf_module(ckpt_dtype):
    hf = _hf(ckpt_dtype, 3)
    assert hf.dtype == ckpt_dtype
    kw = dict(max_slots=256, max_tokens=256, max_logit_rows=128)
    exact = HipLlama.from_hf(hf, dtype=torch.float32, **kw)              # fp32 engine on exactly the checkpoint's values
    fast = HipLlama.from_hf(hf, **kw)                                    # default: the checkpoint's own type (fp16 -> the fp16 flavour)
    assert exact.dtype == torch.float32 and fast.dtype == ckpt_dtype
    ids0, ids1, vis1, r0, r1 = _hf_logits(hf, 40, 9, torch.Generator().manual_seed(1))
    e0, e1 = _engine_logits(exact, ids0, ids1, vis1)
    print("fp32 engine vs HF fp32 on the", ckpt_dtype, "values: max diff", float((e0 - r0).abs().max()), float((e1 - r1).abs().max()),
          "max|logit|", float(r0.abs().max()))
    np.testing.assert_allclose(e0.numpy(), r0.numpy(), atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(e1.numpy(), r1.numpy(), atol=LOGIT_TOL, rtol=0)
    max_tol, mean_tol = (F16_MAX_TOL, F16_MEAN_TOL) if ckpt_dtype == torch.float16 else (BF16_MAX_TOL, BF16_MEAN_TOL)
    f0, f1 = _engine_logits(fast, ids0, ids1, vis1)
    for got, want in ((f0, r0), (f1, r1)):
        scale = float(want.abs().max())
        err = (got - want).abs()
        print(ckpt_dtype, "engine: max err", float(err.max()) / scale, "mean err", float(err.mean()) / scale)
        assert float(err.max()) < max_tol * scale and float(err.mean()) < mean_tol * scale
        # a bit-reinterpreted checkpoint would be off by orders of magnitude, not per cent
    # the weights ARE the checkpoint's values, bit for bit
    w_hf = hf.state_dict()["model.layers.0.self_attn.o_proj.weight"].float()
    w_eng = fast.export_state_dict()["model.layers.0.self_attn.o_proj.weight"]
    assert torch.equal(w_eng, w_hf.cpu())
    if ckpt_dtype == torch.float16:
        # fp16 -> bf16 by value on request (the round-1..3 default): relative error <= 2^-9, and a looser fit to the HF module
        conv = HipLlama.from_hf(hf, dtype=torch.bfloat16, **kw)
        assert conv.dtype == torch.bfloat16
        w_c = conv.export_state_dict()["model.layers.0.self_attn.o_proj.weight"]
        assert float(((w_c - w_hf.cpu()).abs() / w_hf.cpu().abs().clamp_min(1e-6)).max()) <= 2.0 ** -8
        c0, _ = _engine_logits(conv, ids0, ids1, vis1)
        e_f16, e_bf = float((f0 - r0).abs().mean()), float((c0 - r0).abs().mean())
        print("mean |logit error| fp16 engine", e_f16, "bf16-converted engine", e_bf)
        assert e_f16 < 0.5 * e_bf


def test_fp16_engine_is_lossless_and_close_to_the_oracle():
    """The engine in the reference's own dtype (inference.py:75-100 loads fp16): BSSD on an fp16 pair returns exactly what the same engine's
    plain beam search returns (the method's lossless property, beamSD.py:544-595), one user and a lock-step batch, and its top-K items are
    the fp32 oracle's on the same weight values wherever the oracle's margins exceed fp16 noise."""
    from atspeed_amd.beamSD import BSSD_batch, release_decoders
    from oracle import beamsd_ref as R
    from oracle.llama_ref import RefLlama
    t_hf, d_hf = _hf(torch.float16, 5), _hf(torch.float16, 6)
    with torch.no_grad():
        for pt, pd in zip(t_hf.parameters(), d_hf.parameters()):
            pd.copy_((pt.float() + 0.1 * pd.float()).to(torch.float16))
    t_hf.generation_config.num_beams, d_hf.generation_config.num_beams = 20, 40
    kw = dict(max_slots=512, max_tokens=512, max_logit_rows=448)
    tgt, drf = HipLlama.from_hf(t_hf, **kw), HipLlama.from_hf(d_hf, **kw)
    assert tgt.dtype == drf.dtype == torch.float16 and tgt.weights_packed
    fn = atspeed_amd.PositionSetConstraint(synth.BEAUTY.allowed_tokens(), synth.RESPONSE_SEP)
    prompts = [synth.synthetic_prompt(50 + 7 * u, 9 + u) for u in range(6)]
    ins = [{"input_ids": torch.from_numpy(p)[None].cuda()} for p in prompts]
    sd = lambda m: {k: v.float().numpy() for k, v in m.state_dict().items() if "rotary" not in k}
    rt, rd = RefLlama(tgt.dims, sd(t_hf), max_slots=512), RefLlama(drf.dims, sd(d_hf), max_slots=512)
    bat = BSSD_batch(tgt, drf, ins, 4, 4, prefix_allowed_tokens_fn=fn)
    overlap = []
    for u, (p, inp) in enumerate(zip(prompts, ins)):
        out = BSSD(tgt, drf, inp, 4, 4, prefix_allowed_tokens_fn=fn)
        tg = target_generate(tgt, inp, 4, prefix_allowed_tokens_fn=fn)
        assert torch.equal(tg["beam_sequence"], out["beam_sequence"]), u                 # lossless
        assert sorted(map(tuple, bat[u]["beam_sequence"].cpu().tolist())) == sorted(map(tuple, out["beam_sequence"].cpu().tolist())), u
        ref = R.BSSD(rt, rd, p, 4, 4, 20, 40, fn)
        P = len(p)
        a, b = {tuple(x) for x in out["beam_sequence"][:, P:].cpu().tolist()}, {tuple(x) for x in ref["beam_sequence"][:, P:].tolist()}
        overlap.append(len(a & b) / 20)
        np.testing.assert_allclose(np.sort(out["beam_scores"].cpu().numpy()), np.sort(ref["beam_scores"].numpy()), atol=5e-2, rtol=0)
    print("fp16 engine vs fp32 oracle: top-20 overlap per user", overlap)
    assert min(overlap) >= 0.8 and sum(overlap) / len(overlap) >= 0.9
    release_decoders(tgt, drf)


def test_from_hf_pair_runs_bssd_like_the_oracle():
    from oracle import beamsd_ref as R
    from oracle.llama_ref import RefLlama
    t_hf, d_hf = _hf(torch.float16, 5), _hf(torch.float16, 6)
    with torch.no_grad():                                                # a draft that agrees with the target part of the time
        for pt, pd in zip(t_hf.parameters(), d_hf.parameters()):
            pd.copy_((pt.float() + 0.1 * pd.float()).to(torch.float16))
    t_hf.generation_config.num_beams, d_hf.generation_config.num_beams = 20, 40
    kw = dict(max_slots=512, max_tokens=512, max_logit_rows=448)
    tgt = HipLlama.from_hf(t_hf, dtype=torch.float32, **kw)
    drf = HipLlama.from_hf(d_hf, dtype=torch.float32, **kw)
    assert tgt.generation_config.num_beams == 20 and drf.generation_config.num_beams == 40
    fn = atspeed_amd.PositionSetConstraint(synth.BEAUTY.allowed_tokens(), synth.RESPONSE_SEP)
    prompt = synth.synthetic_prompt(60, 9)
    inputs = {"input_ids": torch.from_numpy(prompt)[None].cuda()}
    out = BSSD(tgt, drf, inputs, 4, 4, prefix_allowed_tokens_fn=fn)
    tg = target_generate(tgt, inputs, 4, prefix_allowed_tokens_fn=fn)
    sd = lambda m: {k: v.float().numpy() for k, v in m.state_dict().items() if "rotary" not in k}
    ref = R.BSSD(RefLlama(tgt.dims, sd(t_hf), max_slots=512), RefLlama(drf.dims, sd(d_hf), max_slots=512), prompt, 4, 4, 20, 40, fn)
    P = len(prompt)
    assert out["beam_sequence"][:, P:].cpu().tolist() == ref["beam_sequence"][:, P:].tolist()
    np.testing.assert_allclose(out["beam_scores"].cpu().numpy(), ref["beam_scores"].numpy(), atol=1e-3, rtol=0)
    assert (out["n_run"], out["total_accept_steps"]) == (ref["n_run"], ref["total_accept_steps"])
    assert torch.equal(tg["beam_sequence"], out["beam_sequence"])


def test_unsupported_checkpoints_and_dtypes_raise():
    hf = _hf(torch.float16, 7, layers=1)
    with pytest.raises(TypeError):
        HipLlama.from_hf(hf, dtype=torch.float64)                        # only fp32 / bf16 / fp16 arithmetic exists: nothing is reinterpreted
    with pytest.raises(TypeError):
        HipLlama.from_state_dict(synth.llama_68m(V), {}, torch.int8)
    with pytest.raises(TypeError):
        HipLlama.from_synthetic(synth.llama_68m(V), 1, dtype=torch.float64)
    with pytest.raises(atspeed_amd._lib.AtSpeedError):
        HipLlama.from_synthetic(synth.llama_68m(V), 1, dtype=torch.float32, max_slots=256, max_tokens=256, max_logit_rows=128).enable_fp8()   # fp8 copies are made of 16-bit weights (bf16 or, since round 6, fp16)
    with pytest.raises(NotImplementedError):
        HipLlama.from_hf(_hf(torch.float16, 8, layers=1, kv_heads=4))     # grouped-query attention
