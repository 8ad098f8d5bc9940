"""GPU closures of SURVEY.md section 8 rows that had no hardware test: M1 (`prefix_allowed_tokens_fn(trie)` keyed on the whole
sentence, generation_trie.py:92-98), the post-top-k id filter of one_step_beam_search (beamSD.py:80-86) on the device path,
config 3 at its stated batch (Games, strict trie, 256 users in lock step), the RCCL collective, and the decoder cache's
lifetime across a beam-size sweep (inference.py:151)."""
import ctypes as C
import gc
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import atspeed_amd
from atspeed_amd import _lib, synth
from atspeed_amd.beamSD import BSSD, BSSD_batch, _Decoder, last_trace, release_decoders, target_generate
from atspeed_amd.generation_trie import PositionSetConstraint, SuffixTrieConstraint, Trie, prefix_allowed_tokens_fn
from atspeed_amd.model import HipLlama, vis_bits_from_bool
from oracle import beamsd_ref as R
from oracle.llama_ref import RefLlama
from oracle.trie_ref import RefTrie, ref_whole_sentence_fn
from tests.golden.cases import CASES, build_case_inputs

SCORE_TOL = 1e-3


def _models(ci, K, DK, dtype=torch.float32):
    kw = dict(max_slots=512, max_tokens=512, max_logit_rows=448)
    t = HipLlama.from_state_dict(ci["target_dims"], ci["target_sd"], dtype, num_beams=K, **kw)
    d = HipLlama.from_state_dict(ci["draft_dims"], ci["draft_sd"], dtype, num_beams=DK, **kw)
    return t, d


def _same_as_oracle(out, ref, P):
    nv = out["n_valid"]
    assert nv == ref["beam_sequence"].shape[0]
    assert out["beam_sequence"][:nv, P:].cpu().tolist() == ref["beam_sequence"][:, P:].tolist()
    np.testing.assert_allclose(out["beam_scores"][:nv].cpu().numpy(), ref["beam_scores"].numpy(), atol=SCORE_TOL, rtol=0)


@pytest.mark.parametrize("name", ["k20_dk40_sigma01_s7", "k5_dk10_indep"])
def test_whole_sentence_trie_fn_through_the_hip_path(name):
    """M1: the callable `prefix_allowed_tokens_fn(trie)` looks the ENTIRE sentence (prompt included) up in the trie, so the
    trie holds prompt ++ item codes ++ eos.  Compiled to the device automaton (start node = the prompt's node) it must make
    the oracle's decisions (oracle: the same lookups on the host, `ref_whole_sentence_fn`)."""
    case = next(c for c in CASES if c["name"] == name)
    ci = build_case_inputs(case)
    prompt = [int(t) for t in ci["prompt"]]
    seqs = [prompt + [int(t) for t in it] + [2] for it in ci["items"]]
    fn = prefix_allowed_tokens_fn(Trie(seqs))
    ref_fn = ref_whole_sentence_fn(RefTrie(seqs))
    tgt, drf = _models(ci, case["K"], case["DK"])
    P = len(prompt)
    inputs = {"input_ids": torch.from_numpy(ci["prompt"])[None].cuda()}
    rt, rd = RefLlama(ci["target_dims"], ci["target_sd"]), RefLlama(ci["draft_dims"], ci["draft_sd"])
    ref = R.BSSD(rt, rd, ci["prompt"], case["gamma"], case["max_new_tokens"], case["K"], case["DK"], ref_fn)
    out = BSSD(tgt, drf, inputs, case["gamma"], case["max_new_tokens"], prefix_allowed_tokens_fn=fn)
    _same_as_oracle(out, ref, P)
    assert (out["n_run"], out["accept_steps"]) == (ref["n_run"], [r["n_matches"] for r in ref["rounds"]])
    for r, g in zip(last_trace(tgt, drf), ref["rounds"]):
        for ids, gids in zip(r["draft_ids"], g["draft_ids"]):
            assert [x for x in ids if x >= 0] == gids
    tg = target_generate(tgt, inputs, case["max_new_tokens"], prefix_allowed_tokens_fn=fn)
    _same_as_oracle(tg, R.target_generate(rt, ci["prompt"], case["max_new_tokens"], case["K"], ref_fn), P)
    items = {tuple(int(t) for t in it) for it in ci["items"]}
    assert all(tuple(g) in items for g in out["beam_sequence"][: out["n_valid"], P:].cpu().tolist())
    # a prompt the trie does not contain: the reference's fn returns [] and HF raises ValueError
    with pytest.raises(ValueError):
        BSSD(tgt, drf, {"input_ids": torch.tensor([prompt[:-1] + [77]]).cuda()}, 4, 4, prefix_allowed_tokens_fn=fn)


def test_post_topk_id_filter_on_device_host_and_oracle_paths():
    """beamSD.py:80-86 drops picks whose token is < 32000 and != 2 AFTER the top-k (the beam set shrinks, lower-ranked candidates
    do not move up).  A trie that allows such a token makes the filter bite: the compiled device path, the host-callable path
    (hostmask.py) and the oracle must agree, in plain beam search and inside BSSD's draft steps."""
    case = next(c for c in CASES if c["name"] == "k5_dk10_indep")
    ci = build_case_inputs(case)
    prompt = [int(t) for t in ci["prompt"]]
    firsts = sorted({int(it[0]) for it in ci["items"]})[:9]
    items = [next(it for it in ci["items"] if int(it[0]) == f) for f in firsts]
    seqs = [prompt + [int(t) for t in it] + [2] for it in items]
    seqs += [prompt + [7, int(items[0][1]), int(items[0][2]), int(items[0][3]), 2],          # token 7: allowed by the trie, dropped by the filter
             prompt + [int(items[1][0]), 11, int(items[1][2]), int(items[1][3]), 2]]
    fn = prefix_allowed_tokens_fn(Trie(seqs))
    ref_fn = ref_whole_sentence_fn(RefTrie(seqs))
    closure = lambda b, s: fn(b, s)                                       # no .compile(): served by hostmask.py
    P = len(prompt)
    inputs = {"input_ids": torch.from_numpy(ci["prompt"])[None].cuda()}
    rt, rd = RefLlama(ci["target_dims"], ci["target_sd"]), RefLlama(ci["draft_dims"], ci["draft_sd"])
    bit = 0
    for K, DK in ((5, 10), (8, 11)):
        tgt, drf = _models(ci, K, DK)
        ref_tg = R.target_generate(rt, ci["prompt"], 4, K, ref_fn)
        bit += ref_tg["beam_sequence"].shape[0] < K
        for f in (fn, closure):
            tg = target_generate(tgt, inputs, 4, prefix_allowed_tokens_fn=f)
            nv = tg["n_valid"]
            assert nv == ref_tg["beam_sequence"].shape[0]
            assert tg["beam_sequence"][:nv, P:].cpu().tolist() == ref_tg["beam_sequence"][:, P:].tolist()
            np.testing.assert_allclose(tg["beam_scores"][:nv].cpu().numpy(), ref_tg["beam_scores"].numpy(), atol=SCORE_TOL, rtol=0)
            gen = tg["beam_sequence"][:nv, P:].cpu()
            assert bool(((gen >= 32000) | (gen == 2)).all())              # nothing the filter drops survives a one_step search
        ref = R.BSSD(rt, rd, ci["prompt"], 4, 4, K, DK, ref_fn)
        bit += any(n < DK for r in ref["rounds"] for n in r["step_len"][1:])
        out = BSSD(tgt, drf, inputs, 4, 4, prefix_allowed_tokens_fn=fn)
        _same_as_oracle(out, ref, P)
        assert out["accept_steps"] == [r["n_matches"] for r in ref["rounds"]]
        for r, g in zip(last_trace(tgt, drf), ref["rounds"]):
            for ids, gids in zip(r["draft_ids"], g["draft_ids"]):
                assert [x for x in ids if x >= 0] == gids
    assert bit >= 1, "the filter never dropped a pick: the case does not exercise beamSD.py:80-86"


def test_games_strict_trie_256_users_in_lock_step():
    """BASELINE config 3 at its stated batch: Games vocabulary (V = 33014), strict item trie, 256 users per lock-step batch
    (ATS_MAX_SEGS, the query-tile table and the per-user argument staging at their limits), small fp32 models so that four users
    can be checked against the oracle bit for bit."""
    vocab = synth.GAMES
    V = vocab.vocab_size
    tdims, ddims = synth.LlamaDims(V, 256, 2, 4, 704), synth.LlamaDims(V, 128, 2, 2, 352)
    tsd = synth.synthetic_state_dict(tdims, 21, std=0.05, head_std=0.3)
    dsd = synth.synthetic_state_dict(ddims, 22, std=0.05, head_std=0.3)
    kw = dict(max_slots=512, max_tokens=512, max_logit_rows=448)
    tgt = HipLlama.from_state_dict(tdims, tsd, torch.float32, num_beams=20, **kw)
    drf = HipLlama.from_state_dict(ddims, dsd, torch.float32, num_beams=40, **kw)
    items = synth.synthetic_items(vocab)
    item_set = {tuple(int(t) for t in it) for it in items}
    fn = SuffixTrieConstraint(Trie([[1] + [int(t) for t in it] + [2] for it in items]), synth.RESPONSE_SEP, 1)
    n_users = 256
    plens = synth.prompt_lengths(n_users, 2025, mean_hist=5.98)           # Games history shape
    prompts = [synth.synthetic_prompt(int(plens[u]), 7000 + u) for u in range(n_users)]
    inputs = [{"input_ids": torch.from_numpy(p)[None].cuda()} for p in prompts]
    bat = BSSD_batch(tgt, drf, inputs, 4, 4, prefix_allowed_tokens_fn=fn)
    assert len(bat) == n_users
    rt, rd = RefLlama(tdims, tsd), RefLlama(ddims, dsd)
    for u in (0, 85, 170, 255):
        ref = R.BSSD(rt, rd, prompts[u], 4, 4, 20, 40, fn)
        P = len(prompts[u])
        nv = bat[u]["n_valid"]
        assert bat[u]["beam_sequence"][:nv, P:].cpu().tolist() == ref["beam_sequence"][:, P:].tolist()[:nv]
        np.testing.assert_allclose(bat[u]["beam_scores"][:nv].cpu().numpy(), ref["beam_scores"].numpy()[:nv], atol=SCORE_TOL, rtol=0)
        assert (bat[u]["n_run"], bat[u]["accept_steps"]) == (ref["n_run"], [r["n_matches"] for r in ref["rounds"]])
    for u in range(n_users):
        P = len(prompts[u])
        toks = bat[u]["beam_sequence"][: bat[u]["n_valid"], P:].cpu().tolist()
        assert bat[u]["n_valid"] >= 1 and all(tuple(t) in item_set for t in toks)
        assert len({tuple(t) for t in toks}) == len(toks)
    release_decoders(tgt, drf)


def test_rccl_all_gather_of_counters_one_rank():
    """The path's single collective on the device backend (`nccl` = RCCL on ROCm), world size 1 on the one GPU of this box."""
    import torch.distributed as dist
    from atspeed_amd.dist import Counters, aggregate, all_gather_counters
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29653", rank=0, world_size=1)
    try:
        got = all_gather_counters(Counters(256, 768, 19, 1_250_000_000), torch.device("cuda", 0))
        assert [(c.n_users, c.n_run, c.accept_steps, c.elapsed_ns) for c in got] == [(256, 768, 19, 1_250_000_000)]
        agg = aggregate(got, 20)
        assert agg["items_per_s"] == pytest.approx(256 * 20 / 1.25) and agg["mean_accept_len"] == pytest.approx(19 / 768)
    finally:
        dist.destroy_process_group()


def test_decoder_cache_does_not_keep_models_alive():
    """a `--run_beam_sizes` sweep must not accumulate weights + KV arenas.  The decoder cache refers to models weakly
    and evicts their decoders when a model is collected."""
    case = CASES[0]
    ci = build_case_inputs(case)
    inputs = {"input_ids": torch.from_numpy(ci["prompt"])[None].cuda()}
    release_decoders()
    gc.collect(); torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    n0 = len(_Decoder._cache)
    for beam in (1, 5, 10, 20):                                            # the reference's style of sweep (inference.py:151)
        tgt, drf = _models(ci, beam, 40)
        BSSD_batch(tgt, drf, [inputs] * 3, 4, 4, prefix_allowed_tokens_fn=ci["fn"])
        assert len(_Decoder._cache) == n0 + 3
        del tgt, drf
        gc.collect()
        assert len(_Decoder._cache) == n0, "decoders of a collected model stayed in the cache"
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    assert free0 - torch.cuda.mem_get_info()[0] < 64 << 20, "device memory grew across the sweep"


def test_bf16_engine_without_packed_operands_equals_packed_one():
    """The packed operand layout is an address change only: a bf16 model whose dims do not allow it (ffn % 32 != 0) keeps HF's row-major layout
    and works; and for dims that allow both, the engine with ATSPEED_PACK=0 (row-major weights, `weight_layout = 0`) returns exactly what the
    packed engine returns -- one user's forwards (small-M kernels, split-K) and a lock-step batch (ring kernels)."""
    import os
    V = synth.TINY.vocab_size
    fn = atspeed_amd.PositionSetConstraint(synth.TINY.allowed_tokens(), synth.RESPONSE_SEP)
    kw = dict(dtype=torch.bfloat16, max_slots=512, max_tokens=512, max_logit_rows=448)
    odd = HipLlama.from_synthetic(synth.LlamaDims(V, 160, 2, 5, 208), 3, std=0.05, head_std=0.3, num_beams=8, **kw)
    assert not odd.weights_packed
    inp = {"input_ids": torch.from_numpy(synth.synthetic_prompt(40, 5))[None].cuda()}
    out = target_generate(odd, inp, 4, prefix_allowed_tokens_fn=fn)
    assert out["n_valid"] == 8 and bool(torch.isfinite(out["beam_scores"]).all())
    dims_t, dims_d = synth.LlamaDims(V, 512, 2, 4, 1408), synth.LlamaDims(V, 256, 2, 4, 704)
    prompts = [{"input_ids": torch.from_numpy(synth.synthetic_prompt(30 + 3 * u, 40 + u))[None].cuda()} for u in range(24)]
    res = {}
    old = os.environ.get("ATSPEED_PACK")
    try:
        for mode in ("1", "0"):
            os.environ["ATSPEED_PACK"] = mode
            t = HipLlama.from_synthetic(dims_t, 7, std=0.04, head_std=0.3, num_beams=20, **kw)
            d = HipLlama.from_synthetic(dims_d, 8, std=0.04, head_std=0.3, num_beams=40, **kw)
            assert t.weights_packed == (mode == "1") and d.weights_packed == (mode == "1")
            sd = t.export_state_dict()
            res[mode] = (BSSD(t, d, prompts[0], 4, 4, prefix_allowed_tokens_fn=fn), BSSD_batch(t, d, prompts, 4, 4, prefix_allowed_tokens_fn=fn),
                         sd["model.layers.1.mlp.up_proj.weight"], sd["lm_head.weight"])
            release_decoders(t, d)
    finally:
        if old is None:
            os.environ.pop("ATSPEED_PACK", None)
        else:
            os.environ["ATSPEED_PACK"] = old
    a, b = res["1"], res["0"]
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])                    # export undoes the packing
    assert torch.equal(a[0]["beam_sequence"], b[0]["beam_sequence"]) and torch.equal(a[0]["beam_scores"], b[0]["beam_scores"])
    for x, y in zip(a[1], b[1]):
        assert torch.equal(x["beam_sequence"], y["beam_sequence"]) and torch.equal(x["beam_scores"], y["beam_scores"])
        assert x["accept_steps"] == y["accept_steps"]


@pytest.mark.parametrize("n_seq,fp8", [(24, False), (56, False), (24, True), (56, True)])
def test_rope_and_kv_scatter_in_the_qkv_epilogue_equal_the_separate_pass(n_seq, fp8):
    """Batched bf16 forwards at head_dim 128 rotate q / k and scatter k / v to the caches inside the qkv projection's epilogue
    (gemm.hip: EPI_QKV_ROPE).  Same roundings as the projection + the separate RoPE pass, so logits are BIT-identical to the engine with
    ATSPEED_FUSE_QKV_ROPE=0 -- for a prefill (ragged lengths, a hidden slot) and for a second forward that attends to the cached K / V of
    the first (which checks what the epilogue wrote to the caches); 24 / 56 sequences: the 128- and the 256-row token tiles; fp8: the
    block-scaled kernel's epilogue (its own accumulator layout)."""
    import os
    V = 32000 + 256
    dims = synth.LlamaDims(V, 1024, 2, 8, 2816)
    m = HipLlama.from_synthetic(dims, 91, dtype=torch.bfloat16, max_slots=256, max_tokens=256, max_logit_rows=256)
    if fp8:
        m.enable_fp8()
    g = torch.Generator().manual_seed(11)
    first, second = [], []
    for i in range(n_seq):
        T = 100 if i < 2 else int(torch.randint(61, 100, (1,), generator=g))
        ids = torch.randint(3, V, (T,), generator=g).to(torch.int32)
        vis = torch.tril(torch.ones(T, T, dtype=torch.bool))
        vis[8:, 3] = False
        ar = torch.arange(T, dtype=torch.int32)
        first.append((ids, ar, ar.clone(), vis_bits_from_bool(vis, 256), T, 3))
        B = 70 + i % 7                                            # second forward: B new tokens after the T cached ones, tree-masked
        ids2 = torch.randint(3, V, (B,), generator=g).to(torch.int32)
        vis2 = torch.zeros(B, T + B, dtype=torch.bool)
        vis2[:, :T] = True
        vis2[:, 3] = False
        vis2[:, T:] = torch.tril(torch.ones(B, B, dtype=torch.bool))
        vis2[5:, T + 2] = False
        second.append((ids2, torch.arange(T, T + B, dtype=torch.int32), torch.arange(T, T + B, dtype=torch.int32), vis_bits_from_bool(vis2, 256), T + B, 5))
    res = {}
    for mode in ("1", "0"):
        with _lib.switches(fuse_qkv_rope=int(mode)):
            m.rope_fused_launches(reset=True)
            m.fp8_counters(reset=True)
            a = [o.clone() for o in m.forward_raw_batch(first)]
            b = [o.clone() for o in m.forward_raw_batch(second)]
            torch.cuda.synchronize()
            res[mode] = (a, b, m.rope_fused_launches())
            assert m.fp8_counters()["qkv"] == (dict(fp8=2 * dims.n_layers, other=0) if fp8 else dict(fp8=0, other=2 * dims.n_layers))
    assert res["1"][2] == 2 * dims.n_layers and res["0"][2] == 0, (res["1"][2], res["0"][2])
    for k in (0, 1):
        for x, y in zip(res["1"][k], res["0"][k]):
            assert bool(torch.isfinite(x).all()) and torch.equal(x, y), f"forward {k}: max |diff| {float((x - y).abs().max()):.3e} of max |logit| {float(y.abs().max()):.3f}"


@pytest.mark.parametrize("T,dtype", [(20, torch.bfloat16), (60, torch.bfloat16), (100, torch.bfloat16), (228, torch.bfloat16), (121, torch.float16)],
                         ids=["20", "60", "100", "228", "121_fp16"])
def test_one_user_fp8_rope_in_the_weight_streaming_qkv_epilogue_equals_the_separate_pass(T, dtype):
    """Round 6: ONE user's W8A8 qkv projection (gemm_wdma_kernel<..., EPI_QKV_ROPE, F8>: 192 unsplit tiles of 64 weight rows at hidden 4096)
    rotates q / k and scatters k / v to the caches in its epilogue -- one launch less per layer in the reference's batch-1 regime
    (code/inference.py:162-176).  Bit-identical logits to the `fuse_qkv_rope` switch off (ATSPEED_FUSE_QKV_ROPE=0: 16-bit store + rope_kv_segs_vec_kernel) for a first forward
    under a tree mask and for a second one that attends to the cached K / V the epilogue wrote; every token-tile height (32 / 64 / 128 / 256)."""
    import os
    V = 32000 + 256
    dims = synth.LlamaDims(V, 4096, 2, 32, 1024)
    m = HipLlama.from_synthetic(dims, 93, std=0.02, head_std=0.05, dtype=dtype, max_slots=512, max_tokens=512, max_logit_rows=448)
    m.enable_fp8()
    g = torch.Generator().manual_seed(12)
    ids = torch.randint(3, V, (T,), generator=g).to(torch.int32)
    vis = torch.tril(torch.ones(T, T, dtype=torch.bool))
    if T > 12:
        vis[8:, 3] = False
    ar = torch.arange(T, dtype=torch.int32)
    B = 40
    ids2 = torch.randint(3, V, (B,), generator=g).to(torch.int32)
    vis2 = torch.zeros(B, T + B, dtype=torch.bool)
    vis2[:, :T] = True
    vis2[:, 3] = False
    vis2[:, T:] = torch.tril(torch.ones(B, B, dtype=torch.bool))
    vis2[5:, T + 2] = False
    ar2 = torch.arange(T, T + B, dtype=torch.int32)
    res = {}
    for mode in ("1", "0"):
        with _lib.switches(fuse_qkv_rope=int(mode)):
            m.rope_fused_launches(reset=True)
            m.fp8_counters(reset=True)
            a = m.forward_raw(ids.cuda(), ar.cuda(), ar.clone().cuda(), vis_bits_from_bool(vis, 512).cuda(), T, min(T, 6)).clone()
            b = m.forward_raw(ids2.cuda(), ar2.cuda(), ar2.clone().cuda(), vis_bits_from_bool(vis2, 512).cuda(), T + B, 5).clone()
            torch.cuda.synchronize()
            res[mode] = (a, b, m.rope_fused_launches())
            assert m.fp8_counters()["qkv"] == dict(fp8=2 * dims.n_layers, other=0)
    assert res["1"][2] == 2 * dims.n_layers and res["0"][2] == 0, (res["1"][2], res["0"][2])
    for k in (0, 1):
        x, y = res["1"][k], res["0"][k]
        assert bool(torch.isfinite(x).all()) and torch.equal(x, y), f"forward {k}: max |diff| {float((x - y).abs().max()):.3e} of max |logit| {float(y.abs().max()):.3f}"


def test_fp8_rope_epilogue_of_the_weight_streaming_kernel_with_several_users_in_one_small_forward():
    """Three users' tokens in ONE forward of 177 rows (<= 256: the weight-streaming W8A8 kernels, not the ring): the fused RoPE epilogue finds every row's
    cache, slot and position through RowInfo, as the batched ring epilogue does.  Bit-identical to the separate pass, first forwards and second
    forwards over each user's own cached K / V."""
    V = 32000 + 256
    dims = synth.LlamaDims(V, 4096, 2, 32, 1024)
    m = HipLlama.from_synthetic(dims, 97, std=0.02, head_std=0.05, dtype=torch.bfloat16, max_slots=256, max_tokens=256, max_logit_rows=256)
    m.enable_fp8()
    g = torch.Generator().manual_seed(14)
    first, second = [], []
    for i, T in enumerate((70, 41, 66)):
        ids = torch.randint(3, V, (T,), generator=g).to(torch.int32)
        vis = torch.tril(torch.ones(T, T, dtype=torch.bool))
        vis[8:, 3 + i] = False
        ar = torch.arange(T, dtype=torch.int32)
        first.append((ids, ar, ar.clone(), vis_bits_from_bool(vis, 256), T, 3))
        B = 20 + i
        ids2 = torch.randint(3, V, (B,), generator=g).to(torch.int32)
        vis2 = torch.zeros(B, T + B, dtype=torch.bool)
        vis2[:, :T] = True
        vis2[:, T:] = torch.eye(B, dtype=torch.bool)
        second.append((ids2, torch.full((B,), T, dtype=torch.int32), torch.arange(T, T + B, dtype=torch.int32), vis_bits_from_bool(vis2, 256), T + B, B))
    res = {}
    for mode in (1, 0):
        with _lib.switches(fuse_qkv_rope=mode):
            m.rope_fused_launches(reset=True)
            a = [o.clone() for o in m.forward_raw_batch(first)]
            b = [o.clone() for o in m.forward_raw_batch(second)]
            torch.cuda.synchronize()
            res[mode] = (a, b, m.rope_fused_launches())
    assert res[1][2] == 2 * dims.n_layers and res[0][2] == 0
    for k in (0, 1):
        for x, y in zip(res[1][k], res[0][k]):
            assert bool(torch.isfinite(x).all()) and torch.equal(x, y), f"forward {k}: max |diff| {float((x - y).abs().max()):.3e}"


@pytest.mark.parametrize("dims,T,dtype", [(synth.LlamaDims(32256, 4096, 2, 32, 1024), 20, torch.bfloat16), (synth.LlamaDims(32256, 4096, 2, 32, 1024), 100, torch.bfloat16),
                                          (synth.LlamaDims(32256, 4096, 2, 32, 1024), 228, torch.float16), (synth.LlamaDims(32256, 768, 2, 12, 3072), 40, torch.bfloat16),
                                          (synth.LlamaDims(32256, 768, 2, 12, 3072), 7, torch.float16)], ids=["7b_20", "7b_100", "7b_228_fp16", "68m_40", "68m_7_fp16"])
def test_qkv_slabs_summed_by_the_rope_kernel_equal_the_reduce_launch(dims, T, dtype):
    """One user's 16-bit qkv projection leaves fp32 split-K slabs that RoPE + the KV scatter sum themselves (one launch less per layer).  Round 6 extends
    that to the LDS-tiled kernel's slabs (<= 32 tokens: the K-beam final step, beamSD.py:505-509; the draft's 40-token steps).  Same sums in the same
    order, same rounding: logits bit-identical to the `fuse_qkv_reduce` switch off, first forward and a second one over the cached K / V."""
    V = dims.vocab_size
    m = HipLlama.from_synthetic(dims, 95, std=0.02, head_std=0.05, dtype=dtype, max_slots=512, max_tokens=512, max_logit_rows=448)
    g = torch.Generator().manual_seed(13)
    ids = torch.randint(3, V, (T,), generator=g).to(torch.int32)
    vis = torch.tril(torch.ones(T, T, dtype=torch.bool))
    if T > 12:
        vis[8:, 3] = False
    ar = torch.arange(T, dtype=torch.int32)
    B = 20
    ids2 = torch.randint(3, V, (B,), generator=g).to(torch.int32)
    vis2 = torch.zeros(B, T + B, dtype=torch.bool)
    vis2[:, :T] = True
    vis2[:, T:] = torch.eye(B, dtype=torch.bool)
    ar2 = torch.full((B,), T, dtype=torch.int32)
    sl2 = torch.arange(T, T + B, dtype=torch.int32)
    res = {}
    cnt = (C.c_int64 * 16)()
    for mode in (1, 0):
        with _lib.switches(fuse_qkv_reduce=mode):
            _lib.load().atspeed_gemm_path_counters(cnt, 16, 1)
            a = m.forward_raw(ids.cuda(), ar.cuda(), ar.clone().cuda(), vis_bits_from_bool(vis, 512).cuda(), T, min(T, 6)).clone()
            b = m.forward_raw(ids2.cuda(), ar2.cuda(), sl2.cuda(), vis_bits_from_bool(vis2, 512).cuda(), T + B, B).clone()
            torch.cuda.synchronize()
            res[mode] = (a, b)
    for k in (0, 1):
        assert bool(torch.isfinite(res[1][k]).all()) and torch.equal(res[1][k], res[0][k]), f"forward {k}: max |diff| {float((res[1][k] - res[0][k]).abs().max()):.3e}"


def test_rccl_two_ranks(tmp_path):
    """BASELINE config 4's code path with more than one rank on REAL devices: two fresh child processes (one per GPU, started before anything in
    them touches a card), rank-local BSSD_batch on disjoint user shards, the path's single collective over RCCL; the union of the shards'
    results equals the one-rank run, every rank sees both ranks' counters.  Skipped on a one-GPU box (RCCL refuses two ranks on one device)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n_users, port = 5, 29671
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root)

    def run(world, tag):
        out = str(tmp_path / tag)
        procs = [subprocess.Popen([sys.executable, "-m", "tests.rccl_worker", str(r), str(world), str(port + world), str(n_users), out], cwd=root, env=env)
                 for r in range(world)]
        for p in procs:
            assert p.wait(timeout=600) == 0
        return [json.load(open(f"{out}.{r}.json")) for r in range(world)]

    one = run(1, "one")[0]
    # the worker's one-rank leg runs on any box: its results are the in-process engine's
    case = [c for c in CASES if c["name"] == "k5_dk10_indep"][0]
    ci = build_case_inputs(case)
    kw = dict(max_slots=512, max_tokens=512, max_logit_rows=448, device="cuda:0")
    tgt = HipLlama.from_state_dict(ci["target_dims"], ci["target_sd"], torch.float32, num_beams=case["K"], **kw)
    drf = HipLlama.from_state_dict(ci["draft_dims"], ci["draft_sd"], torch.float32, num_beams=case["DK"], **kw)
    prompts = [synth.synthetic_prompt(16 + u, 100 + u) for u in range(n_users)]
    here = BSSD_batch(tgt, drf, [{"input_ids": torch.from_numpy(p)[None].cuda()} for p in prompts], case["gamma"], case["max_new_tokens"],
                      prefix_allowed_tokens_fn=ci["fn"])
    assert one["tokens"] == [o["beam_sequence"][:, len(p):].cpu().tolist() for o, p in zip(here, prompts)]
    assert one["gathered"] == [[n_users, sum(one["n_run"]), sum(sum(a) for a in one["accept"]), 1_000_000]]
    release_decoders(tgt, drf)
    if int(_lib.load().atspeed_device_count()) < 2:
        pytest.skip("one-rank leg passed; the two-rank leg needs two HIP devices (RCCL refuses two ranks on one device)")
    two = run(2, "two")
    assert [u for r in two for u in r["users"]] == one["users"] == list(range(n_users))
    assert [t for r in two for t in r["tokens"]] == one["tokens"]
    assert [a for r in two for a in r["accept"]] == one["accept"]
    want = [(len(r["users"]), sum(r["n_run"]), sum(sum(a) for a in r["accept"]), 1_000_000 * (i + 1)) for i, r in enumerate(two)]
    for r in two:
        assert [tuple(g) for g in r["gathered"]] == want


def test_batch_calls_on_a_model_that_is_not_on_the_current_device():
    """every library call of BSSD_batch / target_generate_batch (the result assembly included) runs under the MODEL's device, so a
    model on cuda:1 works while cuda:0 is the thread's current device.  Needs two devices."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices")
    from atspeed_amd.beamSD import target_generate_batch
    case = [c for c in CASES if c["name"] == "k5_dk10_indep"][0]
    ci = build_case_inputs(case)
    outs = {}
    for d in (0, 1):
        dev = torch.device("cuda", d)
        kw = dict(max_slots=512, max_tokens=512, max_logit_rows=448, device=dev)
        tgt = HipLlama.from_state_dict(ci["target_dims"], ci["target_sd"], torch.float32, num_beams=case["K"], **kw)
        drf = HipLlama.from_state_dict(ci["draft_dims"], ci["draft_sd"], torch.float32, num_beams=case["DK"], **kw)
        torch.cuda.set_device(0)                                     # the current device stays 0 for both
        ins = [{"input_ids": torch.from_numpy(synth.synthetic_prompt(16 + u, 100 + u))[None].to(dev)} for u in range(3)]
        b = BSSD_batch(tgt, drf, ins, case["gamma"], case["max_new_tokens"], prefix_allowed_tokens_fn=ci["fn"])
        t = target_generate_batch(tgt, ins, case["max_new_tokens"], prefix_allowed_tokens_fn=ci["fn"])
        outs[d] = ([o["beam_sequence"].cpu().tolist() for o in b], [o["beam_sequence"].cpu().tolist() for o in t])
        assert all(o["beam_sequence"].device == dev for o in b + t)
        release_decoders(tgt, drf)
    assert outs[0] == outs[1]


def test_a_filtered_user_does_not_abort_the_lock_step_batch():
    """ATSPEED_ERR_FILTERED is per user in the batched calls.  A whole-sentence trie sends user B into a subtree of ordinary
    tokens (< 32000): the reference's post-top-k id filter (beamSD.py:80-86) drops every pick of B's first step (the reference then dies on a
    shape mismatch).  B ends with n_valid = 0 and status ERR_FILTERED, users A and C get exactly their one-user results; the one-user call for
    B still raises."""
    from atspeed_amd.beamSD import target_generate_batch
    case = [c for c in CASES if c["name"] == "k5_dk10_indep"][0]
    ci = build_case_inputs(case)
    kw = dict(max_slots=512, max_tokens=512, max_logit_rows=448, device="cuda:0")
    tgt = HipLlama.from_state_dict(ci["target_dims"], ci["target_sd"], torch.float32, num_beams=case["K"], **kw)
    drf = HipLlama.from_state_dict(ci["draft_dims"], ci["draft_sd"], torch.float32, num_beams=case["DK"], **kw)
    pa, pb, pc = [1, 50, 51, 52], [1, 60, 61], [1, 70, 71, 72, 73]
    items = [[32000 + i, 32064 + (3 * i) % 64, 32128 + (5 * i) % 64, 32192 + (7 * i) % 64, 2] for i in range(24)]
    seqs = [pa + it for it in items] + [pc + it for it in items[:12]] + [pb + [7 + i, 8, 9, 10, 2] for i in range(6)]
    fn = prefix_allowed_tokens_fn(Trie(seqs))
    ins = [{"input_ids": torch.tensor([p], dtype=torch.int64).cuda()} for p in (pa, pb, pc)]
    bat = BSSD_batch(tgt, drf, ins, 4, 4, prefix_allowed_tokens_fn=fn)
    tgb = target_generate_batch(tgt, ins, 4, prefix_allowed_tokens_fn=fn)
    # run twice: the second call's result buffers are the caching allocator's blocks of the first (a filtered user's block used to
    # keep another user's items and finite scores); the filtered user's block must read scores -inf / tokens 0 after the prompt
    bat = BSSD_batch(tgt, drf, ins, 4, 4, prefix_allowed_tokens_fn=fn)
    tgb = target_generate_batch(tgt, ins, 4, prefix_allowed_tokens_fn=fn)
    for res in (bat, tgb):
        assert [r["status"] for r in res] == [0, _lib.ERR_FILTERED, 0] and res[1]["n_valid"] == 0
        assert bool(torch.isneginf(res[1]["beam_scores"]).all()) and int(res[1]["beam_sequence"][:, len(pb):].abs().sum()) == 0
    for u in (0, 2):
        one = BSSD(tgt, drf, ins[u], 4, 4, prefix_allowed_tokens_fn=fn)
        nv = one["n_valid"]
        assert bat[u]["n_valid"] == nv >= 1 and torch.equal(bat[u]["beam_sequence"][:nv], one["beam_sequence"][:nv])
        assert torch.equal(tgb[u]["beam_sequence"][:nv], one["beam_sequence"][:nv])
    with pytest.raises(_lib.AtSpeedError) as ei:
        BSSD(tgt, drf, ins[1], 4, 4, prefix_allowed_tokens_fn=fn)
    assert ei.value.status == _lib.ERR_FILTERED
    release_decoders(tgt, drf)


def test_graph_replay_of_a_forward_that_takes_the_split_k_tail():
    """VERDICT r4 #7 / ADVICE r4: with the `graphs` switch on (ATSPEED_GRAPHS=1) a recurring forward of <= 512 tokens is captured and replayed as a hipGraph, and from 257
    tokens the N = 4096 projections take the ring kernel's split-K tail -- whose arena used to be allocated, and waited for across streams,
    inside the launch (illegal in a capture).  The arena now belongs to the model's activation context: the 300-token first verification of a
    180-token prompt is captured (second sight) and replayed (third), and every call returns the bits of the ungraphed engine."""
    import ctypes as C
    V = synth.BEAUTY.vocab_size
    kw = dict(max_slots=512, max_tokens=512, max_logit_rows=384, device="cuda:0")
    drf = HipLlama.from_synthetic(synth.llama_68m(V), 12, std=0.02, head_std=0.02, dtype=torch.bfloat16, num_beams=40, **kw)
    tgt = HipLlama.from_synthetic(synth.llama_7b(V, 2), 11, std=0.02, head_std=0.02, dtype=torch.bfloat16, num_beams=20, **kw)
    fn = PositionSetConstraint(synth.BEAUTY.allowed_tokens(), synth.RESPONSE_SEP)
    inp = {"input_ids": torch.from_numpy(synth.synthetic_prompt(180, 77))[None].cuda()}
    cnt = (C.c_int64 * 16)()
    lib = _lib.load()
    assert tgt.sk_arena_bytes() == 0 and drf.sk_arena_bytes() == 0     # no model owns a split-K arena before it runs a forward that can use one
    res = {}
    for mode in ("0", "1"):
        # (gemm_kcut=0: round 6's K-cut form would take o_proj at 300 tokens; this test is about the tail's arena inside a capture)
        with _lib.switches(graphs=int(mode), gemm_kcut=0):
            lib.atspeed_gemm_path_counters(cnt, 16, 1)
            res[mode] = [BSSD(tgt, drf, inp, 4, 4, prefix_allowed_tokens_fn=fn) for _ in range(4)]
            torch.cuda.synchronize()
            lib.atspeed_gemm_path_counters(cnt, 16, 0)
            res[mode + "sk"] = int(cnt[1])
    # ungraphed: every call's 300-token forward launches its o_proj tails (2 layers; down takes the panel form at 300 tokens); graphed: calls 1
    # and 2 (the capture) launch them, calls 3 and 4 replay the graph
    assert res["0sk"] >= 4 * 2 and 2 <= res["1sk"] < res["0sk"], (res["0sk"], res["1sk"])
    assert tgt.sk_arena_bytes() == 128 << 20 and drf.sk_arena_bytes() == 0   # the target's 300-token verification brought its arena; the draft (<= 180 tokens) has none
    for a in res["0"] + res["1"]:
        assert torch.equal(a["beam_sequence"], res["0"][0]["beam_sequence"]) and torch.equal(a["beam_scores"], res["0"][0]["beam_scores"])
        assert a["accept_steps"] == res["0"][0]["accept_steps"]
    release_decoders(tgt, drf)
