"""Parity at the PRODUCTION sizes and dtype (BASELINE config 2: Beauty V=32859, Llama-68M draft / Llama-7B(32L) target,
K=20, DK=40, gamma=4, L=4) — the shapes bench.py's headline number is measured on.

  (a) fp32 engine at the full dims vs the oracle (`oracle.beamsd_ref.BSSD`, beamSD.py:458-542 restated) on the same
      device-generated weights: item token ids, per-round n_matches and the draft's candidate ids bit-exact, scores <= 1e-3
      (north star: "bit-exactly on accepted token indices and within 1e-3 on fp32 logits").
  (b) bf16 engine (`HipLlama.forward_raw` / `forward_raw_batch`: the small-M split-K kernels and the 256x256 ring GEMM +
      32-rows-per-wave attention of the lock-step batches) at hidden 4096 / ffn 11008 / head_dim 128 vs the oracle Llama
      (llama_ref.py, fp32) on exactly the bf16-rounded weights the device holds, with the bf16 tolerance stated below.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import atspeed_amd
from atspeed_amd import synth
from atspeed_amd.beamSD import BSSD, last_trace, target_generate
from atspeed_amd.model import HipLlama, vis_bits_from_bool
from oracle import beamsd_ref as R
from oracle.llama_ref import RefLlama

SCORE_TOL = 1e-3        # BASELINE.json north star
FP32_NOISE = 2e-5       # an fp32 engine that sums in another order: decisions closer than this are not comparable
# bf16 activations / KV (8 significant bits, one rounding per stored tensor) against fp32 on the same bf16 weights, relative to the
# largest |logit| of the forward: worst element and mean.  Observed on MI355X: see the printed numbers of the test.
BF16_MAX_TOL = 0.04
BF16_MEAN_TOL = 0.006


def _bench_models(dtype, layers=32, k=20, dk=40, **kw):
    """The pair bench.py builds (same seeds / std), in `dtype`."""
    V = synth.BEAUTY.vocab_size
    kw = dict(dict(max_slots=512, max_tokens=512, max_logit_rows=384), **kw)
    tgt = HipLlama.from_synthetic(synth.llama_7b(V, layers), 2025, std=0.02, head_std=0.02, dtype=dtype, num_beams=k, **kw)
    drf = HipLlama.from_synthetic(synth.llama_68m(V), 2026, std=0.02, head_std=0.02, dtype=dtype, num_beams=dk, **kw)
    return tgt, drf


# Why K = 20 cannot be asserted exactly and K = 1, 2 can (measured on the CPU oracle at these dims, tools/margin_search.py,
# profiles/r05_margin_search.txt).  A user makes ~10 top-k decisions; a decision sorts K (DK) winners + the first loser out of up to DK x 256
# (beam, token) candidates, i.e. K adjacent gaps that must all exceed the noise between two fp32 evaluations that sum in different orders.
# That noise is 5-7e-5 on a 4-token score (the oracle against ITSELF with another thread count; the HIP engine against the oracle: 8e-5-1.1e-4)
# and scales with the logits: damping the residual branches (resid_scale 0.1: noise 5e-5, smallest gap 1e-5-2e-4) and heavy-tailed heads
# (per-token log-normal row scales: gaps x 25, noise x 20) leave the ratio smallest gap / noise at 0.1-3 for K = 20 -- it is set by the number
# of gaps (~400 per user), not by the recipe.  With K = 1 or 2 (BASELINE config 1's greedy setting, DK = 2K) a user has 10-30 gaps and the
# smallest is 1e-3-6e-3 on all users tried: 20-100 x the noise, on the HEADLINE weights with all 32 layers at full strength.
@pytest.fixture(scope="module")
def fulldims_fp32():
    """The headline pair in fp32 and the oracle on exactly the weights the device holds, built ONCE for the three parametrisations below (27 GB of
    weights: building and exporting them is half of a test's time); a test sets its own beam counts on the models' generation_config."""
    from atspeed_amd.beamSD import release_decoders
    tgt, drf = _bench_models(torch.float32)
    rt = RefLlama(tgt.dims, tgt.export_state_dict(), max_slots=512)       # the oracle on exactly the weights the device holds
    rd = RefLlama(drf.dims, drf.export_state_dict(), max_slots=512)
    yield tgt, drf, rt, rd
    release_decoders(tgt, drf)


FP64_LEG_USERS = 2      # near-tied users judged by the fp64 arbiter (~20-40 s of CPU each at these dims); further ones keep the fp32 criterion
from tests.arbiter import fp64_gap as _fp64_gap_of, fp64_truth as _fp64_truth_of, oracle_scores_of as _oracle_scores_of      # noqa: E402


@pytest.mark.parametrize("K,DK", [(1, 2), (2, 4)])
def test_fp32_engine_at_full_dims_is_exact_where_exactness_is_decidable(fulldims_fp32, K, DK):
    """Full Llama-7B(32L) / Llama-68M dims, headline weights, K = 1 / 2 beams: item ids, per-round n_matches, accepted steps, the draft's
    candidate ids in order and the lossless property hold EXACTLY on all twelve (K = 1) / six (K = 2) users -- no near-tie branch (`near_ties == 0` asserted)."""
    n_users = 12 if K == 1 else 6                                   # (the 600 s budget of the GPU suite: K = 2 on six of the twelve prompts)
    checked, near_ties = _fulldims_against_oracle(fulldims_fp32, K, DK, strict=True, n_users=n_users)
    assert near_ties == 0 and checked == n_users


def test_fp32_engine_at_llama7b_llama68m_dims_equals_oracle(fulldims_fp32):
    """K = 20 / DK = 40 (the headline): exact where the engine and the fp32 oracle agree; where they differ (a user makes ~400 adjacent top-k gaps and
    the smallest is of the size of fp32 re-association noise, profiles/r05_margin_search.txt) BOTH are held to the fp64 arbiter: every rank of each
    list scores, in double precision, within 4 x FP32_NOISE of the fp64 search's own score at that rank (round 6, VERDICT r5 #4: the truth judges,
    not one fp32 evaluation the other)."""
    checked, near_ties = _fulldims_against_oracle(fulldims_fp32, 20, 40, strict=False)
    assert checked >= 8 and checked + near_ties == 12


def _fulldims_against_oracle(models, K, DK, strict, n_users=12):
    from atspeed_amd.beamSD import release_decoders
    tgt, drf, rt, rd = models
    release_decoders(tgt, drf)                                            # decoders (beam blocks) of another beam count
    tgt.generation_config.num_beams, drf.generation_config.num_beams = K, DK
    fn = atspeed_amd.PositionSetConstraint(synth.BEAUTY.allowed_tokens(), synth.RESPONSE_SEP)
    checked = near_ties = fp64_legs = 0
    max_fp64_gap = 0.0
    # twelve users (this engine is the judge of tests/test_decisions_gpu.py, so its own pin to the oracle must not be a
    # two-user link): the mean Beauty prompt, short ones, long ones; ~7 s of CPU oracle per user on the box's 16 allotted CPUs
    PROMPTS = (108, 70, 66, 84, 96, 78, 120, 150, 186, 72, 102, 132)[:n_users]
    for u, P in enumerate(PROMPTS):
        prompt = synth.synthetic_prompt(P, synth.tensor_seed(2025, f"user{u}"))
        inputs = {"input_ids": torch.from_numpy(prompt)[None].cuda()}
        R.MARGINS = []
        try:
            ref = R.BSSD(rt, rd, prompt, 4, 4, K, DK, fn)
        finally:
            margin, R.MARGINS = min(R.MARGINS), None
        out = BSSD(tgt, drf, inputs, 4, 4, prefix_allowed_tokens_fn=fn)
        tg = target_generate(tgt, inputs, 4, prefix_allowed_tokens_fn=fn)
        same = out["beam_sequence"][:, P:].cpu().tolist() == ref["beam_sequence"][:, P:].tolist()
        print(f"K={K} DK={DK} user {u}: P={P} n_run={out['n_run']} accept={out['total_accept_steps']} oracle decision margin={margin:.3e} "
              f"max score diff={float((out['beam_scores'].cpu() - ref['beam_scores']).abs().max()):.2e}")
        if strict:
            assert same, f"user {u}: item token ids differ from the oracle (decision margin {margin:.3e})"
        if same and not strict and u == 0:
            # the arbiter is exercised on every box (a near tie may occur on none of the twelve users): user 0's list, which equals the fp32 oracle's,
            # against the fp64 search -- the engine held to the truth directly
            t_items, t_sc = _fp64_truth(rt, rd, prompt, K, fn)
            gap = _fp64_gap(rt, rd, prompt, out["beam_sequence"][:, P:].cpu().tolist(), t_items, t_sc)
            max_fp64_gap = max(max_fp64_gap, gap)
            print(f"  user 0 against the fp64 search: same items {out['beam_sequence'][:, P:].cpu().tolist() == t_items}, largest fp64 score gap at a rank {gap:.2e}, "
                  f"engine's own scores off by {float(np.abs(out['beam_scores'].cpu().double().numpy() - np.asarray(t_sc)).max()):.2e}")
            assert gap < 4 * FP32_NOISE
            np.testing.assert_allclose(out["beam_scores"].cpu().double().numpy(), np.asarray(t_sc), atol=SCORE_TOL, rtol=0)
            fp64_legs += 1
        if not same:
            # no silent skip: a difference is only tolerated when the oracle's own smallest decision margin is below fp32 summation noise
            # AND every item the engine ranked differently is, by the oracle's own arithmetic, within that noise of the oracle's item there
            assert margin < FP32_NOISE, f"user {u}: item token ids differ from the oracle at full dims (decision margin {margin:.3e})"
            g_items, r_items = out["beam_sequence"][:, P:].cpu().tolist(), ref["beam_sequence"][:, P:].tolist()
            if fp64_legs < FP64_LEG_USERS:
                # the fp64 arbiter: the same beam search on the same weight values in double precision is the truth both fp32 evaluations approximate;
                # each list's items are scored in fp64 (one packed forward) and held, rank by rank, to the fp64 search's own scores
                t_items, t_sc = _fp64_truth(rt, rd, prompt, K, fn)
                for who, items in (("engine", g_items), ("fp32 oracle", r_items)):
                    gap = _fp64_gap(rt, rd, prompt, items, t_items, t_sc)
                    max_fp64_gap = max(max_fp64_gap, gap)
                    print(f"  user {u}: {who}'s list against the fp64 search: {sum(a == b for a, b in zip(items, t_items))} of {K} ranks hold the same item, "
                          f"largest fp64 score gap at a rank {gap:.2e}")
                    assert gap < 4 * FP32_NOISE, (u, who, gap)
                np.testing.assert_allclose(out["beam_scores"].cpu().double().numpy(), np.asarray(t_sc), atol=SCORE_TOL, rtol=0)
                fp64_legs += 1
            else:
                ranks = [i for i, (a, b) in enumerate(zip(g_items, r_items)) if a != b]
                sc = _oracle_scores_of(rt, prompt, [g_items[i] for i in ranks])
                for i, s_gpu in zip(ranks, sc):
                    assert abs(float(ref["beam_scores"][i]) - s_gpu) < 4 * FP32_NOISE, (u, i, float(ref["beam_scores"][i]), s_gpu)
            np.testing.assert_allclose(out["beam_scores"].cpu().numpy(), ref["beam_scores"].numpy(), atol=SCORE_TOL, rtol=0)
            near_ties += 1
            continue
        checked += 1
        assert same, "item token ids differ from the oracle at full dims"
        np.testing.assert_allclose(out["beam_scores"].cpu().numpy(), ref["beam_scores"].numpy(), atol=SCORE_TOL, rtol=0)
        assert (out["n_run"], out["total_accept_steps"]) == (ref["n_run"], ref["total_accept_steps"])
        tr = last_trace(tgt, drf)
        assert [r["n_matches"] for r in tr] == [r["n_matches"] for r in ref["rounds"]]
        for r, g in zip(tr, ref["rounds"]):
            for ids, gids in zip(r["draft_ids"], g["draft_ids"]):
                got = [x for x in ids if x >= 0]
                if margin >= FP32_NOISE or strict:
                    assert got == gids                                    # the draft's candidates, in order
                else:
                    # the oracle's own smallest decision margin is below fp32 summation noise: one pair of the draft's 40 candidates
                    # may swap places or trade its last seat (flat random-init logits); the user's items above are still the oracle's
                    assert len(got) == len(gids) and len(set(got) ^ set(gids)) <= 2 and sum(a != b for a, b in zip(got, gids)) <= 4, (u, got, gids)
        # lossless (beamSD.py:544-595): the plain beam search of the same engine gives the same items
        assert torch.equal(tg["beam_sequence"], out["beam_sequence"])
    print(f"K={K} DK={DK}: checked exactly {checked}, near ties {near_ties} of {len(PROMPTS)} users ({fp64_legs} judged by the fp64 arbiter, largest fp64 gap {max_fp64_gap:.2e})")
    assert checked + near_ties == len(PROMPTS)
    return checked, near_ties


def _fp64_truth(rt, rd, prompt, K, fn):
    return _fp64_truth_of(rt, prompt, K, fn)


def _fp64_gap(rt, rd, prompt, items, t_items, t_sc):
    return _fp64_gap_of(rt, prompt, items, t_items, t_sc)


def _tree_inputs(P, B, V, g, hide=5):
    """A prompt of P tokens then B tree tokens that see the prompt (minus one hidden slot) and themselves: a packed-verify-like forward."""
    ids = torch.cat((torch.randint(3, 32000, (P,), generator=g), torch.randint(32000, V, (B,), generator=g))).to(torch.int32)
    T = P + B
    vis = torch.zeros(T, T, dtype=torch.bool)
    vis[:P, :P] = torch.tril(torch.ones(P, P, dtype=torch.bool))
    vis[P:, :P] = True
    vis[P:, P:] = torch.eye(B, dtype=torch.bool)
    vis[P:, hide] = False
    pos = torch.cat((torch.arange(P), torch.full((B,), P))).to(torch.int32)
    return ids, pos, torch.arange(T, dtype=torch.int32), vis


def _bf16_err(got, want):
    scale = float(want.abs().max())
    err = (got - want).abs()
    return float(err.max()) / scale, float(err.mean()) / scale


def test_bf16_forward_at_llama7b_width_close_to_oracle_on_bf16_rounded_weights():
    V = synth.BEAUTY.vocab_size
    dims = synth.llama_7b(V, 3)                                            # hidden 4096, ffn 11008, 32 heads x 128; 3 layers
    m = HipLlama.from_synthetic(dims, 2025, std=0.02, head_std=0.02, dtype=torch.bfloat16, max_slots=512, max_tokens=512, max_logit_rows=384)
    ref = RefLlama(dims, m.export_state_dict(), max_slots=512)            # fp32 arithmetic on the device's bf16 weight values
    g = torch.Generator().manual_seed(11)
    # one sequence of 228 tokens (the first verification of one user: small-M kernels, split-K ring for the wide projections)
    ids, pos, slots, vis = _tree_inputs(108, 120, V, g)
    want = ref.forward(ids, pos, slots, vis, n_logit_rows=121)
    got = m.forward_raw(ids.cuda(), pos.cuda(), slots.cuda(), vis_bits_from_bool(vis, 512).cuda(), 228, 121).float().cpu()
    e_max, e_mean = _bf16_err(got, want)
    print(f"one sequence, 228 tokens: max err {e_max:.4f} mean err {e_mean:.5f} of max|logit| {float(want.abs().max()):.3f}")
    assert e_max < BF16_MAX_TOL and e_mean < BF16_MEAN_TOL
    # the decision the path takes from these rows: per row, the best allowed token of the first code range (91 columns)
    lo, hi = synth.BEAUTY.level_range(0)
    gap = want[:, lo:hi].topk(2, dim=1).values
    clear = (gap[:, 0] - gap[:, 1]) > 2 * BF16_MAX_TOL * float(want.abs().max())
    assert bool((got[:, lo:hi].argmax(1) == want[:, lo:hi].argmax(1))[clear].all())
    # a lock-step batch: 12 sequences x ~150 tokens = M > 1024 (256x256 ring GEMM, 128-row query tiles, LDS-DMA attention)
    seqs, refs = [], []
    for i in range(12):
        P, B = 70 + 6 * i, 60
        ids, pos, slots, vis = _tree_inputs(P, B, V, g, hide=3 + i)
        seqs.append((ids, pos, slots, vis_bits_from_bool(vis, 512), P + B, 8))
        refs.append((ids, pos, slots, vis))
    outs = m.forward_raw_batch(seqs)
    torch.cuda.synchronize()
    for i in (0, 5, 11):
        want = ref.forward(*refs[i], n_logit_rows=8)
        e_max, e_mean = _bf16_err(outs[i].float().cpu(), want)
        print(f"batched sequence {i}: max err {e_max:.4f} mean err {e_mean:.5f}")
        assert e_max < BF16_MAX_TOL and e_mean < BF16_MEAN_TOL
