"""Decision-level parity of the BENCHED path: the bf16 lock-step engine at the full Llama-7B(32L) / Llama-68M dims (BASELINE config 2:
Beauty V=32859, K=20, DK=40, gamma=4, L=4; the 256x256 ring GEMMs, the 32-rows-per-wave attention, the lm_head with the fused
full-vocabulary normaliser) against the fp32 engine on EXACTLY the same (bf16-valued) weights.

North star: "outputs match the reference bit-exactly on accepted token indices".  For a bf16 engine that can only mean: every decision
whose fp32 margin exceeds the engine's own noise is identical.  tests/replay.py states it precisely and this file asserts it:

  (i)   every top-K (target: beamSD.py:297-298,323-328; final step :505-509) and top-DK (draft: :76-78) membership that is CLEAR under the
        fp32 judge -- further from the decision boundary than 2 x the measured |bf16 - fp32| score error of the decision's items -- is the
        same in the bf16 engine; the acceptance tests (:371-380) are consistent with the traced sets;
  (ii)  for users whose decisions are ALL clear -- and for users where the bf16 engine made exactly the judge's choice at every decision,
        clear or not -- the free-running fp32 engine (itself bit-exact against the oracle at these dims, tests/test_fulldims_gpu.py)
        returns the same items, n_run and accept_steps (with flat random-init logits few users are all-clear: the second group is what
        ties the replay to a free decode);
  (iii) on the two aligned-weight brackets of bench.py the mean accepted length of the bf16 engine over the batch is >= the fp32
        engine's - 0.05 steps ("mean accepted length >= the reference's").
Two more cases run the same replay on BASELINE config 3's mask (Games vocabulary, strict item trie: the candidates of a decision are the
trie's children of each parent) and on config 5 (the target's projections in fp8: the judge stays the fp32 engine, so the quantisation
error is part of the measured noise the margins are held against).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import atspeed_amd
from atspeed_amd import synth
from atspeed_amd.beamSD import BSSD, BSSD_batch, last_decisions, release_decoders
from atspeed_amd.model import HipLlama
from tests.replay import TreeJudge, evaluate, judge_user

N_USERS = 64            # >= 32: every projection of every forward takes the ring kernel (M = 64 x 20..228 tokens), the lm_head EPI_F32_LSE
ACCEPT_EPS = 0.05       # (iii)
# non-vacuity floors, set from the measured values printed by the test (MI355X, seeds below): share of the fp32 judge's top-n memberships
# that are clear, and users whose every decision is clear
# measured: clear 0.177 / 0.842 / 0.951, identical 0.932 / 0.988 / 0.996, noise level of the deepest target step 0.63 / 0.03 / 0.009
MIN_CLEAR_SHARE = {None: 0.10, 3e-5: 0.70, 3e-6: 0.85, "games_trie": 0.10, "fp8_3e-5": 0.60, "peaked": 0.60, "fp16": 0.30, "fp8_fp16_3e-5": 0.60}
MIN_SAME_SHARE = {None: 0.88, 3e-5: 0.97, 3e-6: 0.99, "games_trie": 0.88, "fp8_3e-5": 0.96, "peaked": 0.97, "fp16": 0.97, "fp8_fp16_3e-5": 0.96}
MIN_SAME_USERS = {None: 0, 3e-5: 0, 3e-6: 16, "games_trie": 0, "fp8_3e-5": 0, "peaked": 0, "fp16": 0, "fp8_fp16_3e-5": 0}        # users whose every decision equals the judge's (measured 0 / 0 / 40+)
# "fp8_fp16_3e-5" (round 6): the reference's own dtype combination (code/inference.py:75-91: fp16 checkpoints, the target loaded 8-bit) -- the fp16 flavour with the
# target's projections in W8A8, judged by the fp32 engine on the fp16-valued weights; same floors as the bf16 + W8A8 case.
# "peaked": the HEADLINE regime made verifiable.  bench.py's headline weights are 32 unrelated random layers at full residual
# strength: a chaotic map whose bf16 error on a 4-token score (0.63) is of the size of the gaps between candidates, so only 18 % of its
# decisions can be judged (scaling the residual branches to a tenth and widening the head changes nothing: 18.8 % clear, noise 0.9-1.9 --
# the layers' outputs still dwarf the 0.02-wide embeddings).  A trained model is not chaotic: here the layers' residual contributions are
# scaled by 3e-4 (o_proj / down_proj; at 1e-3: 60 % clear, noise 0.19-0.39) and the head's rows are 3 x wider (logit std ~3.8), draft and target still UNRELATED (no shared table:
# acceptance stays ~0, four target forwards per user, every kernel, shape and launch of the headline) -- the gaps between candidates then stand
# clear of the engine's noise.
PEAKED = dict(resid_scale=3e-4, head_std=0.06)
# "fp16": the engine in the reference's own dtype (code/inference.py:75-100 loads fp16) on the HEADLINE recipe (unrelated weights at full
# residual strength), judged by the fp32 engine on exactly the fp16-valued weights.  Three more significand bits than bf16: the share of
# decisions that stand clear of the engine's noise must exceed the bf16 engine's on the same recipe (0.177, floor 0.10) -- VERDICT r3 #5.
BF16_CLEAR_SHARE_SAME_RECIPE = 0.177


def _pairs(resid_scale, V=synth.BEAUTY.vocab_size, align=True, head_std=0.02, half=torch.bfloat16):
    """bench.py's model pair in bf16 (or fp16) and, with the same weight VALUES, in fp32"""
    tdims, ddims = synth.llama_7b(V, 32), synth.llama_68m(V)
    kw = dict(max_slots=512, max_tokens=512, device=torch.device("cuda", 0))
    rs = 1.0 if resid_scale is None else resid_scale
    out = []
    for dtype, extra in ((half, dict(max_logit_rows=384)), (torch.float32, dict(max_logit_rows=448, round_to=half))):
        d = HipLlama.from_synthetic(ddims, 2026, std=0.02, head_std=head_std, dtype=dtype, num_beams=40, resid_scale=rs, **kw, **extra)
        t = HipLlama.from_synthetic(tdims, 2025, std=0.02, head_std=head_std, dtype=dtype, num_beams=20, resid_scale=rs,
                                    align_to=(d if (resid_scale is not None and align) else None), **kw, **extra)
        out.append((t, d))
    return out


@pytest.mark.parametrize("resid_scale", [None, 3e-5, 3e-6, "games_trie", "fp8_3e-5", "peaked", "fp16", "fp8_fp16_3e-5"],
                         ids=["unrelated_weights", "aligned_3e-5", "aligned_3e-6", "games_strict_trie", "fp8_target_aligned_3e-5", "peaked_unrelated_weights",
                              "fp16_engine_unrelated_weights", "fp16_engine_fp8_target_aligned_3e-5"])
def test_bf16_lockstep_decisions_equal_fp32_engine_where_margins_clear(resid_scale):
    case = resid_scale                                      # key of the floors above
    games = resid_scale == "games_trie"                     # BASELINE config 3's mask at the full dims: Games vocabulary, strict item trie
    fp8 = resid_scale in ("fp8_3e-5", "fp8_fp16_3e-5")      # BASELINE config 5: the target's batched projections in fp8 (W8A8 e4m3); the judge stays fp32:
    vocab = synth.GAMES if games else synth.BEAUTY          # the quantisation error is then part of the measured noise the margins are held against
    peaked = resid_scale == "peaked"                        # the headline regime (unrelated weights, ~0 acceptance) with clear margins
    f16 = resid_scale == "fp16"                             # the fp16 flavour of the engine on the headline recipe
    f16_fp8 = resid_scale == "fp8_fp16_3e-5"                # the fp16 flavour with the W8A8 target (the reference's combination)
    resid_scale = None if (games or f16) else (3e-5 if fp8 else resid_scale)
    (tb, db), (tf, df) = (_pairs(PEAKED["resid_scale"], vocab.vocab_size, align=False, head_std=PEAKED["head_std"]) if peaked
                          else _pairs(resid_scale, vocab.vocab_size, half=torch.float16 if (f16 or f16_fp8) else torch.bfloat16))
    assert tb.dtype == (torch.float16 if (f16 or f16_fp8) else torch.bfloat16)
    dev = tb.device
    if games:
        from atspeed_amd.generation_trie import SuffixTrieConstraint, Trie
        trie = Trie([[1] + [int(t) for t in it] + [2] for it in synth.synthetic_items(vocab)])
        fn = SuffixTrieConstraint(trie, synth.RESPONSE_SEP, 1)
        cache = {}

        def allowed_of(seq):                                # the trie's children of [bos] + generated suffix (generate_teacher_data.py:174-188)
            if seq not in cache:
                cache[seq] = torch.tensor(sorted(trie.get([1] + list(seq))), dtype=torch.long, device=dev)
            return cache[seq]
    else:
        fn = atspeed_amd.PositionSetConstraint(vocab.allowed_tokens(), synth.RESPONSE_SEP)
        by_depth = {d: torch.tensor(t, dtype=torch.long, device=dev) for d, t in vocab.allowed_tokens().items()}
        allowed_of = lambda seq: by_depth[len(seq)]
    plens = synth.prompt_lengths(N_USERS, 2025, mean_hist=5.98 if games else 7.33)
    prompts = [synth.synthetic_prompt(int(plens[u]), synth.tensor_seed(2025, f"user{u}")) for u in range(N_USERS)]
    inputs = [{"input_ids": torch.from_numpy(p)[None].to(dev)} for p in prompts]

    if fp8:
        tb.enable_fp8()
        tb.fp8_counters(reset=True)
    tb.profile(1)
    outs = BSSD_batch(tb, db, inputs, 4, 4, prefix_allowed_tokens_fn=fn, trace_decisions=True)
    torch.cuda.synchronize()
    big, allp = tb.profile_big(), tb.profile(0)
    # the path under test is the benched one: ring-kernel launches (>= 1024 tokens) for every projection and the lm_head
    for kind in ("qkv", "o_proj", "gate_up", "down", "lm_head"):
        assert big[kind]["count"] > 0 and 2 * big[kind]["count"] >= allp[kind]["count"], (kind, big[kind], allp[kind])
    assert tb.rope_fused_launches(reset=True) > 0
    if fp8:
        cnt = tb.fp8_counters()
        assert all(c["fp8"] > 0 and c["other"] == 0 for c in cnt.values()), cnt       # every layer projection of every forward ran in fp8

    jt, jd = TreeJudge(tf), TreeJudge(df)
    per_user, violations, all_clear_users, same_users = [], [], [], []
    for u in range(N_USERS):
        rounds = last_decisions(tb, db, lane=u)
        assert [r["n_matches"] for r in rounds if r["kind"] == "verify"] == outs[u]["accept_steps"]
        reports, accept_ok = judge_user(prompts[u], rounds, jt, jd, allowed_of)
        assert accept_ok, f"user {u}: traced acceptance inconsistent with the traced sets"
        per_user.append(reports)
    eps = evaluate([r for reports in per_user for r in reports])
    n_dec = n_in = n_clear = n_same = 0
    for u, reports in enumerate(per_user):
        for r in reports:
            n_dec += 1; n_in += r["n_in"]; n_clear += r["n_clear"]; n_same += r["n_same"]
            violations += [dict(user=u, what=r["what"], own_noise=r["own_noise"], **v) for v in r["violations"]]
        if all(r["all_clear"] for r in reports):
            all_clear_users.append(u)
        if all(r["n_same"] == r["n_in"] for r in reports):
            same_users.append(u)
    noises = [r["own_noise"] for reports in per_user for r in reports]
    print(f"[{case}] noise level per (model, depth):", {f"{k[0]}@{k[1]}": round(v, 4) for k, v in sorted(eps.items())})
    share = n_clear / max(1, n_in)
    print(f"[{case}] {N_USERS} users, {n_dec} decisions, {n_in} top-n memberships: {n_same} identical ({n_same / n_in:.3f}), "
          f"{n_clear} clear ({share:.3f}); bf16 score noise median {np.median(noises):.4f} max {max(noises):.4f}; "
          f"users with every decision clear: {len(all_clear_users)}, with every decision equal to the judge's: {len(same_users)}; "
          f"violations: {len(violations)}")
    assert not violations, violations[:5]                                                    # (i)
    assert share >= MIN_CLEAR_SHARE[case], f"only {share:.3f} of the memberships are clear: the assertion above would be vacuous"
    if f16:
        assert share > BF16_CLEAR_SHARE_SAME_RECIPE, (share, BF16_CLEAR_SHARE_SAME_RECIPE)
    assert n_same / n_in >= MIN_SAME_SHARE[case]
    assert len(same_users) >= MIN_SAME_USERS[case]

    # (ii) + (iii): the fp32 engine decoding freely
    f_acc = f_runs = b_acc = b_runs = 0
    for u in range(N_USERS):
        fo = BSSD(tf, df, inputs[u], 4, 4, prefix_allowed_tokens_fn=fn)
        f_acc += fo["total_accept_steps"]; f_runs += fo["n_run"]
        b_acc += outs[u]["total_accept_steps"]; b_runs += outs[u]["n_run"]
        if u in all_clear_users or u in same_users:
            P = len(prompts[u])
            assert sorted(map(tuple, outs[u]["beam_sequence"][:, P:].cpu().tolist())) == sorted(map(tuple, fo["beam_sequence"][:, P:].cpu().tolist())), u
            assert (outs[u]["n_run"], outs[u]["accept_steps"]) == (fo["n_run"], fo["accept_steps"]), u
    b_mean, f_mean = b_acc / max(1, b_runs), f_acc / max(1, f_runs)
    print(f"[{case}] mean accepted length: bf16 lock-step {b_mean:.4f} ({b_acc}/{b_runs}), fp32 engine {f_mean:.4f} ({f_acc}/{f_runs})")
    assert b_mean >= f_mean - ACCEPT_EPS                                                     # (iii)
    release_decoders(tb, db, tf, df)
