"""N > 1 path on CPU: user sharding + the single all-gather of counters, world_size 2, gloo."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from atspeed_amd.dist import Counters, aggregate, all_gather_counters, shard_range


def test_shard_range_partitions_users():
    for n in (0, 1, 7, 8, 9, 3553, 8696):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, init_file, n_users, q):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        # each rank decodes its own users with the oracle on tiny models (CPU stand-in for the GPU engine:
        # this test is about the sharding/collective logic, the engine's parity is tested on the GPU)
        from oracle import beamsd_ref as R
        from oracle.llama_ref import RefLlama
        from atspeed_amd import synth
        from tests.golden.cases import CASES, build_case_inputs
        case = [c for c in CASES if c["name"] == "k5_dk10_indep"][0]
        ci = build_case_inputs(case)
        tgt, drf = RefLlama(ci["target_dims"], ci["target_sd"]), RefLlama(ci["draft_dims"], ci["draft_sd"])
        lo, hi = shard_range(n_users, rank, world)
        c = Counters()
        for u in range(lo, hi):
            prompt = synth.synthetic_prompt(16 + u, 100 + u)
            c.add_result(R.BSSD(tgt, drf, prompt, case["gamma"], case["max_new_tokens"], case["K"], case["DK"], ci["fn"]))
        c.elapsed_ns = 1_000_000 * (rank + 1)
        allc = all_gather_counters(c)
        if rank == 0:
            q.put([(x.n_users, x.n_run, x.accept_steps, x.elapsed_ns) for x in allc])
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_all_gather_of_counters():
    world, n_users = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    with tempfile.TemporaryDirectory() as d:
        init = os.path.join(d, "init")
        procs = [ctx.Process(target=_worker, args=(r, world, init, n_users, q)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
    got = q.get()
    assert [g[0] for g in got] == [2, 1]                       # 3 users over 2 ranks
    per_rank = [Counters(*g) for g in got]
    agg = aggregate(per_rank, items_per_user=5)
    assert agg["users"] == 3
    assert agg["elapsed_s"] == pytest.approx(2e-3)             # max over ranks
    assert agg["items_per_s"] == pytest.approx(3 * 5 / 2e-3)
    assert sum(g[1] for g in got) >= 3                         # every user ran at least one verification round
    # single-process run over all users gives the same totals
    from oracle import beamsd_ref as R
    from oracle.llama_ref import RefLlama
    from atspeed_amd import synth
    from tests.golden.cases import CASES, build_case_inputs
    case = [c for c in CASES if c["name"] == "k5_dk10_indep"][0]
    ci = build_case_inputs(case)
    tgt, drf = RefLlama(ci["target_dims"], ci["target_sd"]), RefLlama(ci["draft_dims"], ci["draft_sd"])
    c = Counters()
    for u in range(n_users):
        c.add_result(R.BSSD(tgt, drf, synth.synthetic_prompt(16 + u, 100 + u), case["gamma"], case["max_new_tokens"],
                            case["K"], case["DK"], ci["fn"]))
    assert (c.n_run, c.accept_steps) == (sum(g[1] for g in got), sum(g[2] for g in got))


def test_all_gather_without_process_group_is_identity():
    c = Counters(2, 5, 1, 10)
    assert all_gather_counters(c) == [c]


def _metrics_worker(rank, world, init_file, q):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        from atspeed_amd.harness import InferenceResult, ItemIndex, reduce_metrics
        ix = ItemIndex({str(i): [f"<a_{i}>", f"<b_{i % 3}>", f"<c_{i % 5}>", f"<d_{i % 7}>"] for i in range(30)})
        # rank 0 holds 2 users, rank 1 holds 1: the whole-job metric weights ranks by their user counts
        if rank == 0:
            res = InferenceResult(predictions=[[1, 2, 3, 4], [5, 6, 7, 8]], labels=[[2], [9]], rows=[{}] * 2)
        else:
            res = InferenceResult(predictions=[[9, 1, 2, 3]], labels=[[9]], rows=[{}])
        m = reduce_metrics(res, ix, topN=(1, 4))
        if rank == 0:
            q.put(m)
    finally:
        dist.destroy_process_group()


def test_two_rank_metric_reduction_matches_single_process():
    from atspeed_amd.harness import InferenceResult, ItemIndex
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    with tempfile.TemporaryDirectory() as d:
        init = os.path.join(d, "init")
        procs = [ctx.Process(target=_metrics_worker, args=(r, world, init, q)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
    got = q.get()
    ix = ItemIndex({str(i): [f"<a_{i}>", f"<b_{i % 3}>", f"<c_{i % 5}>", f"<d_{i % 7}>"] for i in range(30)})
    whole = InferenceResult(predictions=[[1, 2, 3, 4], [5, 6, 7, 8], [9, 1, 2, 3]], labels=[[2], [9], [9]], rows=[{}] * 3).metrics(ix, (1, 4))
    assert got["users"] == 3 and got["topN"] == whole["topN"] == [1, 4]
    for key in ("precision", "recall", "ndcg", "mrr"):
        assert got[key] == pytest.approx(whole[key], abs=2e-4)      # per-rank values are rounded to 4 digits before the reduction
