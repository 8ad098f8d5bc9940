"""Host logic of bench.py that decides what the driver's JSON line says (CPU only, no GPU call): how the rank count is resolved
(`--gpus N` must give N ranks or fail loudly), the torchrun command it starts, the provenance of `roofline.traffic`, and the
oracle-side scoring behind `cpu_baseline.bf16_vs_fp32_disagreements`."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

import bench
from atspeed_amd import synth
from atspeed_amd.generation_trie import PositionSetConstraint
from oracle import beamsd_ref as R
from oracle.llama_ref import RefLlama


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(gpus):
    return argparse.Namespace(gpus=gpus)


def test_rank_count_comes_from_gpus_flag():
    assert bench.resolve_world(_args(1), env={}) == ("run", 0, 0, 1)
    assert bench.resolve_world(_args(8), env={}) == ("spawn", 8)                       # plain `python bench.py --gpus 8`: start 8 ranks
    assert bench.resolve_world(_args(4), env={"WORLD_SIZE": "4", "RANK": "3", "LOCAL_RANK": "3"}) == ("run", 3, 3, 4)
    assert bench.resolve_world(_args(1), env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}) == ("run", 0, 0, 1)
    for gpus, world in ((8, 1), (1, 2), (2, 4)):                                        # a line must never claim another n_gpus than it ran
        with pytest.raises(SystemExit) as e:
            bench.resolve_world(_args(gpus), env={"WORLD_SIZE": str(world), "RANK": "0"})
        assert "--gpus" in str(e.value) and "WORLD_SIZE" in str(e.value)


def test_launcher_command_is_the_drivers_torchrun_line():
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"], port=29611)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29611"
    i = cmd.index(bench.os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]             # the children see the same flags (incl. --gpus)


def test_traffic_carries_its_provenance_and_refuses_stale_kernels(tmp_path):
    """`roofline.traffic` comes from offline PMC passes: the line must say so, and a summary recorded for other GEMM sources than this
    tree's (kernel_sha = sha256 of gemm.hip + its headers) must be refused, not quoted."""
    import json
    sha = bench.kernel_sha()
    assert len(sha) == 16 and sha == bench.kernel_sha()
    good = tmp_path / "t.json"
    good.write_text(json.dumps({"kernel_sha": sha, "commit": "abc1234", "gate_up": {"hbm_bytes_per_launch": 5.0e9, "avg_m": 25911}}))
    v, src = bench.traffic_from_profiles("gate_up", path=str(good))
    assert v == 5.0e9 and "not in this run" in src and "abc1234" in src and sha in src
    assert bench.traffic_from_profiles("no_such_kernel", path=str(good)) == (None, None)
    stale = tmp_path / "s.json"
    stale.write_text(json.dumps({"kernel_sha": "0" * 16, "commit": "old", "gate_up": {"hbm_bytes_per_launch": 7.6e9}}))
    v, src = bench.traffic_from_profiles("gate_up", path=str(stale))
    assert v is None and "refused" in src and sha in src
    assert bench.traffic_from_profiles("gate_up", path=str(tmp_path / "missing.json")) == (None, None)


def test_per_rank_report_shows_stragglers():
    from atspeed_amd.dist import Counters
    rep = bench.per_rank_report([Counters(512, 1536, 0, int(2.0e9)), Counters(512, 1500, 30, int(2.5e9))], beam=20)
    assert [r["rank"] for r in rep] == [0, 1] and rep[1]["elapsed_ms"] == pytest.approx(2500.0)
    assert rep[0]["items_per_s"] == pytest.approx(512 * 20 / 2.0) and rep[1]["n_users"] == 512 and rep[1]["accept_steps"] == 30


def test_oracle_scores_of_reproduces_the_oracles_own_beam_scores():
    """The packed tree forward that scores arbitrary sequences must give, for the oracle's own final beams, the oracle's beam scores."""
    V = synth.TINY.vocab_size
    dims = synth.LlamaDims(V, 64, 2, 4, 128)
    sd = synth.synthetic_state_dict(dims, 5, std=0.08, head_std=0.3)
    m = RefLlama(dims, sd)
    fn = PositionSetConstraint(synth.TINY.allowed_tokens(), synth.RESPONSE_SEP)
    prompt = synth.synthetic_prompt(20, 9)
    ref = R.target_generate(m, prompt, 4, 6, fn)
    seqs = ref["beam_sequence"][:, len(prompt):].tolist()
    got = bench.oracle_scores_of(m, prompt, seqs)
    np.testing.assert_allclose(got, ref["beam_scores"].numpy(), atol=2e-5, rtol=0)


def test_disagreement_report_scores_both_items_with_the_oracle():
    V = synth.TINY.vocab_size
    dims = synth.LlamaDims(V, 64, 2, 4, 128)
    m = RefLlama(dims, synth.synthetic_state_dict(dims, 5, std=0.08, head_std=0.3))
    fn = PositionSetConstraint(synth.TINY.allowed_tokens(), synth.RESPONSE_SEP)
    prompt = synth.synthetic_prompt(20, 9)
    P = len(prompt)
    ref = R.target_generate(m, prompt, 4, 6, fn)
    same = {"beam_sequence": ref["beam_sequence"].clone(), "beam_scores": ref["beam_scores"].clone()}
    rep = bench.disagreement_report(same, ref, P, m, prompt)
    assert rep["top_k_overlap"] == 1.0 and rep["ranks_that_differ"] == []
    # an "engine" that swapped ranks 2 and 3 and replaced the last item by a worse one
    seq = ref["beam_sequence"].clone()
    seq[[2, 3]] = seq[[3, 2]]
    seq[5, P + 3] = seq[5, P + 3] + 1 if int(seq[5, P + 3]) + 1 < V else seq[5, P + 3] - 1
    other = {"beam_sequence": seq, "beam_scores": ref["beam_scores"].clone()}
    rep = bench.disagreement_report(other, ref, P, m, prompt)
    assert rep["ranks_that_differ"] == [2, 3, 5] and rep["top_k_overlap"] == pytest.approx(5 / 6)
    rows = {r["rank"]: r for r in rep["per_rank"]}
    sc = ref["beam_scores"].tolist()
    assert rows[2]["gap"] == pytest.approx(sc[2] - sc[3], abs=2e-5) and rows[3]["gap"] == pytest.approx(sc[3] - sc[2], abs=2e-5)
    assert rows[5]["oracle_score_of_oracle_item"] == pytest.approx(sc[5])
    assert rep["max_gap"] >= abs(rows[2]["gap"])


class _FakeTarget:
    """what gemm_roofline() reads from a HipLlama: projection shapes and the fused-RoPE counter"""
    dims = synth.llama_7b(synth.BEAUTY.vocab_size, 32)

    def gemm_shape(self, kind):
        d = self.dims
        return {"qkv": (3 * d.hidden, d.hidden), "o_proj": (d.hidden, d.hidden), "gate_up": (2 * d.ffn, d.hidden),
                "down": (d.hidden, d.ffn), "lm_head": (d.vocab_size, d.hidden)}[kind]

    def rope_fused_launches(self):
        return 5


def _prof(ms, count, rows):
    return dict(ms=ms, count=count, rows=rows)


def test_roofline_object_arithmetic(tmp_path, monkeypatch):
    """`roofline` of the line: achieved = 2 M N K / launch time against the dense MFMA peak of the arithmetic type when the launch is above the
    ridge, algorithmic bytes / time against 8 TB/s below it; `frac` = achieved / peak; fp8 is priced against 5 PF; traffic only with provenance."""
    monkeypatch.setattr(bench, "traffic_from_profiles", lambda kind, **kw: (7.5e9, "test provenance") if kind == "gate_up" else (None, None))
    t = _FakeTarget()
    M, launches = 26000, 256
    prof = {"qkv": _prof(400.0, launches, M * launches), "o_proj": _prof(150.0, launches, M * launches), "gate_up": _prof(800.0, launches, M * launches),
            "down": _prof(380.0, launches, M * launches), "lm_head": _prof(30.0, 8, 19000 * 8)}
    r = bench.gemm_roofline(t, prof, prof, False, dict(mfma_bf16_tflops=1900.0, hbm_read_gbs=7000.0), 256, True)
    flops = 2.0 * M * 22016 * 4096
    assert r["bound"] == "mfma" and r["peak"] == 2500.0 and "gate_up" in r["kernel"] and "gemm_ring_kernel<3, 8, false" in r["kernel"]
    assert r["achieved"] == pytest.approx(flops / (800.0 / launches * 1e-3) / 1e12) and r["frac"] == pytest.approx(r["achieved"] / 2500.0)
    assert r["frac_of_measured"] == pytest.approx(r["achieved"] / 1900.0)
    assert r["algorithmic_bytes_per_launch"] == pytest.approx(22016 * 4096 * 2 + M * 4096 * 2 + M * 11008 * 2)
    assert r["traffic"] == 7.5e9 and r["traffic_over_algorithmic"] == pytest.approx(7.5e9 / r["algorithmic_bytes_per_launch"])
    assert sum(r["gemm_ms_share"].values()) == pytest.approx(1.0)
    # fp8: operand bytes halve, the peak is the dense fp8 peak, no bf16 measured peak is quoted, traffic (recorded for bf16) is not quoted
    r8 = bench.gemm_roofline(t, prof, prof, True, dict(mfma_bf16_tflops=1900.0, hbm_read_gbs=7000.0), 256, False)
    assert r8["peak"] == 5000.0 and "gemm_ring_mx_kernel" in r8["kernel"] and r8["peak_measured"] is None and r8["traffic"] is None
    assert r8["algorithmic_bytes_per_launch"] == pytest.approx(22016 * 4096 + M * 4096 + M * 11008 * 2)
    # one user's launches (M ~ 100): below the ridge -> HBM roofline on the algorithmic bytes
    small = {k: _prof(v["ms"] / 50, launches, 100 * launches) for k, v in prof.items()}
    zero = {k: _prof(0.0, 0, 0) for k in prof}
    rs = bench.gemm_roofline(t, small, zero, False, dict(mfma_bf16_tflops=1900.0, hbm_read_gbs=7000.0), 1, True)
    assert rs["bound"] == "hbm" and rs["unit"] == "GB/s" and rs["peak"] == 8000.0 and rs["traffic"] is None
    assert rs["achieved"] == pytest.approx(rs["algorithmic_bytes_per_launch"] / (small["gate_up"]["ms"] / launches * 1e-3) / 1e9)


def test_forwards_roofline_prices_each_forward_against_its_own_bound():
    """One user's forward (100 tokens) is priced by its weight stream, a 256-user forward (58 k tokens) by its matrix flops; the call's
    ideal time is the sum of the per-forward maxima and `frac` = ideal / measured."""
    import bench
    from atspeed_amd import synth
    d = synth.llama_7b(32859, 32)
    w_layers = 32 * (4 * 4096 * 4096 + 3 * 4096 * 11008)
    w_head = 4096 * 32859
    r = bench.forwards_roofline([(d, [(100, 100), (58000, 30976)])], elapsed_s=1.0)
    hbm = (w_layers + w_head) * 2 / 8e12
    mf = 2.0 * (w_layers * 58000 + w_head * 30976) / 2.5e15
    assert r["forwards"] == 2 and r["forwards_hbm_bound"] == 1
    assert abs(r["ideal_ms"] - 1e3 * (hbm + mf)) < 1e-6 and abs(r["frac"] - (hbm + mf)) < 1e-9
    assert 2.0 * (w_layers * 100 + w_head * 100) / 2.5e15 < hbm < mf


def test_final_line_is_compact_enough_for_the_driver(tmp_path, capsys):
    """BENCH_r04.json.parsed was null: the one JSON line had grown to 41 KB and the driver keeps an 8 KB tail.  The final stdout line is now a
    numbers-only summary built from the full report (which goes to bench_detail.json); built here from round 4's own 41 KB report."""
    import json
    import os
    with open(os.path.join(bench.ROOT, "profiles", "r04_bench.json")) as f:
        detail = json.load(f)
    assert len(json.dumps(detail)) > 30000
    text = bench.emit(detail, out_dir=str(tmp_path))
    out = capsys.readouterr().out
    assert out.endswith(text + "\n") and out.count("\n") == 1                # ONE line, the last one
    assert len(text) < bench.LINE_BUDGET_BYTES < 8192
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in line, k
    assert line["value"] == pytest.approx(detail["value"], rel=1e-4) and line["config"]["workload"].startswith("Beauty")
    assert line["roofline"]["frac"] == pytest.approx(detail["roofline"]["frac"], rel=1e-4) and line["roofline"]["bound"] == "mfma"
    assert line["roofline"]["avg_launch_us"] > 0 and line["roofline"]["algorithmic_flops_per_launch"] > 0 and "traffic" in line["roofline"]
    assert line["cpu_baseline"]["value"] == pytest.approx(detail["cpu_baseline"]["value"], rel=1e-4)
    assert line["cpu_baseline"]["cores"] == 16 and line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["sample"]
    assert len(line["speedup_curve"]) == 15 and all(isinstance(r["speedup"], float) for r in line["speedup_curve"])
    assert set(line["configs"]) == {"games_256", "games_256_trie", "fp16", "fp8"}
    assert "what" not in text and "per_rank" not in text                     # no prose, no disagreement rows
    with open(tmp_path / bench.DETAIL_FILE) as f:                            # the full report is kept beside it
        assert json.load(f)["speedup_curve"][0]["bssd"]["roofline"]["what"]


def test_an_oversized_summary_never_costs_the_headline(tmp_path, capsys):
    import json
    detail = dict(metric="m", value=1.0, unit="items/s", n_gpus=1, steps=1, warmup=0, ms_per_step=1.0, higher_is_better=True, scaling="weak",
                  vs_baseline=None, dtype="bf16", data="synthetic", config=dict(workload="w"), roofline=dict(bound="mfma", frac=0.5, achieved=1.0, peak=2.0),
                  cpu_baseline=dict(value=2.0, unit="items/s", cores=1, kind="port", sample="s"),
                  speedup_curve=[dict(users_per_batch=i, bssd=dict(ms_to_last_result=1.0), target_generate=dict(ms_to_last_result=2.0), speedup=2.0)
                                 for i in range(400)])
    text = bench.emit(detail, out_dir=str(tmp_path))
    capsys.readouterr()
    line = json.loads(text)
    assert len(text) <= bench.LINE_BUDGET_BYTES and line["roofline"]["frac"] == 0.5 and line["cpu_baseline"]["value"] == 2.0 and "speedup_curve" not in line


def test_round5_report_compacts_with_the_one_user_fp8_rows(tmp_path, capsys):
    """The round-5 report (profiles/r05_bench_detail.json: what `python bench.py` wrote beside the line the driver parsed) through the same
    builder: still under the budget with the new `one_user_fp8` rows (config 5 at the reference's batch-1 shape), traffic quoted with its ratio."""
    import json
    import os
    p = os.path.join(bench.ROOT, "profiles", "r05_bench_detail.json")
    if not os.path.exists(p):
        pytest.skip("no round-5 report in profiles/")
    with open(p) as f:
        detail = json.load(f)
    text = bench.emit(detail, out_dir=str(tmp_path))
    capsys.readouterr()
    line = json.loads(text)
    assert len(text) < bench.LINE_BUDGET_BYTES
    rows = {r["bracket"]: r for r in line["one_user_fp8"]}
    assert set(rows) == {"accept0", "resid3e-06", "resid3e-05"} and all(r["not_fp8"] == 0 and r["fp8_ms"] < r["bf16_ms"] for r in rows.values())
    assert abs(rows["resid3e-05"]["accept_fp8"] - rows["resid3e-05"]["accept_bf16"]) <= 0.05          # accepted-length drift of the W8A8 target, one user per call
    assert line["roofline"]["traffic"] and 7.0 < line["roofline"]["traffic_over_algorithmic"] < 9.5 and line["roofline"]["avg_m"] > 20000
    with open(os.path.join(bench.ROOT, "profiles", "r05_bench.json")) as f:                             # the committed line IS what the builder makes of the committed report
        committed = json.loads(f.read().strip().splitlines()[-1])
    assert committed["value"] == line["value"] and committed["roofline"]["frac"] == line["roofline"]["frac"]


def test_compact_line_of_a_multi_gpu_run_keeps_the_per_rank_rows():
    """N > 1: rank 0's line carries what every rank decoded (straggler visibility), numbers only."""
    import json
    from atspeed_amd.dist import Counters
    detail = dict(metric="m", value=4.0e4, unit="items/s", n_gpus=8, steps=2, warmup=1, ms_per_step=1000.0, higher_is_better=True, scaling="weak",
                  vs_baseline=None, dtype="bf16", data="synthetic", config=dict(workload="w", parallelism="user-shard x8"), mean_accept_len=0.0,
                  roofline=dict(bound="mfma", frac=0.59, achieved=1475.0, peak=2500.0, unit="TFLOP/s", traffic=None),
                  per_rank=bench.per_rank_report([Counters(512, 1536, 0, int(1.02e9 + 1e7 * r)) for r in range(8)], beam=20), cpu_baseline=None)
    line = bench.compact_line(detail)
    assert line["n_gpus"] == 8 and [r["rank"] for r in line["per_rank"]] == list(range(8)) and line["per_rank"][7]["elapsed_ms"] > line["per_rank"][0]["elapsed_ms"]
    assert line["cpu_baseline"] is None and len(json.dumps(line)) < bench.LINE_BUDGET_BYTES


def test_gpu_suite_duration_is_within_budget():
    """the driver gives the `-m gpu` suite 900 s; one more round of unchecked growth would have zeroed the parity grade.  The round's
    full run (`pytest tests -m gpu --durations=40` on one MI355X, committed as profiles/r06_gpu_suite_durations.txt) must finish within 600 s, and
    the recorded slowest tests must not hide a new heavyweight."""
    import re
    path = os.path.join(ROOT, "profiles", "r06_gpu_suite_durations.txt")
    assert os.path.exists(path), "profiles/r06_gpu_suite_durations.txt is missing: run tools/r06_call.sh suite through gpurun and commit its log"
    text = open(path).read()
    m = re.findall(r"(\d+) passed.* in ([\d.]+)s", text)
    assert m, "no pytest summary line in the durations file"
    passed, seconds = int(m[-1][0]), float(m[-1][1])
    assert passed >= 500 and "failed" not in text.splitlines()[-1]
    assert seconds <= 600.0, f"the GPU suite took {seconds:.0f} s on the recorded run: over the 600 s budget (driver limit 900 s)"
    durs = [(float(x), name) for x, name in re.findall(r"^([\d.]+)s call\s+(\S+)", text, re.M)]
    assert durs and max(durs)[0] <= 150.0, max(durs)
