"""every environment switch left on the dispatch path is a configuration of the shipped library that no test ran.  The ones read
once per process get a fresh child process each (`tests/env_switch_worker.py`): the same users through the bf16 engine (one user per call and a
24-user lock-step batch) and through the W8A8 target, compared with the default configuration -- rounds and accepted steps equal, scores within
the 16-bit noise of another summation order, item sets overlapping.  One process at a time on the card."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (families that cannot interact share one child process: the three epilogue / pass fusions; the three dispatch forms of the 257-1100-token band.
#  Since round 6 a variable only gives a switch's INITIAL value -- atspeed_set_switch changes it in-process -- which is what a fresh process tests.)
SWITCHES = [{"ATSPEED_GEMM_WDMA": "0"}, {"ATSPEED_FP8_SMALL": "0"}, {"ATSPEED_FP8_SMALL_BN64": "0"},
            {"ATSPEED_FUSE_QKV_REDUCE": "0", "ATSPEED_FUSE_LSE": "0", "ATSPEED_FUSE_QKV_ROPE": "0"},
            {"ATSPEED_ATTN32": "0", "ATSPEED_ATTN_RING": "0"}, {"ATSPEED_RMSNORM_PAIRS": "0", "ATSPEED_QUANT_PAIRS": "0"},
            {"ATSPEED_GEMM_SK": "0", "ATSPEED_GEMM_PANEL": "0", "ATSPEED_GEMM_KCUT": "0"}, {"ATSPEED_FP8_MX": "0"}, {"ATSPEED_GRAPHS": "1"}]


def _run(extra):
    env = dict(os.environ, PYTHONPATH=ROOT, **extra)
    r = subprocess.run([sys.executable, "-m", "tests.env_switch_worker"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (extra, r.stdout[-2000:], r.stderr[-2000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("ENVSWITCH ")][-1]
    return json.loads(line[len("ENVSWITCH "):])


@pytest.fixture(scope="module")
def default_run():
    return _run({})


def test_default_configuration_runs_every_projection_in_fp8(default_run):
    assert all(c["other"] == 0 and c["fp8"] > 0 for c in default_run["fp8_counters"].values()), default_run["fp8_counters"]
    # one user per call and the same user inside a lock-step batch: other kernels, same decisions on these weights
    for tag in ("bf16", "fp8"):
        for a, b in zip(default_run[tag + "_one"], default_run[tag + "_batch"][:3]):
            assert a["n_run"] == b["n_run"] and a["accept"] == b["accept"]


@pytest.mark.parametrize("switch", SWITCHES, ids=[",".join(f"{k}={v}" for k, v in s.items()) for s in SWITCHES])
def test_switched_configuration_decodes_like_the_default(default_run, switch):
    got = _run(switch)
    if switch.get("ATSPEED_FP8_SMALL") == "0":                      # one user's forwards stay 16-bit: the counters must say so
        assert any(c["other"] > 0 for c in got["fp8_counters"].values())
    for key in ("bf16_one", "bf16_batch", "fp8_one", "fp8_batch"):
        same_rounds = sum(1 for a, b in zip(got[key], default_run[key]) if (a["n_run"], a["accept"]) == (b["n_run"], b["accept"]))
        assert same_rounds >= len(got[key]) - max(1, len(got[key]) // 8), (key, same_rounds)
        for a, b in zip(got[key], default_run[key]):
            overlap = len({tuple(x) for x in a["items"]} & {tuple(x) for x in b["items"]}) / 20.0
            assert overlap >= 0.7, (key, overlap)
            assert abs(a["scores"][0] - b["scores"][0]) <= (0.5 if key.startswith("fp8") else 0.25), (key, a["scores"][0], b["scores"][0])
