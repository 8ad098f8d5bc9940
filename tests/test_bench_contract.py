"""The committed bench line (profiles/r01_final_bench.json = the last `python bench.py` of the round on an MI355X) carries every field of
the driver's contract, with consistent arithmetic.  CPU-only: guards the JSON shape, not the numbers."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r01_final_bench.json")) as f:
        return json.load(f)


def test_contract_fields_present():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "items/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and "synthetic" in d["data"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["n_gpus"] == 1 and d["steps"] >= 1


def test_value_is_users_times_beams_over_time():
    d = _line()
    users = d["config"]["users_per_step"] * d["steps"]
    items = users * 20                                    # K = 20 beams per user (BASELINE.json)
    assert abs(d["value"] - items / (d["ms_per_step"] * d["steps"] * 1e-3)) < 1e-6 * d["value"] + 1e-3


def test_roofline_and_cpu_baseline_objects():
    d = _line()
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["peak"] == (2500.0 if r["bound"] == "mfma" else 8000.0)       # nominal peaks of /opt/skills/guides/MI355X_MICROARCH.md
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    s = d["verify_scan"]
    assert s["bound"] == "hbm" and abs(s["frac"] - s["achieved"] / s["peak"]) < 1e-9
