"""`Timer` of the drop-in surface (reference `code/beamSD.py:12-37`): context manager with `.time_cost`, decorator that injects
`result["time_cost"]` into the wrapped function's dict and accepts the `syn_device=` keyword the reference's call sites pass.  CPU only
(the device synchronisation is skipped when no HIP device is there, as `torch.cuda.synchronize` would raise)."""
import time

import pytest

from atspeed_amd.beamSD import Timer


def test_context_manager_measures_the_block():
    with Timer("stage") as t:
        time.sleep(0.02)
    assert t.func == "stage" and 0.015 < t.time_cost < 0.5


def test_decorator_injects_time_cost_and_accepts_syn_device():
    calls = []

    @Timer()
    def work(a, b=2):
        calls.append((a, b))
        time.sleep(0.01)
        return {"sum": a + b}

    out = work(1, b=5)
    assert out["sum"] == 6 and 0.005 < out["time_cost"] < 0.5 and calls == [(1, 5)]
    out = work(3, syn_device=0)                       # the reference's callers pass syn_device=...; it is not forwarded to the function
    assert out["sum"] == 5 and "time_cost" in out and calls[-1] == (3, 2)
    assert work.__name__ == "work"


def test_decorator_requires_a_dict_result_like_the_reference():
    @Timer()
    def bad():
        return 7

    with pytest.raises(TypeError):                     # the reference does result["time_cost"] = ... on whatever comes back
        bad()


def test_sync_flag_is_kept():
    t = Timer("x", sync_cuda=False, syn_device=3)
    assert t.sync_cuda is False and t.syn_device == 3
    with t:
        pass
    assert t.time_cost >= 0
