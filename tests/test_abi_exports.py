"""The drop-in boundary is the C ABI of include/atspeed_hip.h: the built library must export every function the header declares, the
ctypes table of atspeed_amd/_lib.py must bind exactly those, and the status / version calls answer without a GPU (no compute here)."""
import ctypes as C
import os
import re

from atspeed_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "atspeed_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)                      # comments mention functions too
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(re.findall(r"\b(atspeed_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_function_the_header_declares():
    names = _declared()
    assert len(names) >= 40, names
    lib = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/atspeed_hip.h but not exported by {_lib.LIB_PATH}: {missing}"


def test_ctypes_table_binds_only_declared_functions_and_loads():
    names = set(_declared())
    unknown = sorted(set(_lib.SIGNATURES) - names)
    assert not unknown, f"bound in atspeed_amd/_lib.py but not declared in the header: {unknown}"
    lib = _lib.load()                                                     # binds every signature: AttributeError if one is not exported
    assert lib.atspeed_last_error() is not None
    assert b"0.2" in lib.atspeed_version()                                # round 6: bumped (atspeed_gemm_fp8's signature changed in round 5 without a bump)


def test_switches_are_process_wide_integers_behind_the_abi():
    """Round 6: no getenv on the dispatch path -- the tuning / test switches are read once from their variables and changed only through
    atspeed_set_switch.  Host-only calls: defaults, set / get / restore, an unknown name is refused, the test hook has no variable."""
    lib = _lib.load()
    v = C.c_int32(-7)
    for name, dflt in ((b"gemm_sk", 1), (b"gemm_sk_g", 0), (b"gemm_panel", 1), (b"gemm_force_mt", 0), (b"graphs", 0), (b"fuse_qkv_rope", 1),
                       (b"fuse_qkv_reduce", 1), (b"gemm_kcut", 2)):
        env = {b"gemm_sk": "ATSPEED_GEMM_SK", b"gemm_panel": "ATSPEED_GEMM_PANEL", b"gemm_force_mt": "ATSPEED_GEMM_FORCE_MT", b"graphs": "ATSPEED_GRAPHS",
               b"fuse_qkv_rope": "ATSPEED_FUSE_QKV_ROPE", b"fuse_qkv_reduce": "ATSPEED_FUSE_QKV_REDUCE", b"gemm_kcut": "ATSPEED_GEMM_KCUT"}.get(name)
        assert lib.atspeed_get_switch(name, C.byref(v)) == 0
        if not (env and env in os.environ):
            assert v.value == dflt, (name, v.value)
    os.environ["ATSPEED_GEMM_SK_G"] = "17"                                # the unaligned-deal hook ignores the environment: a stray variable cannot reach it
    try:
        assert lib.atspeed_get_switch(b"gemm_sk_g", C.byref(v)) == 0 and v.value == 0
    finally:
        del os.environ["ATSPEED_GEMM_SK_G"]
    with _lib.switches(gemm_sk=3, gemm_kcut=0):
        assert lib.atspeed_get_switch(b"gemm_sk", C.byref(v)) == 0 and v.value == 3
        assert lib.atspeed_get_switch(b"gemm_kcut", C.byref(v)) == 0 and v.value == 0
    assert lib.atspeed_get_switch(b"gemm_sk", C.byref(v)) == 0 and v.value == int(os.environ.get("ATSPEED_GEMM_SK", 1))
    assert lib.atspeed_set_switch(b"no_such_switch", 1) == _lib.ERR_INVALID and b"no_such_switch" in lib.atspeed_last_error()
    assert lib.atspeed_get_switch(b"no_such_switch", C.byref(v)) == _lib.ERR_INVALID
