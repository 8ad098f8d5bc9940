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
