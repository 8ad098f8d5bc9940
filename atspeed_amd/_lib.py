"""ctypes binding of libatspeed_hip.so (C-ABI: include/atspeed_hip.h).

There is no CPU fallback: if the shared library is missing, every HIP-backed entry point
raises ImportError telling the user to build it (`python -c "import __graft_entry__ as g; g.build()"`).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# ATSPEED_LIB: another build of the library (tuning builds under tools/probe); it must exist, there is no fallback
LIB_PATH = os.path.abspath(os.environ["ATSPEED_LIB"]) if os.environ.get("ATSPEED_LIB") else os.path.join(_HERE, "lib", "libatspeed_hip.so")

ATSPEED_F32, ATSPEED_BF16, ATSPEED_F16 = 0, 1, 2
WEIGHTS_ROW_MAJOR, WEIGHTS_PACKED = 0, 1
MAX_BEAMS, MAX_NEW_TOKENS, MAX_GAMMA = 64, 16, 8
ERR_INVALID, ERR_HIP, ERR_CAPACITY, ERR_CONSTRAINT, ERR_NO_DEVICE, ERR_FILTERED = -1, -2, -3, -4, -5, -6
EPI_STORE, EPI_F32, EPI_RESID, EPI_SWIGLU = 0, 1, 2, 3


class LlamaLayerWeights(C.Structure):
    _fields_ = [("input_norm", C.c_void_p), ("wqkv", C.c_void_p), ("wo", C.c_void_p),
                ("post_norm", C.c_void_p), ("wgu", C.c_void_p), ("wd", C.c_void_p)]


class LlamaConfig(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("hidden", C.c_int32), ("n_layers", C.c_int32), ("n_heads", C.c_int32),
                ("ffn", C.c_int32), ("rope_theta", C.c_float), ("rms_eps", C.c_float), ("dtype", C.c_int32),
                ("max_slots", C.c_int32), ("max_tokens", C.c_int32), ("max_logit_rows", C.c_int32), ("weight_layout", C.c_int32)]


class GenStats(C.Structure):
    _fields_ = [("n_run", C.c_int32), ("total_accept_steps", C.c_int32), ("accept_steps", C.c_int32 * MAX_NEW_TOKENS),
                ("n_valid", C.c_int32), ("n_target_forwards", C.c_int32), ("n_draft_forwards", C.c_int32),
                ("draft_ms", C.c_float), ("target_ms", C.c_float), ("verify_ms", C.c_float), ("total_ms", C.c_float),
                ("status", C.c_int32)]


# every symbol include/atspeed_hip.h declares: (restype, argtypes)
_P, _I, _F, _SZ, _U32, _U64 = C.c_void_p, C.c_int32, C.c_float, C.c_size_t, C.c_uint32, C.c_uint64
SIGNATURES = {
    "atspeed_version": (C.c_char_p, []),
    "atspeed_last_error": (C.c_char_p, []),
    "atspeed_device_count": (C.c_int, []),
    "atspeed_fill_hash_normal": (C.c_int, [_P, _SZ, _U32, _F, _F, C.c_int, _U64, _P]),
    "atspeed_fsm_create": (C.c_int, [_P, _P, _P, _I, _I, _I, C.POINTER(_P)]),
    "atspeed_fsm_destroy": (None, [_P]),
    "atspeed_fsm_create_free": (C.c_int, [_I, C.POINTER(_P)]),
    "atspeed_fsm_set_id_filter": (C.c_int, [_P, _I, _I]),
    "atspeed_row_topk": (C.c_int, [_P, _I, _I, _I, _I, _P, _P]),
    "atspeed_beam_expand_prune_free": (C.c_int, [_P, _I, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "atspeed_trie_flatten": (C.c_int, [_P, _P, _I, _P, _P, _P, C.POINTER(_I), C.POINTER(_I)]),
    "atspeed_llama_create": (C.c_int, [C.POINTER(LlamaConfig), _P, _P, _P, C.POINTER(LlamaLayerWeights), C.POINTER(_P)]),
    "atspeed_llama_destroy": (None, [_P]),
    "atspeed_llama_forward": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "atspeed_llama_forward_batch": (C.c_int, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "atspeed_llama_logits": (_P, [_P]),
    "atspeed_probe_mfma_bf16": (C.c_int, [_I, _P, C.c_size_t, _P, _P]),
    "atspeed_probe_hbm_read": (C.c_int, [_P, C.c_size_t, _I, _P, _P, _P]),
    "atspeed_llama_enable_fp8": (C.c_int, [_P, _P]),
    "atspeed_llama_fp8_counters": (C.c_int, [_P, _P, _P, _I]),
    "atspeed_llama_rope_fused_launches": (C.c_int64, [_P, _I]),
    "atspeed_llama_sk_arena_bytes": (C.c_int64, [_P]),
    "atspeed_quant_rows_fp8": (C.c_int, [_P, _I, _I, _P, _P, _P]),
    "atspeed_quant_rows_fp8_packed": (C.c_int, [_P, _I, _I, _P, _P, _P]),
    "atspeed_gemm_fp8": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "atspeed_llama_profile": (C.c_int, [_P, _I, _P, _P, _P]),
    "atspeed_llama_profile_big": (C.c_int, [_P, _P, _P, _P]),
    "atspeed_llama_forward_log": (C.c_int32, [_P, _I, _P, _I]),
    "atspeed_llama_logits_ld": (_I, [_P]),
    "atspeed_lse_rows": (C.c_int, [_P, _I, _I, _I, _P, _P]),
    "atspeed_log_softmax_rows": (C.c_int, [_P, _I, _P, _I, _I, _P, _I, _P]),
    "atspeed_lmhead_lse": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _SZ, C.POINTER(_I), _P]),
    "atspeed_beam_expand_prune": (C.c_int, [_P, _I, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P]),
    "atspeed_accept": (C.c_int, [_P, _P, _I, _P, _I, _P, _P, _P, _P]),
    "atspeed_decoder_create": (C.c_int, [_P, _P, _I, C.POINTER(_P)]),
    "atspeed_decoder_destroy": (None, [_P]),
    "atspeed_decoder_set_sampling": (C.c_int, [_P, _I, _F, C.c_uint32]),
    "atspeed_bssd_generate": (C.c_int, [_P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _P, C.POINTER(GenStats), _P]),
    "atspeed_bssd_generate_batch": (C.c_int, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "atspeed_target_generate": (C.c_int, [_P, _P, _I, _P, _I, _I, _I, _P, _P, C.POINTER(GenStats), _P]),
    "atspeed_target_generate_batch": (C.c_int, [_P, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "atspeed_assemble_sequences": (C.c_int, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "atspeed_decoder_trace": (C.c_int, [_P, _P, _I]),
    "atspeed_decoder_set_trace": (C.c_int, [_P, _I]),
    "atspeed_decoder_decisions": (C.c_int64, [_P, _P, C.c_int64]),
    "atspeed_gemm": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "atspeed_gemm_path_counters": (C.c_int, [_P, _I, _I]),
    "atspeed_set_switch": (C.c_int, [C.c_char_p, _I]),
    "atspeed_get_switch": (C.c_int, [C.c_char_p, C.POINTER(_I)]),
    "atspeed_gemm_packed": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "atspeed_gemm_fp8_packed": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "atspeed_pack_rows": (C.c_int, [_P, _P, _I, _I, _P]),
    "atspeed_unpack_rows": (C.c_int, [_P, _P, _I, _I, _P]),
    "atspeed_rmsnorm": (C.c_int, [_P, _P, _P, _I, _I, _F, _I, _P]),
    "atspeed_rmsnorm_quant_fp8": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "atspeed_tree_attention": (C.c_int, [_P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "atspeed_tree_attention_tiled": (C.c_int, [_P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
}

_lib: Optional[C.CDLL] = None


class AtSpeedError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"[atspeed_hip status {status}] {message}")
        self.status = status
        self.message = message


def load() -> C.CDLL:
    """Load the HIP library (after torch, so both share one libamdhip64)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `make -C atspeed_amd/csrc` (or `__graft_entry__.build()`); atspeed_amd has no CPU fallback.")
    try:
        import torch  # noqa: F401  -- pulls in torch's libamdhip64.so.7 first so the SONAME resolves to it
    except Exception:  # pragma: no cover - torch is plumbing; the library also loads against /opt/rocm
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int) -> None:
    if status == 0:
        return
    msg = load().atspeed_last_error().decode("utf-8", "replace")
    if status == ERR_CONSTRAINT:
        raise ValueError(msg)          # what HF's PrefixConstrainedLogitsProcessor raises
    raise AtSpeedError(status, msg)


class switches:
    """`with _lib.switches(gemm_sk=2, gemm_panel=0): ...` -- set process-wide library switches (atspeed_set_switch) and restore them on exit.
    The environment variables (ATSPEED_GEMM_SK, ...) only give the INITIAL values, read once by the library; tests and sweeps that compare
    two settings in one process go through here."""

    def __init__(self, **values: int):
        self.values = values
        self.old = {}

    def __enter__(self):
        lib = load()
        for name, v in self.values.items():
            cur = _I(0)
            check(lib.atspeed_get_switch(name.encode(), C.byref(cur)))
            self.old[name] = cur.value
            check(lib.atspeed_set_switch(name.encode(), int(v)))
        return self

    def __exit__(self, *exc):
        lib = load()
        for name, v in self.old.items():
            lib.atspeed_set_switch(name.encode(), v)
        return False


def dtype_code(dtype) -> int:
    """torch dtype -> the library's atspeed_dtype code (fp32 parity mode, bf16, fp16)."""
    import torch
    try:
        return {torch.float32: ATSPEED_F32, torch.bfloat16: ATSPEED_BF16, torch.float16: ATSPEED_F16}[dtype]
    except KeyError:
        raise TypeError(f"libatspeed_hip computes in torch.float32, torch.bfloat16 or torch.float16, not {dtype}") from None


def stream_ptr(device=None) -> int:
    import torch
    return int(torch.cuda.current_stream(device).cuda_stream)
