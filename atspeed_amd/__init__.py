"""atspeed_amd — MI355X-native beam speculative decoding (the AtSpeed inference hot path).

Public surface mirrors the reference's (`code/beamSD.py`, `code/generation_trie.py`):
`BSSD` (alias `beam_sd_generate`), `target_generate`, `Trie`, `prefix_allowed_tokens_fn`, `Timer`.
Everything that computes runs in the HIP library `atspeed_amd/lib/libatspeed_hip.so`
(C-ABI in `include/atspeed_hip.h`); there is no CPU fallback.
"""
from .generation_trie import (ConstraintFSM, PositionSetConstraint, SuffixTrieConstraint, Trie,
                              WholeSentenceTrieConstraint, prefix_allowed_tokens_fn)

__all__ = ["Trie", "prefix_allowed_tokens_fn", "PositionSetConstraint", "SuffixTrieConstraint",
           "WholeSentenceTrieConstraint", "ConstraintFSM", "BSSD", "BSSD_batch", "beam_sd_generate", "target_generate", "target_generate_batch",
           "Timer", "HipLlama"]


def __getattr__(name):
    # the HIP-backed entry points import torch + the shared library lazily
    if name in ("BSSD", "BSSD_batch", "beam_sd_generate", "target_generate", "target_generate_batch", "Timer", "one_step_beam_search"):
        from . import beamSD
        return getattr(beamSD, name)
    if name in ("HipLlama",):
        from . import model
        return getattr(model, name)
    raise AttributeError(name)
