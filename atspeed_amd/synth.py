"""Deterministic synthetic inputs for the beam-speculative-decoding path.

There is no tokenizer, dataset or checkpoint offline, so every test, fixture and
benchmark draws its weights / prompts / item index from one counter-based hash
PRNG that is bit-reproducible in numpy (here) and on the device
(`csrc/fill.hip`, same integer recipe).  Nothing here is on the product path's
compute: it only manufactures inputs.

Vocabulary shape follows the reference's data (facts, not copied data):
  * Llama vocab 32000 + item code tokens appended in sorted-string order
    (reference `code/data.py:46-57`, `code/finetune_llama.py:83-85`);
  * Beauty: 91/256/256/256 distinct codes at the 4 positions -> V = 32859,
    Games: 248/256/254/256 -> V = 33014 (from `data/*/*.LCRec-1e-3lr.json`);
  * item tokens are >= 32000 and EOS is 2 (hard-coded at `code/beamSD.py:81`).
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import numpy as np

LLAMA_VOCAB = 32000  # reference code/beamSD.py:81
EOS_ID = 2           # reference code/beamSD.py:81, code/inference.py:107
BOS_ID = 1           # reference code/inference.py:106

_M32 = np.uint64(0xFFFFFFFF)


def _fmix32(h: np.ndarray) -> np.ndarray:
    """murmur3 finaliser on uint32 lanes held in uint64 (so products don't overflow)."""
    h = h & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


def hash_u32(idx: np.ndarray, seed: int) -> np.ndarray:
    """uint32 hash of (idx, seed); idx is an integer array < 2**32."""
    x = (idx.astype(np.uint64) * np.uint64(0x9E3779B1) + np.uint64(seed & 0xFFFFFFFF)) & _M32
    return _fmix32(x)


def tensor_seed(base_seed: int, name: str) -> int:
    """Per-tensor stream seed: crc32(name) mixed with the base seed."""
    c = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    return int(_fmix32(np.array([(c ^ ((base_seed * 0x632BE5AB) & 0xFFFFFFFF))], dtype=np.uint64))[0])


_IH_STD = 65536.0 / np.sqrt(3.0)  # std of the sum of four U{0..65535}


def normal_scale(std: float) -> np.float32:
    """fp32 multiplier used by both the numpy and the device generator."""
    return np.float32(std / _IH_STD)


def hash_normal(n: int, seed: int, std: float, offset: int = 0, chunk: int = 1 << 24) -> np.ndarray:
    """~N(0, std^2) fp32 values: Irwin-Hall(4) of 16-bit hash pieces.

    value(i) = float32(int(s_i) - 131070) * float32(std / (65536/sqrt(3))), exact
    integer arithmetic followed by ONE fp32 multiply, hence bit-identical between
    numpy and the HIP kernel `atspeed_fill_hash_normal`.
    """
    out = np.empty(n, dtype=np.float32)
    c = normal_scale(std)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        idx = np.arange(lo + offset, hi + offset, dtype=np.uint64)
        h1 = hash_u32(idx, seed)
        h2 = hash_u32(idx, seed ^ 0x5BD1E995)
        s = (h1 & np.uint64(0xFFFF)) + (h1 >> np.uint64(16)) + (h2 & np.uint64(0xFFFF)) + (h2 >> np.uint64(16))
        out[lo:hi] = (s.astype(np.int64) - 131070).astype(np.float32) * c
    return out


def hash_randint(n: int, seed: int, lo: int, hi: int) -> np.ndarray:
    """Integers in [lo, hi) (tiny modulo bias is irrelevant for synthetic prompts)."""
    h = hash_u32(np.arange(n, dtype=np.uint64), seed)
    return (lo + (h % np.uint64(hi - lo))).astype(np.int64)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 bit pattern (finite inputs)."""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = (u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)
    return r.astype(np.uint16)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 values rounded to the nearest bf16, returned as fp32."""
    return (f32_to_bf16_bits(x).astype(np.uint32) << np.uint32(16)).view(np.float32)


# --------------------------------------------------------------------------- vocab

@dataclass(frozen=True)
class CodeVocab:
    """Token-id layout of an item-code vocabulary (L code positions + EOS)."""
    name: str
    level_sizes: Tuple[int, ...]
    n_items: int
    base: int = LLAMA_VOCAB

    @property
    def vocab_size(self) -> int:
        return self.base + sum(self.level_sizes)

    @property
    def n_levels(self) -> int:
        return len(self.level_sizes)

    def level_range(self, i: int) -> Tuple[int, int]:
        lo = self.base + sum(self.level_sizes[:i])
        return lo, lo + self.level_sizes[i]

    def allowed_tokens(self) -> Dict[int, List[int]]:
        """position -> allowed ids, EOS after the last code (reference code/data.py:84-94)."""
        d = {i: list(range(*self.level_range(i))) for i in range(self.n_levels)}
        d[self.n_levels] = [EOS_ID]
        return d


# cardinalities read from the reference's index files
BEAUTY = CodeVocab("beauty", (91, 256, 256, 256), 12035)
GAMES = CodeVocab("games", (248, 256, 254, 256), 17332)
# small layout for fixtures: four 64-wide ranges -> V = 32256
TINY = CodeVocab("tiny", (64, 64, 64, 64), 600)

# stand-in for tokenizer("Response:")["input_ids"][1:] (no tokenizer offline): any
# fixed id pair < 32000 that synthetic prompts do not otherwise contain.
RESPONSE_SEP = (13291, 29901)


def synthetic_items(vocab: CodeVocab, seed: int = 2025) -> np.ndarray:
    """[n_items, L] token ids of a synthetic item index with the vocab's level sizes.

    Level 0 is skewed like the real index (few first codes carry most items) by
    squaring a uniform draw; deeper levels are uniform.  Duplicates are kept, as in
    the reference index (12 035 items / 12 023 unique for Beauty).
    """
    n, L = vocab.n_items, vocab.n_levels
    items = np.empty((n, L), dtype=np.int64)
    for lvl in range(L):
        lo, hi = vocab.level_range(lvl)
        h = hash_u32(np.arange(n, dtype=np.uint64), tensor_seed(seed, f"items.{vocab.name}.{lvl}"))
        u = h.astype(np.float64) / 4294967296.0
        if lvl == 0:
            u = u * u
        items[:, lvl] = lo + np.minimum((u * (hi - lo)).astype(np.int64), hi - lo - 1)
    return items


def synthetic_prompt(length: int, seed: int, sep: Sequence[int] = RESPONSE_SEP) -> np.ndarray:
    """BOS + uniform ids in [3, 32000) avoiding sep[0], ending with the Response sep."""
    assert length >= len(sep) + 2
    body = hash_randint(length - 1 - len(sep), seed, 3, LLAMA_VOCAB)
    body[body == sep[0]] = sep[0] + 1
    return np.concatenate([[BOS_ID], body, np.asarray(sep, dtype=np.int64)]).astype(np.int64)


def prompt_lengths(n_users: int, seed: int = 2025, mean_hist: float = 7.33, max_hist: int = 20) -> np.ndarray:
    """Prompt lengths P ~ 64 + 6*H with H a truncated geometric history length
    (Beauty test users: H mean 7.33, max 20 -> P mean ~108)."""
    h = hash_u32(np.arange(n_users, dtype=np.uint64), tensor_seed(seed, "prompt_len"))
    u = (h.astype(np.float64) + 0.5) / 4294967296.0
    p = 1.0 / mean_hist
    H = np.minimum(1 + np.floor(np.log(u) / np.log(1.0 - p)).astype(np.int64), max_hist)
    return (64 + 6 * H).astype(np.int64)


# --------------------------------------------------------------------------- llama weights

@dataclass(frozen=True)
class LlamaDims:
    vocab_size: int
    hidden: int
    n_layers: int
    n_heads: int
    ffn: int
    rope_theta: float = 10000.0
    rms_eps: float = 1e-6

    @property
    def head_dim(self) -> int:
        return self.hidden // self.n_heads

    def n_params_streamed(self) -> int:
        """Matrix parameters read once per forward (W_t / W_d)."""
        return self.n_layers * (4 * self.hidden * self.hidden + 3 * self.hidden * self.ffn) + self.hidden * self.vocab_size


def llama_68m(vocab_size: int) -> LlamaDims:   # reference code/model.py:1023 (2 layers, 12 heads, 768)
    return LlamaDims(vocab_size, 768, 2, 12, 3072)


def llama_7b(vocab_size: int, n_layers: int = 32) -> LlamaDims:
    return LlamaDims(vocab_size, 4096, n_layers, 32, 11008)


def weight_specs(d: LlamaDims) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) in HF `LlamaForCausalLM.state_dict()` naming."""
    specs: List[Tuple[str, Tuple[int, ...], str]] = [("model.embed_tokens.weight", (d.vocab_size, d.hidden), "mat")]
    for l in range(d.n_layers):
        p = f"model.layers.{l}."
        specs += [
            (p + "input_layernorm.weight", (d.hidden,), "norm"),
            (p + "self_attn.q_proj.weight", (d.hidden, d.hidden), "mat"),
            (p + "self_attn.k_proj.weight", (d.hidden, d.hidden), "mat"),
            (p + "self_attn.v_proj.weight", (d.hidden, d.hidden), "mat"),
            (p + "self_attn.o_proj.weight", (d.hidden, d.hidden), "mat"),
            (p + "post_attention_layernorm.weight", (d.hidden,), "norm"),
            (p + "mlp.gate_proj.weight", (d.ffn, d.hidden), "mat"),
            (p + "mlp.up_proj.weight", (d.ffn, d.hidden), "mat"),
            (p + "mlp.down_proj.weight", (d.hidden, d.ffn), "mat"),
        ]
    specs += [("model.norm.weight", (d.hidden,), "norm"), ("lm_head.weight", (d.vocab_size, d.hidden), "mat")]
    return specs


def weight_std(name: str, d: LlamaDims, std: float, head_std: float) -> float:
    return head_std if name == "lm_head.weight" else std


def synthetic_state_dict(d: LlamaDims, seed: int, std: float = 0.02, head_std: float | None = None,
                         norm_jitter: float = 0.1, bf16: bool = False) -> Dict[str, np.ndarray]:
    """fp32 numpy weights (optionally bf16-rounded) for small models; large models are
    filled on the device by the same recipe (`HipLlama.from_synthetic`)."""
    head_std = std if head_std is None else head_std
    sd: Dict[str, np.ndarray] = {}
    for name, shape, kind in weight_specs(d):
        n = int(np.prod(shape))
        s = tensor_seed(seed, name)
        if kind == "norm":
            w = np.float32(1.0) + hash_normal(n, s, norm_jitter)
        else:
            w = hash_normal(n, s, weight_std(name, d, std, head_std))
        if bf16:
            w = bf16_round(w)
        sd[name] = w.reshape(shape)
    return sd


def perturbed_state_dict(sd: Dict[str, np.ndarray], seed: int, sigma: float) -> Dict[str, np.ndarray]:
    """draft = target + sigma * std(param) * N(0,1) per tensor (SURVEY.md section 8c fixture recipe)."""
    out = {}
    for name, w in sd.items():
        if sigma == 0.0:
            out[name] = w.copy()
            continue
        noise = hash_normal(w.size, tensor_seed(seed, "noise." + name), 1.0).reshape(w.shape)
        out[name] = (w + np.float32(sigma) * np.float32(w.astype(np.float64).std()) * noise).astype(np.float32)
    return out
