"""ORACLE (test infrastructure, not product code): fp32 CPU restatement of the Llama
decoder forward that the reference's hot path calls at `code/beamSD.py:52` and
`code/beamSD.py:221`.

The arithmetic lives in a third-party dependency that is NOT vendored under
/root/reference: `transformers==4.41.0` (`README.md:11`), class `LlamaForCausalLM`
(modeling_llama.py).  Its published algorithm is restated here:

    h = embed[ids]
    per layer:  x = rmsnorm(h, w_in)                       (fp32 variance, eps)
                q,k,v = x Wq^T, x Wk^T, x Wv^T ; rotate-half RoPE(position_ids) on q,k
                K,V cache append ; scores = q k^T / sqrt(dh) + additive mask
                p = softmax_fp32(scores) ; a = p v ; h += a Wo^T
                x = rmsnorm(h, w_post) ; h += (silu(x Wg^T) * (x Wu^T)) Wd^T
    logits = rmsnorm(h, w_norm) Wlm^T   (fp32 logits, reference keeps the upcast:
                                          code/model.py:303-304)

Differences in formulation (not in result): the KV cache is slot-addressed and the
4-D additive mask (0 / finfo.min, `code/beamSD.py:89,204-209,399`) is carried as a
boolean visibility matrix `vis[t, s]` over cache slots; a masked score contributes
exactly 0 to the fp32 softmax either way.

Pinned by tests/test_oracle_pins.py against the installed HF `LlamaForCausalLM`
(eager, fp32) and, end to end, by the golden fixtures generated from the imported
reference (tests/golden/gen_golden.py).

`w8a8=True` restates BASELINE config 5 (fp8 target verification): the four layer
projections (q/k/v, o, gate/up, down) run on OCP e4m3 operands -- weights quantised per
output row, activations per token, scale = max|row| / 448, q = e4m3(x / scale) with
torch's float8_e4m3fn rounding -- with fp32 accumulation and the two scales applied to
the sum; embedding, norms, attention, KV cache and lm_head stay as above.  It stands where
the reference loads its target with `load_in_8bit=True` (bitsandbytes LLM.int8,
`code/inference.py:88`): that arithmetic is third-party, unpinned and absent offline, so
this mode defines the build's own 8-bit scheme and is pinned only to itself ("parity
unpinned" against the reference for config 5).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch


@dataclass
class RefDims:
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


def _t(x) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        return x.detach().to(torch.float32).cpu().contiguous()
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))


E4M3_MAX = 448.0


def quant_rows_e4m3(x: torch.Tensor):
    """Per-row e4m3 quantisation: (dequantised codes as fp32, scale[rows]); an all-zero row gets scale 1."""
    amax = x.abs().amax(-1)
    scale = torch.where(amax > 0, amax / E4M3_MAX, torch.ones_like(amax))
    q = (x / scale[..., None]).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32)
    return q, scale


class RefLlama:
    """Slot-addressed fp32 Llama; `state_dict` uses HF parameter names."""

    PROJ = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")

    def __init__(self, dims, state_dict: Dict[str, object], max_slots: int = 1024, w8a8: bool = False, dtype: torch.dtype = torch.float32):
        """dtype=torch.float64: the ARBITER mode (round 6) -- the same model (the fp32 weight VALUES, HF's fp32 rotary angles) evaluated in
        double precision: what two fp32 evaluations that sum in different orders are both approximating (tests/test_fulldims_gpu.py judges
        near-tied decisions with it).  The weights stay stored in fp32 and are widened per use, so the mode costs no second copy of a 7B model."""
        self.d = RefDims(dims.vocab_size, dims.hidden, dims.n_layers, dims.n_heads, dims.ffn,
                         dims.rope_theta, dims.rms_eps)
        self.w = {k: _t(v) for k, v in state_dict.items()}
        self.cdt = dtype
        assert dtype in (torch.float32, torch.float64) and not (w8a8 and dtype != torch.float32)
        self.w8a8 = w8a8
        self.wq: Dict[str, tuple] = {}
        if w8a8:
            for l in range(self.d.n_layers):
                for pj in self.PROJ:
                    name = f"model.layers.{l}.{pj}.weight"
                    self.wq[name] = quant_rows_e4m3(self.w[name])
        d = self.d
        self.kcache = torch.zeros(d.n_layers, max_slots, d.n_heads, d.head_dim, dtype=dtype)
        self.vcache = torch.zeros(d.n_layers, max_slots, d.n_heads, d.head_dim, dtype=dtype)
        half = d.head_dim // 2
        # HF LlamaRotaryEmbedding: inv_freq = 1 / theta^(2i/dh)
        self.inv_freq = 1.0 / (d.rope_theta ** (torch.arange(0, half, dtype=torch.float32) * 2.0 / d.head_dim))

    # -- pieces -----------------------------------------------------------------
    def _proj(self, x: torch.Tensor, name: str) -> torch.Tensor:
        """x W^T for a layer projection: fp32, or W8A8 (e4m3 x e4m3 products are exact in fp32; the sum is fp32)."""
        if not self.w8a8:
            return x @ self._W(name).T
        xq, sx = quant_rows_e4m3(x)
        wq, sw = self.wq[name]
        return (xq @ wq.T) * sx[:, None] * sw[None, :]

    def _W(self, name: str) -> torch.Tensor:
        """A weight in the compute type (fp32: the stored tensor itself; fp64: widened on use)."""
        return self.w[name].to(self.cdt)

    def _rmsnorm(self, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        var = x.pow(2).mean(-1, keepdim=True)
        return w * (x * torch.rsqrt(var + self.d.rms_eps))

    def _rope(self, x: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
        # x [T, H, dh]; rotate_half convention: pairs (i, i + dh/2)
        ang = (pos.to(torch.float32)[:, None] * self.inv_freq[None, :]).to(self.cdt)      # [T, dh/2]; the angle is fp32 in HF (part of the model), its cos / sin in the compute type
        cos = torch.cat((ang.cos(), ang.cos()), -1)[:, None, :]
        sin = torch.cat((ang.sin(), ang.sin()), -1)[:, None, :]
        half = x.shape[-1] // 2
        rot = torch.cat((-x[..., half:], x[..., :half]), -1)
        return x * cos + rot * sin

    # -- forward ----------------------------------------------------------------
    @torch.no_grad()
    def forward(self, ids, pos, slots, vis, n_logit_rows: Optional[int] = None) -> torch.Tensor:
        """ids/pos/slots: [T] ints; vis: bool [T, S] over cache slots (S > max(slots)).
        Writes K/V of the T tokens at `slots`, returns fp32 logits of the LAST
        n_logit_rows tokens ([n, V]); all rows when None."""
        d = self.d
        ids = torch.as_tensor(np.asarray(ids), dtype=torch.long)
        pos = torch.as_tensor(np.asarray(pos), dtype=torch.long)
        slots = torch.as_tensor(np.asarray(slots), dtype=torch.long)
        vis = torch.as_tensor(np.asarray(vis), dtype=torch.bool)
        T, S = vis.shape
        h = self.w["model.embed_tokens.weight"][ids].to(self.cdt)
        scale = 1.0 / math.sqrt(d.head_dim)
        neg = torch.finfo(torch.float32).min                         # (also in fp64 mode: a masked score contributes exactly 0 either way)
        for l in range(d.n_layers):
            p = f"model.layers.{l}."
            x = self._rmsnorm(h, self._W(p + "input_layernorm.weight"))
            q = self._proj(x, p + "self_attn.q_proj.weight").view(T, d.n_heads, d.head_dim)
            k = self._proj(x, p + "self_attn.k_proj.weight").view(T, d.n_heads, d.head_dim)
            v = self._proj(x, p + "self_attn.v_proj.weight").view(T, d.n_heads, d.head_dim)
            q = self._rope(q, pos)
            k = self._rope(k, pos)
            self.kcache[l, slots] = k
            self.vcache[l, slots] = v
            K = self.kcache[l, :S]                                   # [S, H, dh]
            V = self.vcache[l, :S]
            sc = torch.einsum("thd,shd->hts", q, K) * scale          # [H, T, S]
            sc = torch.where(vis[None], sc, torch.full_like(sc, neg))
            pr = torch.softmax(sc, dim=-1, dtype=self.cdt)
            a = torch.einsum("hts,shd->thd", pr, V).reshape(T, d.hidden)
            h = h + self._proj(a, p + "self_attn.o_proj.weight")
            x = self._rmsnorm(h, self._W(p + "post_attention_layernorm.weight"))
            g = self._proj(x, p + "mlp.gate_proj.weight")
            u = self._proj(x, p + "mlp.up_proj.weight")
            h = h + self._proj(torch.nn.functional.silu(g) * u, p + "mlp.down_proj.weight")
        if n_logit_rows is not None:
            h = h[T - n_logit_rows:]
        x = self._rmsnorm(h, self._W("model.norm.weight"))
        return (x @ self._W("lm_head.weight").T).to(self.cdt)        # fp32 logits (fp64 in the arbiter mode)
