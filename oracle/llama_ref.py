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


class RefLlama:
    """Slot-addressed fp32 Llama; `state_dict` uses HF parameter names."""

    def __init__(self, dims, state_dict: Dict[str, object], max_slots: int = 1024):
        self.d = RefDims(dims.vocab_size, dims.hidden, dims.n_layers, dims.n_heads, dims.ffn,
                         dims.rope_theta, dims.rms_eps)
        self.w = {k: _t(v) for k, v in state_dict.items()}
        d = self.d
        self.kcache = torch.zeros(d.n_layers, max_slots, d.n_heads, d.head_dim)
        self.vcache = torch.zeros(d.n_layers, max_slots, d.n_heads, d.head_dim)
        half = d.head_dim // 2
        # HF LlamaRotaryEmbedding: inv_freq = 1 / theta^(2i/dh)
        self.inv_freq = 1.0 / (d.rope_theta ** (torch.arange(0, half, dtype=torch.float32) * 2.0 / d.head_dim))

    # -- pieces -----------------------------------------------------------------
    def _rmsnorm(self, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        var = x.pow(2).mean(-1, keepdim=True)
        return w * (x * torch.rsqrt(var + self.d.rms_eps))

    def _rope(self, x: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
        # x [T, H, dh]; rotate_half convention: pairs (i, i + dh/2)
        ang = pos.to(torch.float32)[:, None] * self.inv_freq[None, :]      # [T, dh/2]
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
        h = self.w["model.embed_tokens.weight"][ids]
        scale = 1.0 / math.sqrt(d.head_dim)
        neg = torch.finfo(torch.float32).min
        for l in range(d.n_layers):
            p = f"model.layers.{l}."
            x = self._rmsnorm(h, self.w[p + "input_layernorm.weight"])
            q = (x @ self.w[p + "self_attn.q_proj.weight"].T).view(T, d.n_heads, d.head_dim)
            k = (x @ self.w[p + "self_attn.k_proj.weight"].T).view(T, d.n_heads, d.head_dim)
            v = (x @ self.w[p + "self_attn.v_proj.weight"].T).view(T, d.n_heads, d.head_dim)
            q = self._rope(q, pos)
            k = self._rope(k, pos)
            self.kcache[l, slots] = k
            self.vcache[l, slots] = v
            K = self.kcache[l, :S]                                   # [S, H, dh]
            V = self.vcache[l, :S]
            sc = torch.einsum("thd,shd->hts", q, K) * scale          # [H, T, S]
            sc = torch.where(vis[None], sc, torch.full_like(sc, neg))
            pr = torch.softmax(sc, dim=-1, dtype=torch.float32)
            a = torch.einsum("hts,shd->thd", pr, V).reshape(T, d.hidden)
            h = h + a @ self.w[p + "self_attn.o_proj.weight"].T
            x = self._rmsnorm(h, self.w[p + "post_attention_layernorm.weight"])
            g = x @ self.w[p + "mlp.gate_proj.weight"].T
            u = x @ self.w[p + "mlp.up_proj.weight"].T
            h = h + (torch.nn.functional.silu(g) * u) @ self.w[p + "mlp.down_proj.weight"].T
        if n_logit_rows is not None:
            h = h[T - n_logit_rows:]
        x = self._rmsnorm(h, self.w["model.norm.weight"])
        return (x @ self.w["lm_head.weight"].T).to(torch.float32)
