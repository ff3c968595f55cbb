"""NeoX rotary embedding (mirror of minivllm/models/layers/rotary_embedding.py:83-242).
The fp32 cos/sin tables are built with the reference's exact operation order (their values
must match torch's fp32 cos/sin of fp32 ``arange * inv_freq``); application runs in the HIP
kernel, normally fused with qk-norm and the KV-cache store (see glue/qwen.py)."""
from __future__ import annotations

from functools import lru_cache
from typing import Any

import torch
from torch import nn

from .. import hip_ops

__all__ = ["RotaryEmbedding", "get_rope"]


def build_cos_sin(rotary_dim: int, max_position: int, base: float,
                  rope_scaling: dict[str, Any] | None = None):
    # always on the CPU: the table values must equal the reference's (it builds them before
    # moving the model to the device), and CPU/GPU cos/sin differ in the last ulp
    exponents = torch.arange(0, rotary_dim, 2, dtype=torch.float, device="cpu") / rotary_dim
    inv_freq = 1.0 / (base ** exponents)
    if rope_scaling:
        kind, factor = rope_scaling.get("type"), rope_scaling.get("factor", 1.0)
        if kind == "linear":
            inv_freq /= factor
        elif kind == "dynamic":  # NTK-style base stretch, as simplified by the reference
            base = base * (factor ** (rotary_dim / (rotary_dim - 2)))
            inv_freq = 1.0 / (base ** exponents)
    freqs = torch.einsum("i,j -> ij", torch.arange(max_position, dtype=torch.float, device="cpu"), inv_freq)
    return freqs.cos(), freqs.sin()


class RotaryEmbedding(nn.Module):
    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float,
                 rope_scaling: dict[str, Any] | None = None) -> None:
        super().__init__()
        if rotary_dim != head_size:
            raise ValueError("rotary_dim must equal head_size")
        self.head_size = head_size
        self.max_position_embeddings = max_position_embeddings
        cos, sin = build_cos_sin(rotary_dim, max_position_embeddings, base, rope_scaling)
        self.register_buffer("cos_cache", cos, persistent=False)  # [max_pos, D/2] fp32
        self.register_buffer("sin_cache", sin, persistent=False)

    def _tables(self, device):
        """fp32 device tables for the kernel.  ``model.to(dtype)`` casts the buffers like it does
        in the reference (whose bf16 runs therefore rotate with bf16-ROUNDED cos/sin, as HF does);
        the rounded values are widened back to fp32 here, so both behaviours are reproduced."""
        src = self.cos_cache
        cached = getattr(self, "_fp32_tables", None)
        if cached is None or cached[0] is not src or cached[1].device != device:
            cached = (src, self.cos_cache.to(device=device, dtype=torch.float32).contiguous(),
                      self.sin_cache.to(device=device, dtype=torch.float32).contiguous())
            self._fp32_tables = cached
        return cached[1], cached[2]

    def forward(self, positions, query, key):
        """query [T,H,D], key [T,Hkv,D] -> rotated copies."""
        cos, sin = self._tables(query.device)
        return hip_ops.rope_qk_store(query, key, None, positions, cos, sin)


def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float,
             rope_scaling: dict[str, Any] | None = None) -> RotaryEmbedding:
    if rope_scaling is not None:  # dicts are unhashable: no cache, as in the reference
        return RotaryEmbedding(head_size, rotary_dim, max_position, base, rope_scaling)
    return _get_rope_cached(head_size, rotary_dim, max_position, base)


@lru_cache(4)
def _get_rope_cached(head_size: int, rotary_dim: int, max_position: int, base: float) -> RotaryEmbedding:
    return RotaryEmbedding(head_size, rotary_dim, max_position, base)
