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
        # the fp32 originals, as plain attributes: ``model.to(dtype)`` rounds the BUFFERS in place, and this module
        # is shared through ``get_rope``'s cache (like the reference's) -- by every layer of a model, but also by a
        # second model of another dtype built later in the same process, which must not inherit the first one's
        # rounded tables (found by an fp32 test that ran behind bf16 tests: 3e-3 of the logit scale)
        self._cos32, self._sin32 = cos.clone(), sin.clone()
        self._dev_tables: dict = {}

    def _tables(self, device, dtype: torch.dtype | None = None):
        """fp32 device tables for the kernel.  The reference's ``model.to(dtype)`` casts its cos/sin buffers, so its
        bf16 runs rotate with bf16-ROUNDED cos/sin (as HF does): ``dtype`` = the model / activation dtype reproduces
        that (rounded through ``dtype``, widened back to fp32); None = the unrounded fp32 tables."""
        dtype = dtype or torch.float32
        key = (str(device), dtype)
        hit = self._dev_tables.get(key)
        if hit is None:
            hit = tuple(t.to(dtype).to(device=device, dtype=torch.float32).contiguous() for t in (self._cos32, self._sin32))
            self._dev_tables[key] = hit
        return hit

    def forward(self, positions, query, key):
        """query [T,H,D], key [T,Hkv,D] -> rotated copies."""
        cos, sin = self._tables(query.device, query.dtype)
        return hip_ops.rope_qk_store(query, key, None, positions, cos, sin)


def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float,
             rope_scaling: dict[str, Any] | None = None) -> RotaryEmbedding:
    if rope_scaling is not None:  # dicts are unhashable: no cache, as in the reference
        return RotaryEmbedding(head_size, rotary_dim, max_position, base, rope_scaling)
    return _get_rope_cached(head_size, rotary_dim, max_position, base)


@lru_cache(4)
def _get_rope_cached(head_size: int, rotary_dim: int, max_position: int, base: float) -> RotaryEmbedding:
    return RotaryEmbedding(head_size, rotary_dim, max_position, base)
