"""Attention over the paged KV cache on the gfx950 kernels (replaces
minivllm/models/layers/attention.py:103-597 and the backends of attention_backend.py).

There is exactly one backend — the HIP kernels behind include/mvhip.h; the reference's
flash-attn / SDPA / NPU dispatch does not exist here.  The module keeps the reference's
contract: ``forward(q, k, v)`` with q [N,H,D], k/v [N,Hkv,D]; implicit arguments arrive through
``get_context()``; ``k_cache`` / ``v_cache`` are plain attributes assigned by the executor.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from .. import hip_ops
from ..utils.context import get_context

__all__ = ["Attention"]


class Attention(nn.Module):
    def __init__(self, num_heads: int, head_dim: int, scale: float, num_kv_heads: int,
                 use_buffered_page_attention: bool = False) -> None:
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.scale = float(scale) if scale != 0.0 else 1.0 / math.sqrt(head_dim)
        self.num_kv_heads = num_kv_heads
        self.k_cache: torch.Tensor = torch.tensor([])
        self.v_cache: torch.Tensor = torch.tensor([])
        self._cache_initialized = False

    def extra_repr(self) -> str:
        return (f"num_heads={self.num_heads}, head_dim={self.head_dim}, num_kv_heads={self.num_kv_heads}, "
                f"scale={self.scale}, backend=mvhip")

    def store_kv_cache(self, key, value, slot_mapping) -> None:
        hip_ops.store_kvcache(key, value, self.k_cache, self.v_cache, slot_mapping)

    def forward(self, q, k, v, kv_stored: bool = False):
        """``kv_stored=True`` when the caller already wrote k/v into the cache (the fused
        RoPE+store kernel); otherwise this call stores them first, like the reference."""
        ctx = get_context()
        has_cache = self.k_cache.numel() > 0 and self.v_cache.numel() > 0
        if has_cache:
            self._cache_initialized = True
            if q.dtype != self.k_cache.dtype:  # attention.py:228-234
                q, k, v = (None if t is None else t.to(self.k_cache.dtype) for t in (q, k, v))
            if not kv_stored and ctx.slot_mapping is not None and ctx.slot_mapping.numel() > 0:
                self.store_kv_cache(k, v, ctx.slot_mapping)
        elif not ctx.is_prefill:
            raise RuntimeError("KV cache has not been initialized. Ensure the executor allocated it "
                               "before the decode phase.")
        if ctx.is_prefill:
            if ctx.cum_seqlens_q is None or ctx.cum_seqlens_k is None:
                raise RuntimeError("Cumulative sequence lengths not set in context for prefill phase. "
                                   "This is a bug in the inference pipeline.")
            if ctx.decode_rows and has_cache and ctx.block_tables is not None:
                # mixed batch: the first rows are one-token continuations of cached sequences (the paged-decode
                # kernel reads their K/V once, at HBM speed), the rest is prefill through the paged cache
                nd = ctx.decode_rows
                out = torch.empty_like(q)
                hip_ops.paged_decode(q[:nd], self.k_cache, self.v_cache, ctx.block_tables[:nd], ctx.context_lens,
                                     self.scale, out=out[:nd])
                if q.shape[0] > nd:
                    hip_ops.prefill_varlen(q[nd:], None, None, ctx.rest_cum_seqlens_q, ctx.rest_cum_seqlens_k,
                                           ctx.rest_max_seqlen_q, ctx.rest_max_seqlen_k, self.scale,
                                           k_cache=self.k_cache, v_cache=self.v_cache,
                                           block_tables=ctx.block_tables[nd:], out=out[nd:])
                return out
            if ctx.prefix_cached and has_cache and ctx.block_tables is not None:
                # prefix-cache hit: the queries are the uncached suffix, the keys live in the cache
                return hip_ops.prefill_varlen(q, None, None, ctx.cum_seqlens_q, ctx.cum_seqlens_k,
                                              ctx.max_seqlen_q, ctx.max_seqlen_k, self.scale,
                                              k_cache=self.k_cache, v_cache=self.v_cache,
                                              block_tables=ctx.block_tables)
            return hip_ops.prefill_varlen(q, k, v, ctx.cum_seqlens_q, ctx.cum_seqlens_k,
                                          ctx.max_seqlen_q, ctx.max_seqlen_k, self.scale)
        if ctx.context_lens is None or ctx.block_tables is None:
            raise RuntimeError("Context lengths or block tables not set for decode phase. "
                               "This is a bug in the inference pipeline.")
        return hip_ops.paged_decode(q, self.k_cache, self.v_cache, ctx.block_tables, ctx.context_lens,
                                    self.scale)
