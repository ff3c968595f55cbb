"""Vocab-parallel embedding and LM head (mirror of minivllm/models/layers/embed_head.py:43-231)."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from ..parallel_state import (get_tensor_parallel_rank, get_tensor_parallel_world_size,
                              tp_all_gather_last_dim, tp_all_reduce)
from ..utils.context import get_context
from .. import hip_ops
from .linear import linear, mid_m, small_m, tuned_plan

__all__ = ["VocabParallelEmbedding", "ParallelLMHead", "get_tensor_parallel_rank",
           "get_tensor_parallel_world_size"]


class VocabParallelEmbedding(nn.Module):
    def __init__(self, num_embeddings: int, embedding_dim: int):
        super().__init__()
        self.tp_rank, self.tp_size = get_tensor_parallel_rank(), get_tensor_parallel_world_size()
        if num_embeddings % self.tp_size != 0:
            raise ValueError(f"num_embeddings ({num_embeddings}) must be divisible by "
                             f"tensor-parallel size ({self.tp_size})")
        self.num_embeddings = num_embeddings
        self.num_embeddings_per_partition = num_embeddings // self.tp_size
        self.vocab_start_idx = self.num_embeddings_per_partition * self.tp_rank
        self.vocab_end_idx = self.vocab_start_idx + self.num_embeddings_per_partition
        self.weight = nn.Parameter(torch.empty(self.num_embeddings_per_partition, embedding_dim))
        self.weight.weight_loader = self.weight_loader

    def weight_loader(self, param, loaded_weight, *args, **kwargs):
        from ..utils.loader import materialize

        shard = param.data.size(0)
        param.data.copy_(materialize(loaded_weight.narrow(0, self.tp_rank * shard, shard)))

    def extra_repr(self) -> str:
        return (f"num_embeddings={self.num_embeddings}, embedding_dim={self.weight.shape[1]}, "
                f"tp_size={self.tp_size}, tp_rank={self.tp_rank}")

    def forward(self, x):
        if self.tp_size == 1:
            return F.embedding(x, self.weight)
        mine = (x >= self.vocab_start_idx) & (x < self.vocab_end_idx)
        y = F.embedding(mine * (x - self.vocab_start_idx), self.weight)
        return tp_all_reduce(mine.unsqueeze(-1) * y)


class ParallelLMHead(VocabParallelEmbedding):
    def __init__(self, num_embeddings: int, embedding_dim: int, bias: bool = False):
        if bias:
            raise ValueError("bias for ParallelLMHead is not supported")
        super().__init__(num_embeddings, embedding_dim)
        self.weight_packed = None

    def pack_for_decode(self) -> bool:
        """Fragment-ordered copy of the (rank's shard of the) head for decode steps of up to 32 rows
        (csrc/linear_small_m.hip streams it in whole lines: 99 -> 91 us for Qwen3's 622 MB at one row)."""
        w = self.weight
        ok = w.is_cuda and w.dtype in (torch.bfloat16, torch.float16) and w.shape[1] % 64 == 0 and w.shape[0] % 4 == 0
        self.weight_packed = hip_ops.pack_weight(w.data) if ok else None
        return ok

    def forward(self, x, gather: bool = True):
        """``gather=False`` returns this rank's vocab shard ``[B, V/tp]`` (greedy decoding merges
        per-rank argmaxes instead of gathering the logits: parallel_state.tp_merge_greedy)."""
        ctx = get_context()
        if ctx.is_prefill:  # only each sequence's last token is sampled
            x = x[(ctx.cum_seqlens_q[1:] - 1).long()].contiguous()
        logits = None
        if small_m(x, self.weight):
            logits = hip_ops.linear_small_m(x, self.weight, packed=self.weight_packed)
        elif mid_m(x, self.weight_packed) and hip_ops.linear_mid_m_ok(x, self.weight.shape[0]):
            # decode batches: mv_linear_mid_m where it MEASURED faster than the library GEMM for this head
            # (tuned/mid_m_gfx950.json: Qwen3-1.7B at 256 rows 184 vs 200 us, Qwen2-7B 307 vs 332 us)
            plan = tuned_plan("col", x.numel() // x.shape[-1], self.weight.shape[0], self.weight.shape[1])
            if plan is not None and plan[0]:
                logits = hip_ops.linear_mid_m(x, self.weight_packed, self.weight.shape[0], None, plan[0])
        if logits is None:
            logits = linear(x, self.weight)
        return tp_all_gather_last_dim(logits) if self.tp_size > 1 and gather else logits
