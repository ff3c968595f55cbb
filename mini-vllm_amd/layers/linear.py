"""Tensor-parallel linear layers (mirror of minivllm/models/layers/linear.py:36-319).
GEMMs stay on hipBLASLt through ``F.linear``, except for decode steps of up to ``SMALL_M_MAX`` tokens:
there every projection is a pure HBM stream of its weights and goes to the weight-streaming HIP kernel
(csrc/linear_small_m.hip).  What else matters here is which slice of a checkpoint tensor each rank owns
(the ``weight_loader`` hooks) and the all-reduce after row-parallel GEMMs."""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F
from torch import nn

from .. import hip_ops
from ..parallel_state import (divide, get_tensor_parallel_rank, get_tensor_parallel_world_size,
                              tp_all_reduce)

__all__ = ["LinearBase", "ReplicatedLinear", "ColumnParallelLinear", "MergedColumnParallelLinear",
           "QKVParallelLinear", "RowParallelLinear", "divide", "linear", "small_m"]

# rows up to which the weight-streaming kernel beats the library GEMM (tools/bench_linear.py); 0 disables
SMALL_M_MAX = int(os.environ.get("MV_LINEAR_SMALL_M_MAX", "16"))


def small_m(x, weight) -> bool:
    """Does this projection go to the weight-streaming kernel?  Up to SMALL_M_MAX rows always; up to twice that
    for the shapes where it still measured faster than hipBLASLt (tools/dbg/xlds_bench.sh, M = 32): the small
    layers (qkv / o of Qwen3-1.7B: 8.3 / 7.2 vs 11.8 / 11.2 us), the LM head (123 vs 149 us) and very deep K
    (Qwen2-7B down: 48 vs 108 us) -- not the 25-270 MB gate_up matrices."""
    if not x.is_cuda:
        return False
    rows = x.numel() // x.shape[-1]
    if not 0 < rows <= 2 * SMALL_M_MAX or not hip_ops.linear_small_m_ok(x, weight):
        return False
    if rows <= SMALL_M_MAX:
        return True
    n, k = weight.shape
    return n * k <= 10_000_000 or n >= 65536 or k >= 16384


def linear(x, weight, bias=None):
    """``F.linear`` semantics; decode-sized inputs take the HIP weight-streaming kernel."""
    if small_m(x, weight):
        return hip_ops.linear_small_m(x, weight, bias)
    return F.linear(x, weight, bias)


class LinearBase(nn.Module):
    def __init__(self, input_size: int, output_size: int, bias: bool = False, tp_dim: int | None = None):
        super().__init__()
        self.input_size, self.output_size, self.tp_dim = input_size, output_size, tp_dim
        self.tp_rank = get_tensor_parallel_rank()
        self.tp_size = get_tensor_parallel_world_size()
        self.weight = nn.Parameter(torch.empty(output_size, input_size))
        self.weight.weight_loader = self.weight_loader
        if bias:
            self.bias = nn.Parameter(torch.empty(output_size))
            self.bias.weight_loader = self.weight_loader
        else:
            self.register_parameter("bias", None)

    def extra_repr(self) -> str:
        return (f"input_size={self.input_size}, output_size={self.output_size}, "
                f"bias={self.bias is not None}, tp_dim={self.tp_dim}, tp_size={self.tp_size}")

    def forward(self, x):
        raise NotImplementedError("Subclasses must implement forward method")

    def weight_loader(self, param, loaded_weight, *args, **kwargs):
        raise NotImplementedError("Subclasses must implement weight_loader method")


class ReplicatedLinear(LinearBase):
    def weight_loader(self, param, loaded_weight, *args, **kwargs):
        param.data.copy_(loaded_weight)

    def forward(self, x):
        return linear(x, self.weight, self.bias)


class ColumnParallelLinear(LinearBase):
    """Output features split over ranks: rank r owns rows [r*n/tp, (r+1)*n/tp) of the weight."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        super().__init__(input_size, divide(output_size, get_tensor_parallel_world_size()), bias, 0)

    def weight_loader(self, param, loaded_weight, *args, **kwargs):
        shard = param.data.size(0)
        param.data.copy_(loaded_weight.narrow(0, self.tp_rank * shard, shard))

    def forward(self, x):
        return linear(x, self.weight, self.bias)


class MergedColumnParallelLinear(ColumnParallelLinear):
    """Several column-parallel matrices fused along the output dim (gate|up)."""

    def __init__(self, input_size: int, output_sizes: list[int], bias: bool = False):
        self.output_sizes = list(output_sizes)
        super().__init__(input_size, sum(output_sizes), bias)

    def weight_loader(self, param, loaded_weight, loaded_shard_id: int = 0, *args, **kwargs):
        offset = sum(self.output_sizes[:loaded_shard_id]) // self.tp_size
        size = self.output_sizes[loaded_shard_id] // self.tp_size
        piece = loaded_weight.chunk(self.tp_size, 0)[self.tp_rank]
        param.data.narrow(0, offset, size).copy_(piece)


class QKVParallelLinear(ColumnParallelLinear):
    """q|k|v fused; each rank owns its q heads and its kv heads, stored [q_r | k_r | v_r]."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int,
                 total_num_kv_heads: int | None = None, bias: bool = False):
        tp = get_tensor_parallel_world_size()
        total_num_kv_heads = total_num_kv_heads or total_num_heads
        self.head_size = head_size
        self.num_heads = divide(total_num_heads, tp)
        self.num_kv_heads = divide(total_num_kv_heads, tp)
        super().__init__(hidden_size, (total_num_heads + 2 * total_num_kv_heads) * head_size, bias)

    def weight_loader(self, param, loaded_weight, loaded_shard_id: str | None = None, *args, **kwargs):
        if loaded_shard_id is None:
            raise ValueError("loaded_shard_id must be provided for QKVParallelLinear")
        q, kv = self.num_heads * self.head_size, self.num_kv_heads * self.head_size
        try:
            offset, size = {"q": (0, q), "k": (q, kv), "v": (q + kv, kv)}[loaded_shard_id]
        except KeyError:
            raise AssertionError(f"Invalid shard_id: {loaded_shard_id}") from None
        piece = loaded_weight.chunk(self.tp_size, 0)[self.tp_rank]
        param.data.narrow(0, offset, size).copy_(piece)

    def extra_repr(self) -> str:
        return (f"hidden_size={self.input_size}, head_size={self.head_size}, num_heads={self.num_heads}, "
                f"num_kv_heads={self.num_kv_heads}, bias={self.bias is not None}, tp_size={self.tp_size}")


class RowParallelLinear(LinearBase):
    """Input features split over ranks; partial products are summed with one all-reduce.
    The bias lives on rank 0 only (linear.py:296-316)."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        super().__init__(divide(input_size, get_tensor_parallel_world_size()), output_size, bias, 1)

    def weight_loader(self, param, loaded_weight, *args, **kwargs):
        if param.data.dim() == 1:
            if self.tp_rank == 0:
                param.data.copy_(loaded_weight)
            return
        shard = param.data.size(1)
        param.data.copy_(loaded_weight.narrow(1, self.tp_rank * shard, shard))

    def forward(self, x, reduce: bool = True):
        """``reduce=False`` returns this rank's partial product: the caller sums it together with the
        residual add + RMSNorm that follows (parallel_state.tp_all_reduce_add_rmsnorm)."""
        y = linear(x, self.weight, self.bias if self.tp_rank == 0 else None)
        return tp_all_reduce(y) if self.tp_size > 1 and reduce else y
