"""RMSNorm on the fused HIP kernels (mirror of minivllm/models/layers/layernorm.py:39-130)."""
from __future__ import annotations

import torch
from torch import nn

from .. import hip_ops

__all__ = ["RMSNorm"]


class RMSNorm(nn.Module):
    """``forward(x) -> (y, None)``; ``forward(x, residual) -> (y, x + residual)``.

    fp32 math inside the kernel; with a residual the norm is taken of the UNROUNDED fp32 sum
    (layernorm.py:95-105).  Unlike the reference, ``residual`` is updated in place (one HBM
    round trip instead of an allocation + two): the caller's tensor IS the returned residual."""

    def __init__(self, hidden_size: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.eps = float(eps)
        self.weight = nn.Parameter(torch.ones(hidden_size))

    def rms_forward(self, x):
        return hip_ops.rmsnorm(x, self.weight, self.eps)

    def add_rms_forward(self, x, residual):
        return hip_ops.add_rmsnorm(x, residual, self.weight, self.eps, inplace=True)

    def forward(self, x, residual=None):
        if hasattr(x, "to_tensor"):   # layers.linear.PartialSum: K-split partial sums of the projection in front
            if residual is not None and residual.is_contiguous():
                return hip_ops.add_rmsnorm_partial(x.data, residual, self.weight, self.eps)
            x = x.to_tensor()
        if residual is None:
            return self.rms_forward(x), None
        return self.add_rms_forward(x, residual)
