"""SiluAndMul on the HIP kernel (mirror of minivllm/models/layers/activation.py:70-96)."""
from torch import nn

from .. import hip_ops

__all__ = ["SiluAndMul"]


class SiluAndMul(nn.Module):
    def forward(self, x):
        return hip_ops.silu_and_mul(x)  # raises ValueError on an odd last dim, like the reference
