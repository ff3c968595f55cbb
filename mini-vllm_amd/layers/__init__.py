from .activation import SiluAndMul
from .attention import Attention
from .embed_head import ParallelLMHead, VocabParallelEmbedding
from .layernorm import RMSNorm
from .linear import (ColumnParallelLinear, MergedColumnParallelLinear, QKVParallelLinear,
                     ReplicatedLinear, RowParallelLinear)
from .rotary_embedding import RotaryEmbedding, get_rope
from .sampler import Sampler

__all__ = ["Attention", "ColumnParallelLinear", "MergedColumnParallelLinear", "ParallelLMHead",
           "QKVParallelLinear", "RMSNorm", "ReplicatedLinear", "RotaryEmbedding", "RowParallelLinear",
           "Sampler", "SiluAndMul", "VocabParallelEmbedding", "get_rope"]
