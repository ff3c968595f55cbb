"""OPT decoder glue (mirror of minivllm/models/opt.py): learned positions (offset 2), MHA,
LayerNorm, ReLU MLP, pre- or post-LN.  Attention runs on the same HIP kernels (D = 64 for
OPT-125M); LayerNorm/ReLU/bias adds stay torch ops — OPT is the reference's CPU plumbing
config, not a throughput target."""
from __future__ import annotations

from typing import Any

import torch.nn.functional as F
from torch import nn

from ..layers import (Attention, ColumnParallelLinear, ParallelLMHead, QKVParallelLinear, RowParallelLinear,
                      VocabParallelEmbedding)
from ..parallel_state import get_tensor_parallel_world_size

__all__ = ["OPTForCausalLM"]


class OPTLearnedPositionalEmbedding(nn.Embedding):
    def __init__(self, num_embeddings: int, embedding_dim: int):
        self.offset = 2
        super().__init__(num_embeddings + self.offset, embedding_dim)

    def forward(self, positions):
        return super().forward(positions + self.offset)


class OPTAttention(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, bias: bool = True) -> None:
        super().__init__()
        if embed_dim % num_heads:
            raise ValueError(f"embed_dim must be divisible by num_heads (got 'embed_dim': {embed_dim} "
                             f"and 'num_heads': {num_heads}).")
        tp = get_tensor_parallel_world_size()
        self.embed_dim, self.total_num_heads = embed_dim, num_heads
        self.num_heads = num_heads // tp
        self.head_dim = embed_dim // num_heads
        self.scaling = self.head_dim ** -0.5
        self.qkv_proj = QKVParallelLinear(embed_dim, self.head_dim, num_heads, bias=bias)
        self.out_proj = RowParallelLinear(embed_dim, embed_dim, bias=bias)
        self.attn = Attention(self.num_heads, self.head_dim, self.scaling, self.num_heads)

    def forward(self, positions, hidden_states):
        t = hidden_states.shape[0]
        qkv = self.qkv_proj(hidden_states)
        width = self.num_heads * self.head_dim
        q, k, v = (qkv[:, i * width:(i + 1) * width].view(t, self.num_heads, self.head_dim) for i in range(3))
        o = self.attn(q, k, v)  # stores K/V, then prefill / paged decode
        return self.out_proj(o.reshape(t, width))


class OPTDecoderLayer(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        d = config.hidden_size
        self.self_attn = OPTAttention(d, config.num_attention_heads, bias=config.enable_bias)
        self.do_layer_norm_before = config.do_layer_norm_before
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = ColumnParallelLinear(d, config.ffn_dim, bias=config.enable_bias)
        self.fc2 = RowParallelLinear(config.ffn_dim, d, bias=config.enable_bias)
        self.final_layer_norm = nn.LayerNorm(d)

    def forward(self, positions, x, residual=None):
        if self.do_layer_norm_before:
            x = x + self.self_attn(positions, self.self_attn_layer_norm(x))
            x = x + self.fc2(F.relu(self.fc1(self.final_layer_norm(x))))
        else:
            x = self.self_attn_layer_norm(x + self.self_attn(positions, x))
            x = self.final_layer_norm(x + self.fc2(F.relu(self.fc1(x))))
        return x, None


class OPTDecoder(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        self.config = config
        d, w = config.hidden_size, config.word_embed_proj_dim
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, d)
        self.embed_positions = OPTLearnedPositionalEmbedding(config.max_position_embeddings, d)
        self.project_in = nn.Linear(w, d, bias=False) if w != d else None
        self.project_out = nn.Linear(d, w, bias=False) if w != d else None
        self.final_layer_norm = nn.LayerNorm(d) if config.do_layer_norm_before else None
        self.layers = nn.ModuleList(OPTDecoderLayer(config) for _ in range(config.num_hidden_layers))

    def forward(self, input_ids, positions):
        x = self.embed_tokens(input_ids)
        if self.project_in is not None:
            x = self.project_in(x)
        x = x + self.embed_positions(positions)
        for layer in self.layers:
            x, _ = layer(positions, x, None)
        if self.final_layer_norm is not None:
            x = self.final_layer_norm(x)
        if self.project_out is not None:
            x = self.project_out(x)
        return x


class OPTModel(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        self.decoder = OPTDecoder(config)

    def forward(self, input_ids, positions):
        return self.decoder(input_ids, positions)


class OPTForCausalLM(nn.Module):
    packed_modules_mapping = {"q_proj": ("qkv_proj", "q"), "k_proj": ("qkv_proj", "k"), "v_proj": ("qkv_proj", "v")}

    def __init__(self, config) -> None:
        super().__init__()
        self.config = config
        self.model = OPTModel(config)
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if config.tie_word_embeddings:
            self.lm_head.weight = self.model.decoder.embed_tokens.weight

    def forward(self, input_ids, positions):
        return self.model(input_ids, positions)

    def compute_logits(self, hidden_states, gather: bool = True):
        return self.lm_head(hidden_states, gather=gather)

    def set_kv_cache(self, kv_cache) -> None:
        for layer, (k, v) in zip(self.model.decoder.layers, kv_cache):
            layer.self_attn.attn.k_cache, layer.self_attn.attn.v_cache = k, v

    def load_weights(self, weights: dict[str, Any]) -> None:
        from ..utils.loader import load_weight

        for name, tensor in weights.items():
            load_weight(self, name, tensor, self.packed_modules_mapping)
