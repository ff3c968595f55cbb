"""Qwen2 / Qwen3 decoder glue (mirror of minivllm/models/qwen_base.py, qwen2.py, qwen3.py).

Op order per layer fixes the fusion boundaries:
    qkv GEMM -> [q/k head RMSNorm] -> RoPE -> KV store   == ONE HIP launch (mv_rope_qk_store)
    -> attention (varlen prefill | paged decode)          == ONE HIP launch
    -> o_proj (+all-reduce) -> add+RMSNorm (fused HIP) -> gate_up GEMM -> SiluAndMul (HIP)
    -> down GEMM (+all-reduce) -> add+RMSNorm (fused HIP)
GEMMs go to hipBLASLt through F.linear.  Qwen2 = qkv bias, no qk-norm, theta 1e6;
Qwen3 = no bias, per-head q/k RMSNorm, theta 1e4 default (qwen2.py:23-25, qwen3.py:23-25).
"""
from __future__ import annotations

import os
from typing import Any

import torch
from torch import nn

from .. import hip_ops
from ..layers import (Attention, MergedColumnParallelLinear, ParallelLMHead, QKVParallelLinear, RMSNorm,
                      RowParallelLinear, SiluAndMul, VocabParallelEmbedding, get_rope)
from ..layers.linear import mid_m, mid_plan, small_m
from ..parallel_state import get_tensor_parallel_world_size, tp_all_reduce_add_rmsnorm
from ..utils.context import get_context

__all__ = ["QwenAttention", "QwenMLP", "QwenDecoderLayer", "QwenModel", "QwenForCausalLM",
           "Qwen2ForCausalLM", "Qwen3ForCausalLM"]


def _rope_theta(config, default: float) -> float:
    """transformers >= 5 moved rope_theta into ``rope_parameters`` (qwen_base.py:31-41)."""
    if hasattr(config, "rope_theta"):
        return config.rope_theta
    for attr in ("rope_parameters", "rope_scaling"):
        d = getattr(config, attr, None) or {}
        if isinstance(d, dict) and "rope_theta" in d:
            return d["rope_theta"]
    return default


def _rope_scaling(config) -> dict[str, Any] | None:
    raw = getattr(config, "rope_scaling", None)
    if not isinstance(raw, dict):
        return None
    keep = {"factor", "type", "low_freq_factor", "high_freq_factor", "original_max_position_embeddings",
            "short_factor", "long_factor"}
    return {k: v for k, v in raw.items() if k in keep} or None


class PendingNorm:
    """The input of a layer norm whose launch is folded into the projection behind it (decode steps of <= 4 rows,
    hip_ops.norm_linear_small_m): ``x`` (+ ``residual``) and the norm's weight / eps.  The projection that consumes
    it leaves the new residual in ``residual_out``."""
    __slots__ = ("x", "residual", "norm", "residual_out")

    def __init__(self, x, residual, norm):
        self.x, self.residual, self.norm, self.residual_out = x, residual, norm, None

    @property
    def shape(self):
        return self.x.shape

    def dim(self):
        return self.x.dim()

    def project(self, lin, gated: bool = False):
        """RMSNorm + ``lin`` (a LinearBase with a packed weight) in one launch; None when the kernel declines the
        shape (the caller then reads ``materialize()``)."""
        n = lin.weight.shape[0] // 2 if gated else lin.weight.shape[0]
        got = hip_ops.norm_linear_small_m(self.x, self.residual, self.norm.weight, self.norm.eps, lin.weight_packed,
                                          n, None if gated else lin.bias, gated)
        if got is None:
            return None
        out, self.residual_out = got
        return out

    def materialize(self):
        """The norm as a launch of its own."""
        if self.residual is None:
            self.residual_out = self.x
            return self.norm(self.x)[0]
        h, self.residual_out = self.norm(self.x, self.residual)
        return h


def _fold_norm(x, residual, lin) -> bool:
    """Fold the layer norm in front of ``lin`` into its launch?  Single GPU, 2-D activations of <= 4 rows (8 with MV_NORM_FOLD_MAX_ROWS=8: measured slower) whose
    projection runs on the packed weight (csrc/linear_small_m.hip, norm_prologue)."""
    return (NORM_FOLD and torch.is_tensor(x) and get_tensor_parallel_world_size() == 1
            and (residual is None or torch.is_tensor(residual))
            and small_m(x, lin.weight) and hip_ops.norm_linear_small_m_ok(x, lin.weight.shape[0], lin.weight_packed))


NORM_FOLD = os.environ.get("MV_NORM_FOLD", "1") != "0"
ROPE_FOLD = os.environ.get("MV_ROPE_FOLD", "1") != "0"
# rows up to which a decode step folds head norms + RoPE + K/V store into the attention launch: a launch less per layer
# (B = 1: 1.42 -> 1.38 ms, B = 16 ... 128: -1.2 ... -2 %); at B = 256 the work costs the attention kernel what the
# separate launch cost (7.00 ms both ways), so larger steps keep mv_rope_qk_store
ROPE_FOLD_MAX_ROWS = int(os.environ.get("MV_ROPE_FOLD_MAX", "128"))
# decode steps of <= 16 rows: every decoder layer in ONE persistent launch (csrc/decode_persistent.hip), bit-identical to
# the multi-launch path.  OFF by default: measured SLOWER than the six launches per layer it replaces (round 4, Qwen3-1.7B:
# B = 1 1.55 vs 1.26 ms per step, B = 8 2.23 vs 1.65 — profiles/r04_persistent_decode.md says which phases lose);
# MV_PERSISTENT=1 selects it.  MV_PERSISTENT_PREFETCH = weight chunks per wave in flight across a barrier (default 2,
# "none" = 0xff); MV_PERSISTENT_STAMPS=1 = phase stamps (diagnostic)
PERSISTENT = os.environ.get("MV_PERSISTENT", "0") == "1"
PERSISTENT_FLAGS = {"none": 0xff}.get(os.environ.get("MV_PERSISTENT_PREFETCH", ""), None)
if PERSISTENT_FLAGS is None:
    PERSISTENT_FLAGS = int(os.environ.get("MV_PERSISTENT_PREFETCH", "0") or 0)
if os.environ.get("MV_PERSISTENT_STAMPS") == "1":
    PERSISTENT_FLAGS |= 0x100


class QwenAttention(nn.Module):
    default_qkv_bias: bool = True
    default_rope_theta: float = 1000000

    def __init__(self, hidden_size: int, num_heads: int, num_kv_heads: int, max_position: int = 4096 * 32,
                 head_dim: int | None = None, rms_norm_eps: float = 1e-06, qkv_bias: bool | None = None,
                 rope_theta: float | None = None, rope_scaling: dict[str, Any] | None = None,
                 use_buffered_page_attention: bool = False) -> None:
        super().__init__()
        qkv_bias = self.default_qkv_bias if qkv_bias is None else qkv_bias
        rope_theta = self.default_rope_theta if rope_theta is None else rope_theta
        tp = get_tensor_parallel_world_size()
        self.total_num_heads, self.total_num_kv_heads = num_heads, num_kv_heads
        if num_heads % tp != 0:
            raise ValueError(f"total_num_heads ({num_heads}) must be divisible by tp_size ({tp})")
        if num_kv_heads % tp != 0:
            raise ValueError(f"total_num_kv_heads ({num_kv_heads}) must be divisible by tp_size ({tp})")
        self.num_heads, self.num_kv_heads = num_heads // tp, num_kv_heads // tp
        self.head_dim = int(head_dim or (hidden_size // num_heads))
        self.q_size, self.kv_size = self.num_heads * self.head_dim, self.num_kv_heads * self.head_dim
        self.scaling = self.head_dim ** -0.5
        self.qkv_bias = qkv_bias
        self.rms_norm_eps = float(rms_norm_eps)
        self.qkv_proj = QKVParallelLinear(hidden_size, self.head_dim, num_heads, num_kv_heads, bias=qkv_bias)
        self.o_proj = RowParallelLinear(num_heads * self.head_dim, hidden_size, bias=False)
        self.rotary_emb = get_rope(self.head_dim, rotary_dim=self.head_dim, max_position=max_position,
                                   base=rope_theta, rope_scaling=rope_scaling)
        self.attn = Attention(self.num_heads, self.head_dim, self.scaling, self.num_kv_heads)
        # q/k head norms exist iff there is no qkv bias (Qwen3), qwen_base.py:139-144
        self.q_norm = None if qkv_bias else RMSNorm(self.head_dim, eps=rms_norm_eps)
        self.k_norm = None if qkv_bias else RMSNorm(self.head_dim, eps=rms_norm_eps)

    def forward(self, positions, hidden_states, reduce: bool = True):
        lead = hidden_states.shape[:-1]
        if hidden_states.dim() == 3:  # [B,S,h]: same math on the flattened tokens
            hidden_states = hidden_states.reshape(-1, hidden_states.shape[-1])
            positions = positions.reshape(-1)
        t = hidden_states.shape[0]
        if isinstance(hidden_states, PendingNorm):   # the input norm rides in the qkv launch
            qkv = hidden_states.project(self.qkv_proj)
            if qkv is None:
                qkv = self.qkv_proj(hidden_states.materialize())
        else:
            qkv = self.qkv_proj(hidden_states)
        q = qkv[:, :self.q_size].view(t, self.num_heads, self.head_dim)
        k = qkv[:, self.q_size:self.q_size + self.kv_size].view(t, self.num_kv_heads, self.head_dim)
        v = qkv[:, self.q_size + self.kv_size:].view(t, self.num_kv_heads, self.head_dim)
        ctx = get_context()
        attn = self.attn
        has_cache = attn.k_cache.numel() > 0
        cos, sin = self.rotary_emb._tables(qkv.device, qkv.dtype)
        if (ROPE_FOLD and t <= ROPE_FOLD_MAX_ROWS and has_cache and not ctx.is_prefill and ctx.slot_mapping is not None
                and ctx.block_tables is not None and ctx.context_lens is not None):
            # decode step: head norms, RoPE and the K/V store ride in the attention launch (mv_rope_paged_decode)
            o = hip_ops.rope_paged_decode(
                q, k, v, positions, cos, sin, None if self.q_norm is None else self.q_norm.weight,
                None if self.k_norm is None else self.k_norm.weight, self.rms_norm_eps, ctx.slot_mapping,
                attn.k_cache, attn.v_cache, ctx.block_tables, ctx.context_lens, attn.scale)
            if o is not None:
                out = self.o_proj(o.view(t, self.q_size), reduce=reduce, partial_ok=len(lead) == 1)
                return out if hasattr(out, "to_tensor") else out.view(*lead, out.shape[-1])
        # prefill over contiguous K needs the rotated keys as a tensor; decode and prefix-cached
        # prefill read them back from the paged cache
        want_k = ctx.is_prefill and not (ctx.prefix_cached and has_cache)
        q_rot, k_rot = hip_ops.rope_qk_store(
            q, k, v, positions, cos, sin,
            None if self.q_norm is None else self.q_norm.weight,
            None if self.k_norm is None else self.k_norm.weight, self.rms_norm_eps,
            attn.k_cache if has_cache else None, attn.v_cache if has_cache else None,
            ctx.slot_mapping if has_cache else None, want_k_out=want_k)
        o = attn(q_rot, k_rot, v, kv_stored=True)
        out = self.o_proj(o.view(t, self.q_size), reduce=reduce, partial_ok=len(lead) == 1)
        return out if hasattr(out, "to_tensor") else out.view(*lead, out.shape[-1])

    def extra_repr(self) -> str:
        return (f"num_heads={self.num_heads}, num_kv_heads={self.num_kv_heads}, head_dim={self.head_dim}, "
                f"qkv_bias={self.qkv_bias}")


class QwenMLP(nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int, hidden_act: str) -> None:
        super().__init__()
        if hidden_act != "silu":
            raise ValueError(f"Unsupported activation function: '{hidden_act}'. Supported activations: ['silu']")
        self.gate_up_proj = MergedColumnParallelLinear(hidden_size, [intermediate_size] * 2, bias=False)
        self.down_proj = RowParallelLinear(intermediate_size, hidden_size, bias=False)
        self.act_fn = SiluAndMul()
        self.hidden_act, self.intermediate_size, self.hidden_size = hidden_act, intermediate_size, hidden_size

    def forward(self, x, reduce: bool = True):
        w, gu = self.gate_up_proj.weight, self.gate_up_proj
        flat = x.dim() == 2
        if isinstance(x, PendingNorm):   # the post-attention norm rides in the gate_up launch
            act = x.project(gu, gated=True)
            if act is not None:
                return self.down_proj(act, reduce=reduce, partial_ok=flat)
            x = x.materialize()
        if small_m(x, w):  # decode-sized batch: SiluAndMul in the epilogue of the weight-streaming gate_up kernel
            return self.down_proj(hip_ops.gate_up_silu_small_m(x, w, packed=gu.weight_packed), reduce=reduce,
                                  partial_ok=flat)
        n_inter = w.shape[0] // 2
        if mid_m(x, gu.weight_packed) and n_inter % 16 == 0 and hip_ops.linear_mid_m_ok(x, n_inter):
            cfg, _ = mid_plan("gated", x.numel() // x.shape[-1], n_inter, w.shape[1])
            if cfg:
                act = hip_ops.gate_up_silu_mid_m(x, gu.weight_packed, n_inter, cfg)
                return self.down_proj(act, reduce=reduce, partial_ok=flat)
        return self.down_proj(self.act_fn(self.gate_up_proj(x)), reduce=reduce, partial_ok=flat)


class QwenDecoderLayer(nn.Module):
    attention_cls = QwenAttention

    def __init__(self, config) -> None:
        super().__init__()
        acls = self.attention_cls
        self.self_attn = acls(
            hidden_size=config.hidden_size, num_heads=config.num_attention_heads,
            num_kv_heads=config.num_key_value_heads, max_position=config.max_position_embeddings,
            rms_norm_eps=config.rms_norm_eps, qkv_bias=getattr(config, "attention_bias", acls.default_qkv_bias),
            head_dim=getattr(config, "head_dim", None), rope_theta=_rope_theta(config, acls.default_rope_theta),
            rope_scaling=_rope_scaling(config))
        self.mlp = QwenMLP(config.hidden_size, config.intermediate_size, config.hidden_act)
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, positions, hidden_states, residual, pending_reduce: bool = False):
        """``pending_reduce``: ``hidden_states`` is the previous layer's un-reduced down_proj output (tensor
        parallel): its all-reduce runs fused with this layer's input norm.  Under TP the layer defers both of
        its own all-reduces the same way and returns the un-reduced MLP output (see ``defers_reduce``)."""
        ln1, ln2, defer = self.input_layernorm, self.post_attention_layernorm, self.defers_reduce
        if not pending_reduce and _fold_norm(hidden_states, residual, self.self_attn.qkv_proj) \
                and _fold_norm(hidden_states, residual, self.mlp.gate_up_proj):
            # decode steps of <= 4 rows: both norms ride in the launches of the projections behind them
            pend = PendingNorm(hidden_states, residual, ln1)
            hidden_states = self.self_attn(positions, pend, reduce=True)
            pend = PendingNorm(hidden_states, pend.residual_out, ln2)
            return self.mlp(pend, reduce=True), pend.residual_out
        if residual is None:
            residual = hidden_states
            hidden_states, _ = ln1(hidden_states)
        elif pending_reduce:
            hidden_states, residual = tp_all_reduce_add_rmsnorm(hidden_states, residual, ln1.weight, ln1.eps)
        else:
            hidden_states, residual = ln1(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states, reduce=not defer)
        if defer:
            hidden_states, residual = tp_all_reduce_add_rmsnorm(hidden_states, residual, ln2.weight, ln2.eps)
        else:
            hidden_states, residual = ln2(hidden_states, residual)
        return self.mlp(hidden_states, reduce=not defer), residual

    @property
    def defers_reduce(self) -> bool:
        """Tensor parallel, 2-D token-major activations: the all-reduces after o_proj / down_proj are issued
        together with the residual add + RMSNorm behind them (one launch on the peer-to-peer kernel)."""
        return get_tensor_parallel_world_size() > 1


class QwenModel(nn.Module):
    decoder_layer_cls = QwenDecoderLayer

    def __init__(self, config) -> None:
        super().__init__()
        self.config = config
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList(self.decoder_layer_cls(config) for _ in range(config.num_hidden_layers))
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    # ---- small decode steps: all layers in one persistent launch -----------------------------------------------------
    def _persistent_state(self, x):
        """Layer table + workspace + status word for mv_decode_layers_persistent, (re)built when the KV planes or the
        packed weights moved; None when the model does not qualify (unpacked weights, tensor parallel, other dtypes)."""
        layers = list(self.layers)
        a0 = layers[0].self_attn
        if (get_tensor_parallel_world_size() != 1 or x.dtype not in (torch.bfloat16, torch.float16) or a0.head_dim != 128
                or any(lin.weight_packed is None for l in layers
                       for lin in (l.self_attn.qkv_proj, l.self_attn.o_proj, l.mlp.gate_up_proj, l.mlp.down_proj))
                or a0.o_proj.bias is not None or layers[0].mlp.down_proj.bias is not None):
            return None
        key = tuple((l.self_attn.attn.k_cache.data_ptr(), l.self_attn.qkv_proj.weight_packed.data_ptr()) for l in layers)
        st = getattr(self, "_persist", None)
        if st is None or st["key"] != key:
            rows = []
            for l in layers:
                at, mlp = l.self_attn, l.mlp
                rows.append(dict(ln1_w=l.input_layernorm.weight, wqkv=at.qkv_proj.weight_packed, bqkv=at.qkv_proj.bias,
                                 q_norm_w=None if at.q_norm is None else at.q_norm.weight,
                                 k_norm_w=None if at.k_norm is None else at.k_norm.weight, wo=at.o_proj.weight_packed,
                                 ln2_w=l.post_attention_layernorm.weight, wgu=mlp.gate_up_proj.weight_packed,
                                 wdown=mlp.down_proj.weight_packed, k_cache=at.attn.k_cache, v_cache=at.attn.v_cache))
            st = self._persist = dict(
                key=key, table=hip_ops.persist_layer_table(rows, x.device), keep=rows,
                status=torch.zeros(32, dtype=torch.int32, device=x.device),
                status_host=torch.zeros(32, dtype=torch.int32).pin_memory(), ws={})
        return st

    def _forward_persistent(self, hidden_states, positions):
        """-> (h, residual) of the last layer, or None (the caller then runs the layers one launch at a time)."""
        ctx = get_context()
        t = hidden_states.shape[0] if hidden_states.dim() == 2 else 0
        a0 = self.layers[0].self_attn
        if (not PERSISTENT or not 0 < t <= hip_ops.PERSIST_MAX_ROWS or ctx.is_prefill or ctx.slot_mapping is None
                or ctx.block_tables is None or ctx.context_lens is None or a0.attn.k_cache.numel() == 0
                or ctx.block_tables.dim() != 2 or ctx.block_tables.shape[1] == 0):
            return None
        st = self._persistent_state(hidden_states)
        if st is None or st.get("declined"):
            return None
        kc = a0.attn.k_cache
        width = ctx.block_tables.shape[1]
        ws = st["ws"].get((t, width))
        if ws is None:
            ws = st["ws"][(t, width)] = hip_ops.decode_layers_persistent_workspace(
                t, hidden_states.shape[1], a0.num_heads, a0.num_kv_heads, a0.head_dim, self.layers[0].mlp.intermediate_size,
                kc.shape[1], width, hidden_states.device)
        cos, sin = a0.rotary_emb._tables(hidden_states.device, hidden_states.dtype)
        got = hip_ops.decode_layers_persistent(
            st["table"], len(self.layers), hidden_states, positions, ctx.slot_mapping, ctx.block_tables, ctx.context_lens,
            cos, sin, kc, a0.num_heads, a0.num_kv_heads, a0.head_dim, self.layers[0].mlp.intermediate_size,
            a0.rms_norm_eps, a0.attn.scale, ws, st["status"], PERSISTENT_FLAGS)
        if got is None:
            st["declined"] = True           # a shape the kernel does not take: do not ask again
            return None
        st["status_host"].copy_(st["status"], non_blocking=True)   # read by the executor once the step's tokens are back
        return got

    def persistent_give_ups(self) -> int:
        """Bounded barrier waits of the persistent decode launch that gave up (0 = fine).  Valid once the stream work of the
        step has completed (the executor asks after it has read the step's tokens)."""
        st = getattr(self, "_persist", None)
        return 0 if st is None else int(st["status_host"][0])

    def persistent_status(self) -> list[int]:
        """The four status words of the persistent launch: give-ups, then the first expired wait: (barrier epoch << 16 |
        workgroup; 0xffff.... = the start-up census), the value waited for, the value last seen."""
        st = getattr(self, "_persist", None)
        return [0, 0, 0, 0] if st is None else [int(v) & 0xFFFFFFFF for v in st["status_host"][:4]]

    def persistent_stamps(self) -> list[float]:
        """(MV_PERSISTENT_PREFETCH flags bit 8) microseconds between the phase edges of layer 1 as workgroup 1 saw them."""
        st = getattr(self, "_persist", None)
        if st is None:
            return []
        t = [int(v) & 0xFFFFFFFF for v in st["status_host"][8:23]]
        return [((b - a) & 0xFFFFFFFF) / 100.0 for a, b in zip(t, t[1:])]

    def forward(self, input_ids, positions):
        hidden_states, residual, pending = self.embed_tokens(input_ids), None, False
        fused = self._forward_persistent(hidden_states, positions)
        if fused is not None:
            hidden_states, _ = self.norm(fused[0], fused[1])
            return hidden_states
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, residual, pending)
            pending = layer.defers_reduce
        if pending:   # the last layer's down_proj is still un-reduced
            hidden_states, _ = tp_all_reduce_add_rmsnorm(hidden_states, residual, self.norm.weight, self.norm.eps)
        else:
            hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states


class QwenForCausalLM(nn.Module):
    model_cls = QwenModel
    # checkpoint name fragment -> (fused parameter, shard id)   (qwen_base.py:330-336)
    packed_modules_mapping = {"q_proj": ("qkv_proj", "q"), "k_proj": ("qkv_proj", "k"), "v_proj": ("qkv_proj", "v"),
                              "gate_proj": ("gate_up_proj", 0), "up_proj": ("gate_up_proj", 1)}

    def __init__(self, config) -> None:
        super().__init__()
        self.config = config
        self.model = self.model_cls(config)
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if getattr(config, "tie_word_embeddings", False):
            self.lm_head.weight = self.model.embed_tokens.weight

    def forward(self, input_ids, positions):
        return self.model(input_ids, positions)

    def compute_logits(self, hidden_states, gather: bool = True):
        return self.lm_head(hidden_states, gather=gather)

    def set_kv_cache(self, kv_cache) -> None:
        for layer, (k, v) in zip(self.model.layers, kv_cache):
            layer.self_attn.attn.k_cache, layer.self_attn.attn.v_cache = k, v

    def load_weights(self, weights: dict[str, Any]) -> None:
        from ..utils.loader import load_weight

        for name, tensor in weights.items():
            if "rotary_emb.inv_freq" not in name:
                load_weight(self, name, tensor, self.packed_modules_mapping)


class Qwen2Attention(QwenAttention):
    default_qkv_bias = True
    default_rope_theta = 1000000


class Qwen3Attention(QwenAttention):
    default_qkv_bias = False
    default_rope_theta = 10000


def _family(attention):
    layer = type(f"{attention.__name__[:5]}DecoderLayer", (QwenDecoderLayer,), {"attention_cls": attention})
    model = type(f"{attention.__name__[:5]}Model", (QwenModel,), {"decoder_layer_cls": layer})
    return type(f"{attention.__name__[:5]}ForCausalLM", (QwenForCausalLM,), {"model_cls": model})


Qwen2ForCausalLM = _family(Qwen2Attention)
Qwen3ForCausalLM = _family(Qwen3Attention)
