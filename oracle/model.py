"""CPU restatement of the reference's eager model path (TEST INFRASTRUCTURE — see oracle/ops.py).

Plain-torch Qwen2 / Qwen3 / OPT forward passes that follow the reference's op order and
rounding points (minivllm/models/qwen_base.py:173-190,281-298,317-357; models/opt.py:84-279),
built from the operators in oracle/ops.py and a paged KV cache indexed exactly like the
reference's (slot = block*block_size + offset).  Pinned to the reference by the logits in
tests/golden/e2e_*.npz (tests/test_oracle_model.py).  Used as
  * the on-box checker for end-to-end GPU parity,
  * the runner behind the CPU plumbing config (OPT-125M, reference examples/cpu_inference_opt.py),
  * bench.py's ``cpu_baseline`` ("port").
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import ops as O


def _cfg(d: dict) -> SimpleNamespace:
    c = SimpleNamespace(**d)
    if not hasattr(c, "num_key_value_heads"):
        c.num_key_value_heads = c.num_attention_heads
    if getattr(c, "head_dim", None) is None:
        c.head_dim = c.hidden_size // c.num_attention_heads
    return c


class OracleModel:
    """weights: HF-named tensors (q_proj/k_proj/v_proj, gate_proj/up_proj separate)."""

    def __init__(self, config: dict, weights: dict[str, torch.Tensor], dtype=torch.float32,
                 num_blocks: int = 16, block_size: int = 64):
        self.c = c = _cfg(config)
        self.kind = c.model_type
        self.dtype = dtype
        self.w = {k: v.to(dtype) for k, v in weights.items()}
        self.block_size = block_size
        L = c.num_hidden_layers
        self.k_cache = [torch.zeros(num_blocks, block_size, c.num_key_value_heads, c.head_dim, dtype=dtype)
                        for _ in range(L)]
        self.v_cache = [torch.zeros_like(k) for k in self.k_cache]
        if self.kind in ("qwen2", "qwen3"):
            default_theta = 1e6 if self.kind == "qwen2" else 1e4
            theta = getattr(c, "rope_theta", None)
            if theta is None:
                theta = (getattr(c, "rope_parameters", None) or {}).get("rope_theta", default_theta)
            cos, sin = O.rope_tables(c.head_dim, c.max_position_embeddings, theta, getattr(c, "rope_scaling", None))
            # the reference's model.to(dtype) rounds the tables to the model dtype
            self.cos, self.sin = cos.to(dtype).float(), sin.to(dtype).float()
            self.qk_norm = not getattr(c, "attention_bias", self.kind == "qwen2")

    @classmethod
    def random(cls, config: dict, dtype=torch.float32, seed: int = 0, **kw):
        """Random weights at the architecture's true dimensions (throughput is value-independent)."""
        c = _cfg(config)
        g = torch.Generator().manual_seed(seed)
        h, hd, H, Hkv = c.hidden_size, c.head_dim, c.num_attention_heads, c.num_key_value_heads
        w = {}
        # torch.randn manages ~20 M values/s on a host core: 1.7 B parameters would take minutes.
        # Tile a 4 Mi-value random block into each (distinct) weight instead.
        base = (torch.randn(1 << 22, generator=g) * 0.02).to(dtype)

        def rnd(out_f, in_f):
            n = out_f * in_f
            return base.repeat(-(-n // base.numel()))[:n].view(out_f, in_f)

        def lin(name, out_f, in_f, bias=False):
            w[name + ".weight"] = rnd(out_f, in_f)
            if bias:
                w[name + ".bias"] = torch.zeros(out_f, dtype=dtype)
        if c.model_type == "opt":          # HF OPT names (minivllm/models/opt.py:84-279): biased projections, LayerNorm
            pre = "model.decoder."
            w[pre + "embed_tokens.weight"] = rnd(c.vocab_size, getattr(c, "word_embed_proj_dim", h))
            w[pre + "embed_positions.weight"] = rnd(c.max_position_embeddings + 2, h)
            if getattr(c, "word_embed_proj_dim", h) != h:
                lin(pre + "project_in", h, c.word_embed_proj_dim)
                lin(pre + "project_out", c.word_embed_proj_dim, h)

            def norm(name):
                w[name + ".weight"] = torch.ones(h, dtype=dtype)
                w[name + ".bias"] = torch.zeros(h, dtype=dtype)
            for i in range(c.num_hidden_layers):
                p = f"{pre}layers.{i}."
                for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                    lin(p + "self_attn." + n, h, h, bias=True)
                lin(p + "fc1", c.ffn_dim, h, bias=True)
                lin(p + "fc2", h, c.ffn_dim, bias=True)
                norm(p + "self_attn_layer_norm")
                norm(p + "final_layer_norm")
            if getattr(c, "do_layer_norm_before", True):
                norm(pre + "final_layer_norm")
            return cls(config, w, dtype=dtype, **kw)
        assert c.model_type in ("qwen2", "qwen3")
        bias = getattr(c, "attention_bias", c.model_type == "qwen2")
        w["model.embed_tokens.weight"] = rnd(c.vocab_size, h)
        for i in range(c.num_hidden_layers):
            p = f"model.layers.{i}."
            lin(p + "self_attn.q_proj", H * hd, h, bias)
            lin(p + "self_attn.k_proj", Hkv * hd, h, bias)
            lin(p + "self_attn.v_proj", Hkv * hd, h, bias)
            lin(p + "self_attn.o_proj", h, H * hd)
            lin(p + "mlp.gate_proj", c.intermediate_size, h)
            lin(p + "mlp.up_proj", c.intermediate_size, h)
            lin(p + "mlp.down_proj", h, c.intermediate_size)
            for n in ("input_layernorm", "post_attention_layernorm"):
                w[p + n + ".weight"] = torch.ones(h, dtype=dtype)
            if not bias:
                w[p + "self_attn.q_norm.weight"] = torch.ones(hd, dtype=dtype)
                w[p + "self_attn.k_norm.weight"] = torch.ones(hd, dtype=dtype)
        w["model.norm.weight"] = torch.ones(h, dtype=dtype)
        if not getattr(c, "tie_word_embeddings", False):
            lin("lm_head", c.vocab_size, h)
        return cls(config, w, dtype=dtype, **kw)

    # ------------------------------------------------------------------ attention core
    def _attend(self, layer, q, k, v, meta):
        """store_kv_cache, then prefill (contiguous, or from the cache on a prefix hit) or paged decode."""
        c = self.c
        scale = c.head_dim ** -0.5
        if meta["slot_mapping"] is not None:
            O.store_kvcache(k, v, self.k_cache[layer], self.v_cache[layer], meta["slot_mapping"])
        if meta["is_prefill"]:
            cu_q, cu_k = meta["cu_q"], meta["cu_k"]
            if meta.get("prefix_cached"):
                lens = (cu_k[1:] - cu_k[:-1]).tolist()
                k, v = O.gather_paged_kv(self.k_cache[layer], self.v_cache[layer], meta["block_tables"], lens)
            return O.prefill_varlen(q, k, v, cu_q, cu_k, scale).to(self.dtype)
        return O.paged_decode(q, self.k_cache[layer], self.v_cache[layer], meta["block_tables"],
                              meta["context_lens"], scale).to(self.dtype)

    # ------------------------------------------------------------------ Qwen
    def _qwen_forward(self, ids, pos, meta):
        c, w = self.c, self.w
        H, Hkv, D, eps = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.rms_norm_eps
        x = F.embedding(ids, w["model.embed_tokens.weight"])
        res = None
        for i in range(c.num_hidden_layers):
            p = f"model.layers.{i}."
            if res is None:
                res, x = x, O.rmsnorm(x, w[p + "input_layernorm.weight"], eps)
            else:
                x, res = O.add_rmsnorm(x, res, w[p + "input_layernorm.weight"], eps)
            t = x.shape[0]
            q = F.linear(x, w[p + "self_attn.q_proj.weight"], w.get(p + "self_attn.q_proj.bias")).view(t, H, D)
            k = F.linear(x, w[p + "self_attn.k_proj.weight"], w.get(p + "self_attn.k_proj.bias")).view(t, Hkv, D)
            v = F.linear(x, w[p + "self_attn.v_proj.weight"], w.get(p + "self_attn.v_proj.bias")).view(t, Hkv, D)
            qn = w.get(p + "self_attn.q_norm.weight") if self.qk_norm else None
            kn = w.get(p + "self_attn.k_norm.weight") if self.qk_norm else None
            q, k = O.rope_qk_store(q, k, v, pos, self.cos, self.sin, qn, kn, eps)
            o = self._attend(i, q, k, v, meta)
            x = F.linear(o.reshape(t, H * D), w[p + "self_attn.o_proj.weight"])
            x, res = O.add_rmsnorm(x, res, w[p + "post_attention_layernorm.weight"], eps)
            gate_up = torch.cat([F.linear(x, w[p + "mlp.gate_proj.weight"]), F.linear(x, w[p + "mlp.up_proj.weight"])], -1)
            x = F.linear(O.silu_and_mul(gate_up), w[p + "mlp.down_proj.weight"])
        x, _ = O.add_rmsnorm(x, res, w["model.norm.weight"], eps)
        return x

    # ------------------------------------------------------------------ OPT
    def _opt_forward(self, ids, pos, meta):
        c, w = self.c, self.w
        H, D, h = c.num_attention_heads, c.head_dim, c.hidden_size
        pre = "model.decoder."
        x = F.embedding(ids, w[pre + "embed_tokens.weight"])
        if pre + "project_in.weight" in w:
            x = F.linear(x, w[pre + "project_in.weight"])
        x = x + F.embedding(pos + 2, w[pre + "embed_positions.weight"])

        def ln(t, name):
            return F.layer_norm(t, (h,), w[name + ".weight"], w[name + ".bias"])
        for i in range(c.num_hidden_layers):
            p = f"{pre}layers.{i}."

            def attn(t_in, p=p, i=i):
                t = t_in.shape[0]
                q = F.linear(t_in, w[p + "self_attn.q_proj.weight"], w.get(p + "self_attn.q_proj.bias")).view(t, H, D)
                k = F.linear(t_in, w[p + "self_attn.k_proj.weight"], w.get(p + "self_attn.k_proj.bias")).view(t, H, D)
                v = F.linear(t_in, w[p + "self_attn.v_proj.weight"], w.get(p + "self_attn.v_proj.bias")).view(t, H, D)
                o = self._attend(i, q, k, v, meta)
                return F.linear(o.reshape(t, H * D), w[p + "self_attn.out_proj.weight"], w.get(p + "self_attn.out_proj.bias"))

            def mlp(t_in, p=p):
                return F.linear(F.relu(F.linear(t_in, w[p + "fc1.weight"], w.get(p + "fc1.bias"))),
                                w[p + "fc2.weight"], w.get(p + "fc2.bias"))
            if c.do_layer_norm_before:
                x = x + attn(ln(x, p + "self_attn_layer_norm"))
                x = x + mlp(ln(x, p + "final_layer_norm"))
            else:
                x = ln(x + attn(x), p + "self_attn_layer_norm")
                x = ln(x + mlp(x), p + "final_layer_norm")
        if c.do_layer_norm_before:
            x = ln(x, pre + "final_layer_norm")
        if pre + "project_out.weight" in w:
            x = F.linear(x, w[pre + "project_out.weight"])
        return x

    # ------------------------------------------------------------------ public
    @torch.inference_mode()
    def forward_logits(self, input_ids, positions, meta):
        """meta: is_prefill, slot_mapping, block_tables, and cu_q/cu_k (prefill) or context_lens (decode)."""
        hidden = (self._opt_forward if self.kind == "opt" else self._qwen_forward)(input_ids, positions, meta)
        if meta["is_prefill"]:
            hidden = hidden[(meta["cu_q"][1:] - 1).long()]
        head = self.w.get("lm_head.weight")
        if head is None:
            head = self.w["model.decoder.embed_tokens.weight" if self.kind == "opt" else "model.embed_tokens.weight"]
        return F.linear(hidden, head)


class OracleRunner:
    """Duck-typed stand-in for the build's ModelRunner so that the HOST engine (LLMEngine /
    Scheduler / InputBuilder) can be exercised end to end on the CPU in tests:
    ``LLMEngine(config, model_runner=OracleRunner(...))``."""

    def __init__(self, model: OracleModel, builder, seed: int = 0):
        self.model = model
        self.builder = builder   # the build's InputBuilder instance (host logic under test)
        self.gen = torch.Generator().manual_seed(seed)
        self.model_manager = SimpleNamespace(tokenizer=None)

    def call(self, name, *args):
        return getattr(self, name)(*args)

    def exit(self):
        pass

    def run(self, sequences, is_prefill):
        inp = self.builder.prefill(sequences) if is_prefill else self.builder.decode(sequences)
        meta = dict(is_prefill=is_prefill, slot_mapping=inp.slot_mapping, block_tables=inp.block_tables,
                    cu_q=inp.cum_seqlens_q, cu_k=inp.cum_seqlens_k, context_lens=inp.context_lens,
                    prefix_cached=is_prefill and any(s.num_cached_tokens > 0 for s in sequences))
        logits = self.model.forward_logits(inp.input_ids.clone(), inp.positions.clone(), meta).float()
        t = torch.tensor([s.temperature for s in sequences])
        probs = O.sampler_probs(logits, t, torch.tensor([s.top_k for s in sequences]),
                                torch.tensor([s.top_p for s in sequences]), torch.tensor([s.min_p for s in sequences]))
        return torch.multinomial(probs, 1, generator=self.gen).squeeze(1).tolist()
