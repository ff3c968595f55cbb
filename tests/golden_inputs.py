"""Seeded input generators shared by the fixture generator (tools/gen_goldens.py, which feeds them to the
REFERENCE in the build container) and by the tests (which feed the same numbers to the HIP kernels / the oracle).
Large inputs are regenerated from their seed instead of being stored; every fixture carries a float64 checksum of
what the generator produced so that a drift of the random stream is noticed, not mistaken for a parity failure.

Data only — no reference code lives here."""
from __future__ import annotations

import zlib

import numpy as np
import torch


def checksum(*tensors) -> float:
    """Order-dependent float64 fingerprint of the given tensors."""
    total = 0.0
    for i, t in enumerate(tensors):
        a = t.detach().double().flatten()
        w = torch.arange(1, a.numel() + 1, dtype=torch.float64) % 97 + 1.0
        total += float((a * w).sum()) * (i + 1)
    return total


def make_paged(B, Hkv, D, bs, ctxs, seed, dtype=torch.float32, extra_blocks=2):
    """Random paged K/V caches + a shuffled (non-contiguous) block table, -1 padded (one spare column)."""
    g = torch.Generator().manual_seed(seed)
    nblk = [(c + bs - 1) // bs for c in ctxs]
    total = sum(nblk) + extra_blocks
    perm = torch.randperm(total, generator=g).tolist()
    width = max(nblk) + 1
    table = torch.full((B, width), -1, dtype=torch.int32)
    it = iter(perm)
    for b in range(B):
        for j in range(nblk[b]):
            table[b, j] = next(it)
    kc = torch.randn(total, bs, Hkv, D, generator=g).to(dtype)
    vc = torch.randn(total, bs, Hkv, D, generator=g).to(dtype)
    return kc, vc, table


# decode cases of SURVEY §8(c): head layouts of Qwen3 (16/8), Qwen2-7B (28/4), Qwen2-72B@TP8 (8/1), OPT-125M (12/12,
# D = 64); context lengths 1 .. 4096 including block edges; inputs bf16-representable so that the fp32 reference
# output serves 16-bit and fp32 kernels alike
WIDE_DECODE = [
    dict(name="qwen3_16_8", H=16, Hkv=8, D=128, bs=64, ctxs=[4096, 1, 1000, 2049, 64, 65], seed=1101),
    dict(name="qwen2_28_4", H=28, Hkv=4, D=128, bs=64, ctxs=[3000, 4096, 63, 129], seed=1102),
    dict(name="q72b_tp8_8_1", H=8, Hkv=1, D=128, bs=64, ctxs=[4096, 2048, 7, 640, 1025], seed=1103),
    dict(name="opt_12_12", H=12, Hkv=12, D=64, bs=64, ctxs=[1024, 33, 2], seed=1104),
]


def wide_decode_inputs(case):
    kc, vc, table = make_paged(len(case["ctxs"]), case["Hkv"], case["D"], case["bs"], case["ctxs"], case["seed"],
                               dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(case["seed"] + 7)
    q = torch.randn(len(case["ctxs"]), case["H"], case["D"], generator=g).bfloat16()
    ctx = torch.tensor(case["ctxs"], dtype=torch.int32)
    return q, kc, vc, table, ctx


# prefill cases: packed varlen causal attention; the same outputs pin the prefix-cached form (the rows of the new
# tokens of a FULL no-cache prefill are what a prefix-cached prefill must produce)
WIDE_PREFILL = [
    dict(name="qwen3_16_8", H=16, Hkv=8, D=128, lens=[300, 77, 1024, 1], cached=[256, 64, 960, 0], seed=1201),
    dict(name="qwen2_28_4", H=28, Hkv=4, D=128, lens=[640, 129], cached=[576, 128], seed=1202),
    dict(name="opt_12_12", H=12, Hkv=12, D=64, lens=[200, 65], cached=[128, 64], seed=1203),
]


def wide_prefill_inputs(case):
    g = torch.Generator().manual_seed(case["seed"])
    T = sum(case["lens"])
    q = torch.randn(T, case["H"], case["D"], generator=g).bfloat16()
    k = torch.randn(T, case["Hkv"], case["D"], generator=g).bfloat16()
    v = torch.randn(T, case["Hkv"], case["D"], generator=g).bfloat16()
    cu = torch.tensor([0] + np.cumsum(case["lens"]).tolist(), dtype=torch.int32)
    return q, k, v, cu


def wide_prefill_rows(case):
    """Rows of the packed [T, H, D] output kept in the fixture (the full tensor would be tens of MB): the first 8
    tokens of every sequence and every token behind its `cached` prefix (the rows a prefix-cached prefill computes)."""
    rows, o = [], 0
    for n, c in zip(case["lens"], case["cached"]):
        rows += list(range(o, o + min(8, n))) + list(range(o + c, o + n))
        o += n
    return torch.tensor(sorted(set(rows)), dtype=torch.int64)


def seeded_hf_weights(cfg: dict, seed: int = 1):
    """HF-checkpoint-style weights of a Qwen2/Qwen3 config, every tensor drawn from its own generator keyed by its
    NAME (so that the packed layouts of the reference and of the build both load the same numbers), rounded to
    bf16: normal(0, fan_in^-0.5) matrices, norm weights 1 + 0.1 normal, biases 0.05 normal."""
    hs, L = cfg["hidden_size"], cfg["num_hidden_layers"]
    H, Hkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    D = cfg.get("head_dim", hs // H)
    inter, V = cfg["intermediate_size"], cfg["vocab_size"]
    qk_norm = cfg["model_type"] == "qwen3"
    bias = cfg.get("attention_bias", cfg["model_type"] == "qwen2")

    def draw(name, shape, kind):
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + seed * 1000003) & 0x7FFFFFFF)
        t = torch.randn(shape, generator=g)
        if kind == "matrix":
            t = t * shape[-1] ** -0.5
        elif kind == "norm":
            t = 1 + 0.1 * t
        else:
            t = 0.05 * t
        return t.bfloat16()

    w = {"model.embed_tokens.weight": draw("model.embed_tokens.weight", (V, hs), "matrix"),
         "model.norm.weight": draw("model.norm.weight", (hs,), "norm")}
    if not cfg.get("tie_word_embeddings", False):
        w["lm_head.weight"] = draw("lm_head.weight", (V, hs), "matrix")
    for i in range(L):
        p = f"model.layers.{i}."
        for nm, rows in (("q_proj", H * D), ("k_proj", Hkv * D), ("v_proj", Hkv * D)):
            w[p + f"self_attn.{nm}.weight"] = draw(p + f"self_attn.{nm}.weight", (rows, hs), "matrix")
            if bias:
                w[p + f"self_attn.{nm}.bias"] = draw(p + f"self_attn.{nm}.bias", (rows,), "bias")
        w[p + "self_attn.o_proj.weight"] = draw(p + "self_attn.o_proj.weight", (hs, H * D), "matrix")
        if qk_norm:
            w[p + "self_attn.q_norm.weight"] = draw(p + "self_attn.q_norm.weight", (D,), "norm")
            w[p + "self_attn.k_norm.weight"] = draw(p + "self_attn.k_norm.weight", (D,), "norm")
        w[p + "mlp.gate_proj.weight"] = draw(p + "mlp.gate_proj.weight", (inter, hs), "matrix")
        w[p + "mlp.up_proj.weight"] = draw(p + "mlp.up_proj.weight", (inter, hs), "matrix")
        w[p + "mlp.down_proj.weight"] = draw(p + "mlp.down_proj.weight", (hs, inter), "matrix")
        w[p + "input_layernorm.weight"] = draw(p + "input_layernorm.weight", (hs,), "norm")
        w[p + "post_attention_layernorm.weight"] = draw(p + "post_attention_layernorm.weight", (hs,), "norm")
    return w


# BASELINE config 2 at its TRUE dimensions (public HF config.json of Qwen3-0.6B), all 28 layers
CONFIG2 = dict(model_type="qwen3", architectures=["Qwen3ForCausalLM"], hidden_size=1024, num_hidden_layers=28,
               num_attention_heads=16, num_key_value_heads=8, head_dim=128, intermediate_size=3072,
               vocab_size=151936, max_position_embeddings=40960, rms_norm_eps=1e-6, rope_theta=1000000.0,
               tie_word_embeddings=True, hidden_act="silu", attention_bias=False, torch_dtype="bfloat16",
               eos_token_id=151645)
CONFIG2_PROMPT_LENS = (70, 33)
CONFIG2_DECODE_STEPS = 3
CONFIG2_SAMPLED_COLUMNS = 512


def config2_prompts():
    g = torch.Generator().manual_seed(0)
    return [torch.randint(0, CONFIG2["vocab_size"], (n,), generator=g).tolist() for n in CONFIG2_PROMPT_LENS]


def config2_columns():
    g = torch.Generator().manual_seed(2)
    return torch.randperm(CONFIG2["vocab_size"], generator=g)[:CONFIG2_SAMPLED_COLUMNS].sort().values
