"""Pins the CPU oracle (oracle/ops.py) to the reference: every array in tests/golden/ops.npz
was produced by running jianzhnie/mini-vllm itself on the CPU (tools/gen_goldens.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import ops as O


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "ops.npz"))


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_paged_decode_matches_reference(G, ci):
    p = f"decode{ci}_"
    out = O.paged_decode(t(G[p + "q"]), t(G[p + "k_cache"]), t(G[p + "v_cache"]), t(G[p + "block_tables"]),
                         t(G[p + "context_lens"]), float(G[p + "scale"]))
    # the reference's own unit-test tolerance, tests/test_page_attention.py:120
    torch.testing.assert_close(out, t(G[p + "out"]), rtol=1e-4, atol=1e-5)
    # bf16 inputs: fp32 oracle on the rounded inputs vs the reference's bf16 eager result
    bf = lambda a: t(a).bfloat16()
    out16 = O.paged_decode(bf(G[p + "q"]), bf(G[p + "k_cache"]), bf(G[p + "v_cache"]), t(G[p + "block_tables"]),
                           t(G[p + "context_lens"]), float(G[p + "scale"]))
    torch.testing.assert_close(out16, t(G[p + "out_bf16"]), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("ci", [0, 1])
def test_prefill_matches_reference(G, ci):
    p = f"prefill{ci}_"
    cu = t(G[p + "cu"])
    out = O.prefill_varlen(t(G[p + "q"]), t(G[p + "k"]), t(G[p + "v"]), cu, cu, float(G[p + "scale"]))
    torch.testing.assert_close(out, t(G[p + "out"]), rtol=1e-4, atol=1e-5)
    bf = lambda a: t(a).bfloat16()
    out16 = O.prefill_varlen(bf(G[p + "q"]), bf(G[p + "k"]), bf(G[p + "v"]), cu, cu, float(G[p + "scale"]))
    torch.testing.assert_close(out16, t(G[p + "out_bf16"]), rtol=3e-2, atol=3e-2)


def test_store_kvcache_matches_reference(G):
    n, Hq, Hkv, D, nb, bs = [int(x) for x in G["store_dims"]]
    qkv = t(G["store_qkv"])
    k = qkv[:, Hq * D:(Hq + Hkv) * D].view(n, Hkv, D)
    v = qkv[:, (Hq + Hkv) * D:].view(n, Hkv, D)
    kc, vc = t(G["store_kc_in"]).clone(), t(G["store_vc_in"]).clone()
    O.store_kvcache(k, v, kc, vc, t(G["store_slots"]))
    assert torch.equal(kc, t(G["store_kc_out"])) and torch.equal(vc, t(G["store_vc_out"]))


@pytest.mark.parametrize("cols", [128, 448])
@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_rmsnorm_matches_reference(G, cols, tag):
    dt = torch.float32 if tag == "f32" else torch.bfloat16
    p = f"rms{cols}_{tag}_"
    x, r, w = t(G[p + "x"]).to(dt), t(G[p + "r"]).to(dt), t(G[p + "w"]).to(dt)
    y = O.rmsnorm(x, w, 1e-6)
    y2, res = O.add_rmsnorm(x, r, w, 1e-6)
    # same ops in the same order on the same machine: bit-exact
    assert torch.equal(y.float(), t(G[p + "y"]))
    assert torch.equal(y2.float(), t(G[p + "y_add"]))
    assert torch.equal(res.float(), t(G[p + "res"]))


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_rope_matches_reference(G, ci):
    p = f"rope{ci}_"
    D, max_pos, base, kind, factor = G[p + "cfg"]
    scaling = None if kind == 0 else {"type": "linear" if kind == 1 else "dynamic", "factor": float(factor)}
    cos, sin = O.rope_tables(int(D), int(max_pos), float(base), scaling)
    rows = t(G[p + "rows"])
    assert torch.equal(cos[rows], t(G[p + "cos_rows"])) and torch.equal(sin[rows], t(G[p + "sin_rows"]))
    assert cos.shape == (int(max_pos), int(D) // 2)   # tests/test_rotary_embedding.py:27-40
    assert torch.all(cos[0] == 1) and torch.all(sin[0] == 0)  # :161-173
    pos, q, k = t(G[p + "pos"]), t(G[p + "q"]), t(G[p + "k"])
    qo, ko = O.rope_qk_store(q, k, None, pos, cos, sin, None, None, 1e-6)
    assert torch.equal(qo, t(G[p + "q_out"])) and torch.equal(ko, t(G[p + "k_out"]))
    qo16, ko16 = O.rope_qk_store(q.bfloat16(), k.bfloat16(), None, pos, cos, sin, None, None, 1e-6)
    assert torch.equal(qo16.float(), t(G[p + "q_out_bf16"])) and torch.equal(ko16.float(), t(G[p + "k_out_bf16"]))


def test_silu_matches_reference(G):
    x = t(G["silu_x"])
    assert torch.equal(O.silu_and_mul(x), t(G["silu_y"]))
    assert torch.equal(O.silu_and_mul(x.bfloat16()).float(), t(G["silu_y_bf16"]))


def test_sampler_filters_match_reference(G):
    logits, temps = t(G["samp_logits"]), t(G["samp_temps"])
    top_ks, top_ps, min_ps = t(G["samp_top_ks"]), t(G["samp_top_ps"]), t(G["samp_min_ps"])
    x = O.apply_temperature(logits, temps)
    torch.testing.assert_close(x, t(G["samp_after_temp"]), rtol=1e-6, atol=0)
    keep_k = O.top_k_mask(x, top_ks)
    assert torch.equal(keep_k, t(G["samp_keep_topk"]))
    x = x.masked_fill(~keep_k, float("-inf"))
    keep_p = keep_k & O.top_p_mask(x, top_ps)
    assert torch.equal(keep_p, t(G["samp_keep_topp"]))
    x = x.masked_fill(~keep_p, float("-inf"))
    keep_m = keep_p & O.min_p_mask(x, min_ps)
    assert torch.equal(keep_m, t(G["samp_keep_minp"]))
    keep, _ = O.sampler_keep_mask(logits, temps, top_ks, top_ps, min_ps)
    assert torch.equal(keep, t(G["samp_keep_minp"]))


# ---------------------------------------------------------------- SURVEY §8(c) wide cases (tests/golden/ops_wide.npz)
@pytest.fixture(scope="module")
def GW(golden_dir):
    return np.load(os.path.join(golden_dir, "ops_wide.npz"))


def _wide_cases(kind):
    import golden_inputs as GI

    return [c["name"] for c in (GI.WIDE_DECODE if kind == "decode" else GI.WIDE_PREFILL)]


@pytest.mark.parametrize("name", _wide_cases("decode"))
def test_paged_decode_wide_matches_reference(GW, name):
    """Head layouts 16/8, 28/4, 8/1, 12/12 and contexts 1..4096: inputs regenerated from their seed (checksum
    pinned in the fixture), output against what the reference's PageAttention produced."""
    import golden_inputs as GI

    case = next(c for c in GI.WIDE_DECODE if c["name"] == name)
    q, kc, vc, table, ctx = GI.wide_decode_inputs(case)
    assert GI.checksum(q, kc, vc, table, ctx) == float(GW[f"decode_{name}_checksum"]), "seeded inputs drifted"
    out = O.paged_decode(q.float(), kc.float(), vc.float(), table, ctx, case["D"] ** -0.5)
    torch.testing.assert_close(out, t(GW[f"decode_{name}_out"]), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", _wide_cases("prefill"))
def test_prefill_wide_matches_reference(GW, name):
    import golden_inputs as GI

    case = next(c for c in GI.WIDE_PREFILL if c["name"] == name)
    q, k, v, cu = GI.wide_prefill_inputs(case)
    assert GI.checksum(q, k, v, cu) == float(GW[f"prefill_{name}_checksum"]), "seeded inputs drifted"
    out = O.prefill_varlen(q.float(), k.float(), v.float(), cu, cu, case["D"] ** -0.5)
    torch.testing.assert_close(out[GI.wide_prefill_rows(case)], t(GW[f"prefill_{name}_out_rows"]), rtol=1e-4, atol=2e-5)
