"""GPU parity: every libmvhip.so kernel, called through the C-ABI (ctypes), against the
CPU oracle on the same seeded inputs.  Integer/byte work is bit-exact; floating point is
held to the tolerance written next to each check."""
import math

import os

import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def H():
    from minivllm_amd import hip_ops

    return hip_ops


def _randn(*shape, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32).to(dtype)


def _tol(dtype):
    # fp32: the reference's own unit-test tolerance (tests/test_page_attention.py:120)
    # bf16/f16: 2 ulp of the output scale (outputs are O(1))
    return {torch.float32: dict(rtol=1e-4, atol=1e-5),
            torch.bfloat16: dict(rtol=2 ** -7, atol=2 ** -7),
            torch.float16: dict(rtol=2 ** -10, atol=2 ** -9)}[dtype]


def _make_paged(B, Hkv, D, bs, ctxs, dtype, seed, nb_extra=3, pad_cols=2):
    """Random caches + a shuffled (non-contiguous) block table, -1 padded."""
    g = torch.Generator().manual_seed(seed)
    nblk = [(c + bs - 1) // bs for c in ctxs]
    total = sum(nblk) + nb_extra
    perm = torch.randperm(total, generator=g).tolist()
    width = max(nblk + [1]) + pad_cols
    table = torch.full((B, width), -1, dtype=torch.int32)
    it = iter(perm)
    for b in range(B):
        for j in range(nblk[b]):
            table[b, j] = next(it)
    kc = torch.randn(total, bs, Hkv, D, generator=g).to(dtype)
    vc = torch.randn(total, bs, Hkv, D, generator=g).to(dtype)
    return kc, vc, table


# ------------------------------------------------------------------ K1
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_store_kvcache_bit_exact(H, dtype):
    n, Hq, Hkv, D, nb, bs = 37, 4, 2, 128, 5, 64
    qkv = _randn(n, (Hq + 2 * Hkv) * D, dtype=dtype, seed=1)
    k = qkv[:, Hq * D:(Hq + Hkv) * D].view(n, Hkv, D)        # strided views, like the engine
    v = qkv[:, (Hq + Hkv) * D:].view(n, Hkv, D)
    slots = torch.randperm(nb * bs, generator=torch.Generator().manual_seed(2))[:n].to(torch.int32)
    slots[3] = -1
    slots[11] = -1
    kc0 = _randn(nb, bs, Hkv, D, dtype=dtype, seed=3)
    vc0 = _randn(nb, bs, Hkv, D, dtype=dtype, seed=4)
    kc_ref, vc_ref = kc0.clone(), vc0.clone()
    O.store_kvcache(k, v, kc_ref, vc_ref, slots)
    kc, vc = kc0.to(DEV), vc0.to(DEV)
    qkv_d = qkv.to(DEV)
    kd = qkv_d[:, Hq * D:(Hq + Hkv) * D].view(n, Hkv, D)
    vd = qkv_d[:, (Hq + Hkv) * D:].view(n, Hkv, D)
    H.store_kvcache(kd, vd, kc, vc, slots.to(DEV))
    assert torch.equal(kc.cpu(), kc_ref)
    assert torch.equal(vc.cpu(), vc_ref)


def test_store_kvcache_int64_slots_and_size_check(H):
    k = torch.zeros(4, 2, 64, dtype=torch.bfloat16, device=DEV)
    kc = torch.zeros(2, 64, 2, 64, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    H.store_kvcache(k + 1, k + 2, kc, vc, torch.tensor([5, -1, 70, 127], device=DEV))  # int64 ok
    assert float(kc.view(-1, 128)[5].sum()) == 128.0 and float(vc.view(-1, 128)[127].sum()) == 256.0
    assert float(kc.view(-1, 128)[0].sum()) == 0.0
    with pytest.raises(ValueError):
        H.store_kvcache(k, k, kc, vc, torch.tensor([1, 2], device=DEV))


# ------------------------------------------------------------------ K3
DECODE_CASES = [
    # B, H, Hkv, D, bs, ctxs
    (3, 16, 8, 128, 64, [1, 64, 200]),          # Qwen3 head layout, boundary lengths
    (4, 28, 4, 128, 64, [130, 4, 777, 65]),     # Qwen2-7B: G = 7
    (2, 8, 1, 128, 64, [300, 1023]),            # 72B@TP8 per-GPU: G = 8
    (2, 12, 12, 64, 64, [90, 17]),              # OPT: MHA, D = 64
    (2, 8, 2, 128, 16, [33, 95]),               # small block size (reference unit tests)
    (2, 32, 2, 128, 64, [257, 31]),             # G = 16 -> two head chunks
    (1, 16, 8, 128, 64, [4096]),                # long context, split-KV
    (5, 4, 2, 64, 4, [1, 2, 3, 9, 40]),         # tiny blocks, generic (non-uniform) path
    (3, 16, 4, 128, 64, [31, 32, 33]),          # G = 4: smallest matrix-core group, tile edges
    (2, 40, 2, 128, 64, [500, 129]),            # G = 20: two head chunks, ragged second chunk
    (2, 28, 4, 128, 32, [100, 2000]),           # G = 7 with 32-token blocks, split-KV
    (3, 16, 8, 64, 4, [1, 9, 40]),              # whole-row mapping (8 kv heads of D = 64 per load), generic path
    (2, 24, 8, 128, 64, [70, 300]),             # whole-row mapping, G = 3
    (2, 8, 8, 128, 12, [5, 100]),               # whole-row mapping, MHA, block size not a multiple of the tile
]


@pytest.mark.parametrize("case", DECODE_CASES, ids=[f"case{i}" for i in range(len(DECODE_CASES))])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
def test_paged_decode_vs_oracle(H, case, dtype):
    B, Hq, Hkv, D, bs, ctxs = case
    kc, vc, table = _make_paged(B, Hkv, D, bs, ctxs, dtype, seed=10 + B)
    q = _randn(B, Hq, D, dtype=dtype, seed=99)
    ctx = torch.tensor(ctxs, dtype=torch.int32)
    scale = 1.0 / math.sqrt(D)
    ref = O.paged_decode(q, kc, vc, table, ctx, scale)
    out = H.paged_decode(q.to(DEV), kc.to(DEV), vc.to(DEV), table.to(DEV), ctx.to(DEV), scale)
    assert out.dtype == dtype and out.shape == (B, Hq, D)
    torch.testing.assert_close(out.float().cpu(), ref, **_tol(dtype))


def test_paged_decode_whole_row_and_per_head_mappings_agree(H):
    """mv_paged_decode picks the whole-row mapping (a workgroup owns 4 adjacent kv heads) when the head layout
    allows; MV_DECODE_PER_HEAD forces the one-kv-head-per-workgroup mapping.  Same numbers up to the order the
    partial softmax states are merged in, and both within the oracle's tolerance."""
    dtype = torch.bfloat16
    B, Hq, Hkv, D, bs = 6, 16, 8, 128, 64
    ctxs = [1, 63, 64, 65, 700, 2049]
    kc, vc, table = _make_paged(B, Hkv, D, bs, ctxs, dtype, seed=21)
    table[4, 3] = -1                       # a hole in the table contributes nothing (DESIGN Q9)
    q = _randn(B, Hq, D, dtype=dtype, seed=22)
    ctx = torch.tensor(ctxs, dtype=torch.int32)
    scale = D ** -0.5
    ref = O.paged_decode(q, kc, vc, table, ctx, scale)
    args = (q.to(DEV), kc.to(DEV), vc.to(DEV), table.to(DEV), ctx.to(DEV), scale)
    wide = H.paged_decode(*args, flags=0)
    per_head = H.paged_decode(*args, flags=H.MV_DECODE_PER_HEAD)
    torch.testing.assert_close(wide.float().cpu(), ref, **_tol(dtype))
    torch.testing.assert_close(per_head.float().cpu(), ref, **_tol(dtype))
    torch.testing.assert_close(wide.float(), per_head.float(), rtol=2 ** -7, atol=2e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("layout", [(16, 8, 128, True), (16, 8, 128, False), (6, 2, 128, True), (4, 4, 64, False),
                                    (12, 4, 128, True), (2, 1, 128, True),
                                    (28, 4, 128, False), (8, 1, 128, False), (32, 8, 128, False),     # matrix-core kernel
                                    (32, 8, 128, True), (8, 1, 128, True), (20, 4, 128, True)])       # ... with head norms
def test_rope_and_store_folded_into_paged_decode_are_bit_identical(H, dtype, layout):
    """mv_rope_paged_decode == mv_rope_qk_store followed by mv_paged_decode, bit for bit: the attention output AND the
    K/V cache afterwards (only the new token's rows may change), with and without the per-head norms, on raw q/k/v
    that are views of one qkv tensor, ragged contexts (split and unsplit), a graph-padding row (slot -1, context 0)."""
    Hq, Hkv, D, norms = layout
    bs = 64
    ctxs = [1, 64, 65, 300, 1000, 2049, 0]
    B = len(ctxs)
    kc, vc, table = _make_paged(B - 1, Hkv, D, bs, ctxs[:-1], dtype, seed=51)
    table = torch.cat([table, torch.full((1, table.shape[1]), -1, dtype=torch.int32)])
    qkv = _randn(B, (Hq + 2 * Hkv) * D, dtype=dtype, seed=52).to(DEV)
    q = qkv[:, :Hq * D].view(B, Hq, D)
    k = qkv[:, Hq * D:(Hq + Hkv) * D].view(B, Hkv, D)
    v = qkv[:, (Hq + Hkv) * D:].view(B, Hkv, D)
    ctx = torch.tensor(ctxs, dtype=torch.int32)
    pos = (ctx.long() - 1).clamp(min=0)
    slots = torch.tensor([int(table[b, (c - 1) // bs]) * bs + (c - 1) % bs if c > 0 else -1 for b, c in enumerate(ctxs)],
                         dtype=torch.int32)
    half = D // 2
    ang = torch.arange(4096, dtype=torch.float32)[:, None] * (10000.0 ** (-torch.arange(half, dtype=torch.float32) / half))[None]
    cos, sin = ang.cos().to(DEV), ang.sin().to(DEV)
    qn = (1.0 + 0.1 * _randn(D, dtype=torch.float32, seed=53)).to(dtype).to(DEV) if norms else None
    kn = (1.0 + 0.1 * _randn(D, dtype=torch.float32, seed=54)).to(dtype).to(DEV) if norms else None
    scale = D ** -0.5
    kc1, vc1, kc2, vc2 = kc.to(DEV), vc.to(DEV), kc.to(DEV), vc.to(DEV)
    q_rot, _ = H.rope_qk_store(q, k, v, pos.to(DEV), cos, sin, qn, kn, 1e-6, kc1, vc1, slots.to(DEV), want_k_out=False)
    want = H.paged_decode(q_rot, kc1, vc1, table.to(DEV), ctx.to(DEV), scale)
    got = H.rope_paged_decode(q, k, v, pos.to(DEV), cos, sin, qn, kn, 1e-6, slots.to(DEV), kc2, vc2, table.to(DEV),
                              ctx.to(DEV), scale)
    assert got is not None
    assert torch.equal(got, want)
    assert torch.equal(kc2, kc1) and torch.equal(vc2, vc1)
    assert not torch.equal(kc2.cpu(), kc)                       # the new rows were written
    # more than 16 query heads per kv head (two head chunks) is declined: None, nothing written
    big = _randn(2, (40 + 4) * 128, dtype=dtype, seed=55).to(DEV)
    none = H.rope_paged_decode(big[:, :40 * 128].view(2, 40, 128), big[:, 40 * 128:42 * 128].view(2, 2, 128),
                               big[:, 42 * 128:].view(2, 2, 128), pos[:2].to(DEV), cos[:, :64], sin[:, :64], None, None,
                               1e-6, slots[:2].to(DEV), *[_randn(40, 64, 2, 128, dtype=dtype, seed=1).to(DEV)] * 2,
                               table[:2].to(DEV), ctx[:2].to(DEV), scale)
    assert none is None


def test_paged_decode_edge_cases(H):
    dtype = torch.bfloat16
    B, Hq, Hkv, D, bs = 4, 16, 8, 128, 64
    kc, vc, table = _make_paged(B, Hkv, D, bs, [64, 10, 128, 5], dtype, seed=5)
    q = _randn(B, Hq, D, dtype=dtype, seed=6)
    scale = D ** -0.5
    # (a) ctx == 0 rows (graph-replay padding) give zeros and touch nothing
    ctx = torch.tensor([64, 0, 128, 0], dtype=torch.int32)
    out = H.paged_decode(q.to(DEV), kc.to(DEV), vc.to(DEV), table.to(DEV), ctx.to(DEV), scale)
    ref = O.paged_decode(q, kc, vc, table, ctx, scale)
    torch.testing.assert_close(out.float().cpu(), ref, **_tol(dtype))
    assert float(out[1].abs().sum()) == 0.0 and float(out[3].abs().sum()) == 0.0
    # (b) zero-column block table (warm-up decode): zeros, page_attention.py:62-65
    out0 = H.paged_decode(q.to(DEV), kc.to(DEV), vc.to(DEV),
                          torch.empty(B, 0, dtype=torch.int32, device=DEV), ctx.to(DEV), scale)
    assert float(out0.abs().sum()) == 0.0
    # (c) int64 tables / context_lens (reference unit tests pass int64) and a row-strided view
    big = torch.full((8, 16), -1, dtype=torch.int64)
    big[:B, :table.shape[1]] = table.long()
    view = big.to(DEV)[:B, :table.shape[1]]
    ctx2 = torch.tensor([64, 10, 128, 5], dtype=torch.int64)
    out2 = H.paged_decode(q.to(DEV), kc.to(DEV), vc.to(DEV), view, ctx2.to(DEV), scale)
    ref2 = O.paged_decode(q, kc, vc, table, ctx2, scale)
    torch.testing.assert_close(out2.float().cpu(), ref2, **_tol(dtype))


def test_paged_decode_large_batch_property(H):
    """BASELINE config-3 shape (B=256, ctx=1024): checked through a size-independent
    property — permuting the physical blocks (and the table with them) must not change a bit."""
    dtype = torch.bfloat16
    B, Hq, Hkv, D, bs, ctxlen = 256, 16, 8, 128, 64, 1024
    nblk = ctxlen // bs
    nb = B * nblk
    g = torch.Generator(device=DEV).manual_seed(0)
    kc = torch.randn(nb, bs, Hkv, D, generator=g, device=DEV, dtype=torch.float32).to(dtype)
    vc = torch.randn(nb, bs, Hkv, D, generator=g, device=DEV, dtype=torch.float32).to(dtype)
    q = torch.randn(B, Hq, D, generator=g, device=DEV, dtype=torch.float32).to(dtype)
    table = torch.arange(nb, dtype=torch.int32, device=DEV).view(B, nblk)
    ctx = torch.full((B,), ctxlen, dtype=torch.int32, device=DEV)
    ctx[::7] = 1000
    out1 = H.paged_decode(q, kc, vc, table, ctx, D ** -0.5)
    perm = torch.randperm(nb, generator=torch.Generator().manual_seed(1)).to(DEV)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(nb, device=DEV)
    kc2, vc2 = kc[perm], vc[perm]                  # new block j holds old block perm[j]
    table2 = inv[table.long()].to(torch.int32)     # old id -> new id
    out2 = H.paged_decode(q, kc2, vc2, table2, ctx, D ** -0.5)
    assert torch.equal(out1, out2)
    # and a sampled subset against the oracle
    rows = [0, 7, 100, 255]
    ref = O.paged_decode(q[rows].cpu(), kc.cpu(), vc.cpu(), table[rows].cpu(), ctx[rows].cpu(), D ** -0.5)
    torch.testing.assert_close(out1[rows].float().cpu(), ref, **_tol(dtype))


# ------------------------------------------------------------------ K6
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("cols", [128, 1024, 2048, 3584, 8192])
def test_rmsnorm_and_add(H, dtype, cols):
    rows = 19
    x = _randn(rows, cols, dtype=dtype, seed=1)
    r = _randn(rows, cols, dtype=dtype, seed=2)
    w = (1.0 + 0.1 * _randn(cols, dtype=torch.float32, seed=3)).to(dtype)
    eps = 1e-6
    y_ref = O.rmsnorm(x, w, eps)
    y = H.rmsnorm(x.to(DEV), w.to(DEV), eps)
    # single rounding of an fp32 result: at most 1 ulp of the output dtype
    ulp = {torch.bfloat16: 2 ** -7, torch.float16: 2 ** -10, torch.float32: 1e-5}[dtype]
    torch.testing.assert_close(y.float().cpu(), y_ref.float(), rtol=ulp, atol=ulp * 1e-2)
    y2_ref, res_ref = O.add_rmsnorm(x, r, w, eps)
    rd = r.to(DEV)
    y2, res = H.add_rmsnorm(x.to(DEV), rd, w.to(DEV), eps, inplace=True)
    assert res.data_ptr() == rd.data_ptr()
    assert torch.equal(res.cpu(), res_ref)  # the residual is an exactly rounded fp32 sum
    torch.testing.assert_close(y2.float().cpu(), y2_ref.float(), rtol=ulp, atol=ulp * 1e-2)
    if dtype != torch.float32:
        frac_exact = (y.cpu() == y_ref).float().mean().item()
        assert frac_exact > 0.99  # bit-exact but for reduction-order flips


def test_rmsnorm_strided_rows_and_3d(H):
    big = _randn(10, 4096, dtype=torch.bfloat16, seed=5).to(DEV)
    x = big[:, 1024:3072]  # row-strided view
    w = torch.ones(2048, dtype=torch.bfloat16, device=DEV)
    y = H.rmsnorm(x, w, 1e-6)
    torch.testing.assert_close(y.float().cpu(), O.rmsnorm(x.cpu(), w.cpu(), 1e-6).float(), rtol=2 ** -7, atol=1e-4)
    x3 = _randn(6, 8, 128, dtype=torch.bfloat16, seed=6).to(DEV)  # per-head norm shape
    w3 = torch.ones(128, dtype=torch.bfloat16, device=DEV)
    y3 = H.rmsnorm(x3, w3, 1e-6)
    assert y3.shape == x3.shape
    torch.testing.assert_close(y3.float().cpu(), O.rmsnorm(x3.cpu(), w3.cpu(), 1e-6).float(), rtol=2 ** -7, atol=1e-4)


# ------------------------------------------------------------------ K9
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_silu_and_mul(H, dtype):
    x = _randn(33, 2 * 6144, dtype=dtype, seed=8)
    ref = O.silu_and_mul(x)
    out = H.silu_and_mul(x.to(DEV))
    ulp = {torch.bfloat16: 2 ** -7, torch.float32: 1e-5}[dtype]
    torch.testing.assert_close(out.float().cpu(), ref.float(), rtol=2 * ulp, atol=1e-6)
    with pytest.raises(ValueError):
        H.silu_and_mul(torch.zeros(2, 7, device=DEV))


# ------------------------------------------------------------------ K7
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("qk_norm", [True, False])
@pytest.mark.parametrize("D", [128, 64])
def test_rope_qk_store(H, dtype, qk_norm, D):
    T, Hq, Hkv, nb, bs, max_pos = 45, 8, 4, 3, 64, 4096
    qkv = _randn(T, (Hq + 2 * Hkv) * D, dtype=dtype, seed=20)
    q = qkv[:, :Hq * D].view(T, Hq, D)
    k = qkv[:, Hq * D:(Hq + Hkv) * D].view(T, Hkv, D)
    v = qkv[:, (Hq + Hkv) * D:].view(T, Hkv, D)
    pos = torch.randint(0, max_pos, (T,), generator=torch.Generator().manual_seed(21))
    pos[0], pos[1] = 0, max_pos - 1
    cos, sin = O.rope_tables(D, max_pos, 1e6 if not qk_norm else 1e4)
    qw = (1 + 0.1 * _randn(D, dtype=torch.float32, seed=22)).to(dtype) if qk_norm else None
    kw = (1 + 0.1 * _randn(D, dtype=torch.float32, seed=23)).to(dtype) if qk_norm else None
    slots = torch.randperm(nb * bs, generator=torch.Generator().manual_seed(24))[:T].to(torch.int32)
    slots[5] = -1
    kc_ref = torch.zeros(nb, bs, Hkv, D, dtype=dtype)
    vc_ref = torch.zeros_like(kc_ref)
    q_ref, k_ref = O.rope_qk_store(q, k, v, pos, cos, sin, qw, kw, 1e-6, kc_ref, vc_ref, slots)
    d = lambda t: None if t is None else t.to(DEV)
    qkv_d = qkv.to(DEV)
    qd = qkv_d[:, :Hq * D].view(T, Hq, D)
    kd = qkv_d[:, Hq * D:(Hq + Hkv) * D].view(T, Hkv, D)
    vd = qkv_d[:, (Hq + Hkv) * D:].view(T, Hkv, D)
    kc, vc = torch.zeros_like(kc_ref).to(DEV), torch.zeros_like(vc_ref).to(DEV)
    q_out, k_out = H.rope_qk_store(qd, kd, vd, d(pos), d(cos), d(sin), d(qw), d(kw), 1e-6, kc, vc, d(slots))
    tol = dict(rtol=2 ** -7, atol=2 ** -7) if dtype == torch.bfloat16 else dict(rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(q_out.float().cpu(), q_ref.float(), **tol)
    torch.testing.assert_close(k_out.float().cpu(), k_ref.float(), **tol)
    assert torch.equal(vc.cpu(), vc_ref)                 # V rows are a byte copy
    assert torch.equal(kc.cpu().view(-1, Hkv, D)[slots[slots >= 0].long()], k_out.cpu()[slots >= 0])
    if not qk_norm:  # pure elementwise path restates the reference's rounding points: bit-exact
        assert torch.equal(q_out.cpu(), q_ref) and torch.equal(k_out.cpu(), k_ref)
    # no cache / no k_out variants
    q2, k2 = H.rope_qk_store(qd, kd, None, d(pos), d(cos), d(sin), d(qw), d(kw), 1e-6, want_k_out=False)
    assert k2 is None and torch.equal(q2, q_out)


# ------------------------------------------------------------------ K2
PREFILL_CASES = [
    # Hq, Hkv, D, seqlens
    (4, 2, 128, [5, 130, 64, 1]),
    (7, 1, 128, [200, 33]),
    (4, 4, 64, [17, 90]),
    (16, 8, 128, [257]),
    (8, 4, 128, [300, 1000, 129, 128, 127, 63, 65]),   # MFMA path: every tile / block edge
    (2, 1, 128, [2048]),                               # long sequence, many KV tiles
]


@pytest.mark.parametrize("case", PREFILL_CASES, ids=[f"case{i}" for i in range(len(PREFILL_CASES))])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_prefill_varlen_vs_oracle(H, case, dtype):
    Hq, Hkv, D, lens = case
    T = sum(lens)
    q = _randn(T, Hq, D, dtype=dtype, seed=30)
    k = _randn(T, Hkv, D, dtype=dtype, seed=31)
    v = _randn(T, Hkv, D, dtype=dtype, seed=32)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    scale = D ** -0.5
    ref = O.prefill_varlen(q, k, v, cu, cu, scale)
    out = H.prefill_varlen(q.to(DEV), k.to(DEV), v.to(DEV), cu.to(DEV), cu.to(DEV), max(lens), max(lens), scale)
    torch.testing.assert_close(out.float().cpu(), ref, **_tol(dtype))


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_prefill_mfma_every_workgroup_shape(H, variant, dtype):
    """The flash prefill kernel in both of its workgroup shapes (flags >> 4: 1 = 4 waves x 32 rows, 2 = 8 waves x 32
    rows; the default picks by prompt length), on lengths around every tile / row-block edge, contiguous and paged
    (prefix-cached) K/V."""
    Hq, Hkv, D = 8, 4, 128
    lens = [300, 1000, 129, 128, 127, 63, 65, 257, 1, 2049]
    T = sum(lens)
    q = _randn(T, Hq, D, dtype=dtype, seed=40)
    k = _randn(T, Hkv, D, dtype=dtype, seed=41)
    v = _randn(T, Hkv, D, dtype=dtype, seed=42)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    scale = D ** -0.5
    ref = O.prefill_varlen(q, k, v, cu, cu, scale)
    out = H.prefill_varlen(q.to(DEV), k.to(DEV), v.to(DEV), cu.to(DEV), cu.to(DEV), max(lens), max(lens), scale,
                           flags=variant << 4)
    torch.testing.assert_close(out.float().cpu(), ref, **_tol(dtype))
    # paged: the keys of each sequence live in the cache, the queries are the last third of it
    bs = 64
    kc, vc, table = _make_paged(3, Hkv, D, bs, [700, 131, 2300], dtype, seed=43)
    ks = [700, 131, 2300]
    qs = [233, 131, 767]
    qn = _randn(sum(qs), Hq, D, dtype=dtype, seed=44)
    cuq = torch.tensor([0] + list(torch.tensor(qs).cumsum(0)), dtype=torch.int32)
    cuk = torch.tensor([0] + list(torch.tensor(ks).cumsum(0)), dtype=torch.int32)
    kflat = torch.cat([torch.stack([kc[table[b, t // bs], t % bs] for t in range(n)]) for b, n in enumerate(ks)])
    vflat = torch.cat([torch.stack([vc[table[b, t // bs], t % bs] for t in range(n)]) for b, n in enumerate(ks)])
    ref2 = O.prefill_varlen(qn, kflat, vflat, cuq, cuk, scale)
    out2 = H.prefill_varlen(qn.to(DEV), None, None, cuq.to(DEV), cuk.to(DEV), max(qs), max(ks), scale,
                            k_cache=kc.to(DEV), v_cache=vc.to(DEV), block_tables=table.to(DEV), flags=variant << 4)
    torch.testing.assert_close(out2.float().cpu(), ref2, **_tol(dtype))


def test_prefill_mfma_strided_qkv_and_f16(H):
    """q/k/v as views of one fused qkv buffer (the engine's layout), f16, and a rescale-forcing
    input: one key per row with a huge score late in the sequence (online-softmax max jump)."""
    Hq, Hkv, D, lens = 4, 2, 128, [190, 70]
    T = sum(lens)
    for dtype in (torch.float16, torch.bfloat16):
        qkv = _randn(T, (Hq + 2 * Hkv) * D, dtype=dtype, seed=33)
        q = qkv[:, :Hq * D].view(T, Hq, D)
        k = qkv[:, Hq * D:(Hq + Hkv) * D].view(T, Hkv, D).clone()
        v = qkv[:, (Hq + Hkv) * D:].view(T, Hkv, D)
        k[150] = k[150] * 6.0     # spike: the running max jumps in the third KV tile
        k[10] = k[10] * -5.0
        cu = torch.tensor([0, lens[0], T], dtype=torch.int32)
        ref = O.prefill_varlen(q, k, v, cu, cu, D ** -0.5)
        qkv_d = qkv.to(DEV)
        out = H.prefill_varlen(qkv_d[:, :Hq * D].view(T, Hq, D), k.to(DEV),
                               qkv_d[:, (Hq + Hkv) * D:].view(T, Hkv, D), cu.to(DEV), cu.to(DEV),
                               max(lens), max(lens), D ** -0.5)
        torch.testing.assert_close(out.float().cpu(), ref, **_tol(dtype))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_prefill_paged_prefix(H, dtype):
    """Prefix-cache hit: queries are the uncached suffix, keys come from the paged cache
    (flash-attn semantics; oracle = full no-cache prefill rows)."""
    Hq, Hkv, D, bs = 8, 4, 128, 64
    full = [64 + 5, 128 + 64, 40]       # total lengths
    cached = [64, 128, 0]               # cached prefix (multiples of bs)
    B = len(full)
    kc, vc, table = _make_paged(B, Hkv, D, bs, full, dtype, seed=40)
    qs = [f - c for f, c in zip(full, cached)]
    q_full = _randn(sum(full), Hq, D, dtype=dtype, seed=41)
    cu_k = torch.tensor([0] + list(torch.tensor(full).cumsum(0)), dtype=torch.int32)
    cu_q = torch.tensor([0] + list(torch.tensor(qs).cumsum(0)), dtype=torch.int32)
    kk, vv = O.gather_paged_kv(kc, vc, table, full)
    ref_full = O.prefill_varlen(q_full, kk, vv, cu_k, cu_k, D ** -0.5)
    sel = torch.cat([torch.arange(int(cu_k[i]) + cached[i], int(cu_k[i + 1])) for i in range(B)])
    out = H.prefill_varlen(q_full[sel].to(DEV), None, None, cu_q.to(DEV), cu_k.to(DEV), max(qs), max(full),
                           D ** -0.5, k_cache=kc.to(DEV), v_cache=vc.to(DEV), block_tables=table.to(DEV))
    torch.testing.assert_close(out.float().cpu(), ref_full[sel], **_tol(dtype))


# ------------------------------------------------------------------ K8
def _sampler_inputs(B, V, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    # well separated values: a random permutation of a fine grid (no ties, exact in bf16 for small V)
    base = torch.stack([torch.randperm(V, generator=g).float() for _ in range(B)])
    logits = (base - V / 2) * (8.0 / V)
    return logits.to(dtype)


@pytest.mark.parametrize("V", [50, 1000, 50272])
def test_sampler_support_sets_fp32(H, V):
    B = 8
    logits = _sampler_inputs(B, V, torch.float32, seed=50)
    temps = torch.tensor([1.0, 0.6, 0.6, 1.5, 0.0, 1.0, 0.8, 1.0])
    top_ks = torch.tensor([-1, 40, -1, 5, 3, V + 5, 40, 1], dtype=torch.int64)
    top_ps = torch.tensor([1.0, 0.95, 0.9, 1.0, 0.5, 0.7, 0.95, 1.0])
    min_ps = torch.tensor([0.0, 0.0, 0.0, 0.0, 0.0, 0.05, 0.1, 0.0])
    keep_ref, _ = O.sampler_keep_mask(logits, temps, top_ks, top_ps, min_ps)
    d = lambda t: t.to(DEV)
    toks, mask = H.sample(d(logits), d(temps), d(top_ks), d(top_ps), d(min_ps), seed=1, offset=0, return_mask=True)
    mask = mask.bool().cpu()
    for b in range(B):
        diff = (mask[b] != keep_ref[b]).sum().item()
        # fp32 cumulative sums in a different order may move the nucleus edge by one token
        assert diff <= 1, f"row {b}: support differs in {diff} tokens"
        assert mask[b, toks[b]].item(), f"row {b}: sampled a filtered token"
    assert toks[4].item() == int(torch.argmax(logits[4]))  # temperature 0 -> argmax
    assert toks[7].item() == int(torch.argmax(logits[7]))  # top_k = 1


def test_sampler_bf16_and_ties(H):
    V = 151936
    g = torch.Generator().manual_seed(3)
    logits = (torch.randn(4, V, generator=g) * 2).to(torch.bfloat16)
    temps = torch.tensor([0.6, 1.0, 0.0, 0.7])
    top_ks = torch.tensor([40, 1000, -1, -1], dtype=torch.int64)
    top_ps = torch.tensor([0.95, 1.0, 1.0, 0.9])
    min_ps = torch.zeros(4)
    d = lambda t: t.to(DEV)
    toks, mask = H.sample(d(logits), d(temps), d(top_ks), d(top_ps), d(min_ps), seed=7, return_mask=True)
    mask = mask.bool().cpu()
    assert mask[0].sum().item() <= 40 and mask[1].sum().item() == 1000  # exactly k survive ties
    # every kept logit >= every dropped logit (a top set by value)
    for b in (0, 1, 3):
        assert logits[b][mask[b]].float().min() >= logits[b][~mask[b]].float().max()
    assert toks[2].item() == int(torch.argmax(logits[2].float()))
    # nucleus mass of row 3 reaches p, and dropping its smallest kept value group would fall below p
    probs = torch.softmax(logits[3].float() / 0.7, -1)
    assert probs[mask[3]].sum().item() >= 0.9 - 1e-3


def test_sampler_distribution_chi2(H):
    """Multinomial draw: 20000 draws of one row against the oracle's distribution."""
    V, N = 64, 20000
    logits = _sampler_inputs(1, V, torch.float32, seed=60).repeat(N, 1)
    temps = torch.full((N,), 0.9)
    top_ks = torch.full((N,), 20, dtype=torch.int64)
    top_ps = torch.full((N,), 0.9)
    min_ps = torch.zeros(N)
    d = lambda t: t.to(DEV)
    toks = H.sample(d(logits), d(temps), d(top_ks), d(top_ps), d(min_ps), seed=123, offset=5).cpu()
    probs = O.sampler_probs(logits[:1], temps[:1], top_ks[:1], top_ps[:1], min_ps[:1])[0]
    counts = torch.bincount(toks, minlength=V).float()
    assert counts[probs == 0].sum().item() == 0
    sup = probs > 0
    chi2 = (((counts[sup] - N * probs[sup]) ** 2) / (N * probs[sup])).sum().item()
    dof = int(sup.sum()) - 1
    assert chi2 < dof + 6 * math.sqrt(2 * dof), (chi2, dof)
    # different offset -> different stream; same (seed, offset) -> identical draws
    t2 = H.sample(d(logits), d(temps), d(top_ks), d(top_ps), d(min_ps), seed=123, offset=5).cpu()
    t3 = H.sample(d(logits), d(temps), d(top_ks), d(top_ps), d(min_ps), seed=123, offset=6).cpu()
    assert torch.equal(toks, t2) and not torch.equal(toks, t3)


def test_sampler_degenerate_rows(H):
    V = 1000
    logits = torch.full((2, V), float("-inf"))
    logits[1, 7] = 3.0
    d = lambda t: t.to(DEV)
    toks = H.sample(d(logits), d(torch.ones(2)), d(torch.tensor([-1, -1])), d(torch.ones(2)), d(torch.zeros(2)), seed=1)
    assert 0 <= toks[0].item() < V   # all filtered -> uniform fallback (functional.py:419-429)
    assert toks[1].item() == 7


# ------------------------------------------------------------------ f2, sampled rows: per-rank top-64 candidates
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("world,vocab", [(4, 151936), (8, 152064), (2, 320), (8, 320)])
def test_topc_candidates_hold_the_top_k_set_and_sample_like_the_full_row(H, dtype, world, vocab):
    """Every rank packs the 64 largest logits of its vocab shard (+ global indices); the slots are summed (here on
    the host: x + 0, what the all-reduce does) and unpacked.  The candidates must be exactly each shard's top 64
    by value with the right indices, and the sampler's kept set on the candidates, mapped back, must be the kept
    set of the full-row sampler for top_k <= 64 (ties at the k-th value may name another, equal-valued token)."""
    B, vpr = 6, vocab // world
    g = torch.Generator().manual_seed(vocab + world)
    full = (torch.randn(B, vocab, generator=g) * 3).to(dtype).to(DEV)
    bufs = [H.topc_pack(full[:, r * vpr:(r + 1) * vpr], r, world, B) for r in range(world)]
    assert all(b.shape == (world, B, 64, 4) for b in bufs)
    summed = torch.stack([b.float() for b in bufs]).sum(0).to(dtype)          # exact: one non-zero term per slot
    assert torch.equal(summed, torch.stack(bufs).float().sum(0).to(dtype))
    cl, ci = H.topc_unpack(summed, B)
    assert cl.shape == (B, world * 64) and ci.dtype == torch.int32
    real = min(64, vpr)
    for r in range(world):
        shard = full[:, r * vpr:(r + 1) * vpr].float()
        want_vals = shard.topk(real, dim=1).values
        got_vals, got_idx = cl[:, r * 64:r * 64 + real].float(), ci[:, r * 64:r * 64 + real].long()
        assert torch.equal(got_vals.sort(dim=1, descending=True).values, want_vals)
        assert ((got_idx >= r * vpr) & (got_idx < (r + 1) * vpr)).all()
        assert torch.equal(full.float().gather(1, got_idx), got_vals)            # index <-> value
        assert (got_idx.sort(dim=1).values.diff(dim=1) > 0).all()                   # no token twice
        if real < 64:
            assert torch.isinf(cl[:, r * 64 + real:(r + 1) * 64].float()).all()
    # the sampler on the candidates == the sampler on the full row (kept sets, then the same set of drawable tokens)
    temps = torch.tensor([0.7, 1.0, 0.0, 1.3, 0.9, 0.6], device=DEV)
    ks = torch.tensor([40, 1, -1, 64, 7, 50], dtype=torch.int64, device=DEV)     # row 2 greedy: k irrelevant
    ps = torch.tensor([0.95, 1.0, 1.0, 0.8, 1.0, 0.5], device=DEV)
    ms = torch.tensor([0.0, 0.0, 0.0, 0.0, 0.05, 0.0], device=DEV)
    tok_c, mask_c = H.sample(cl, temps, ks, ps, ms, seed=5, return_mask=True)
    tok_f, mask_f = H.sample(full, temps, ks, ps, ms, seed=5, return_mask=True)
    for b in range(B):
        kept_c = set(ci[b][mask_c[b].bool()].tolist())
        kept_f = set(mask_f[b].nonzero().flatten().tolist())
        vals_c = sorted(full[b, sorted(kept_c)].float().tolist())
        vals_f = sorted(full[b, sorted(kept_f)].float().tolist())
        assert len(kept_c) == len(kept_f) and vals_c == vals_f, b               # same values; ties may swap tokens
        assert int(ci[b, tok_c[b]]) in kept_c
    assert int(ci[2, tok_c[2]]) == int(full[2].float().argmax())                 # the greedy row: the global argmax (no tie here)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_topc_greedy_rows_break_ties_towards_the_lowest_token_id(H, dtype):
    """A tie at the maximum is plausible with 16-bit logits.  TP = 1 (argmax: lowest index) and the all-greedy exchange
    (mv_greedy_pack / mv_greedy_merge) name the LOWEST token id; the candidate path lays its candidates out in
    shard-visiting order, so its greedy rows must not take "the first maximum by candidate position" (mv_topc_pick)."""
    world, vocab, B = 4, 151936, 5
    vpr = vocab // world
    g = torch.Generator().manual_seed(11)
    full = (torch.randn(B, vocab, generator=g) * 2).to(dtype)
    top = float(full.float().max()) + 1.0
    # ties at the maximum: inside one shard (far apart, so that different threads visit them), across shards, adjacent
    tied = [[vpr + 9000, vpr + 17], [3 * vpr + 5, 40000, 2 * vpr + 1], [70001, 70000], [vocab - 1, 123], [777]]
    for b, idxs in enumerate(tied):
        full[b, idxs] = top
    full = full.to(DEV)
    want = [min(t) for t in tied]
    assert full.float().argmax(-1).tolist() == want                         # torch: first maximum = lowest index
    temps = torch.tensor([0.0, 0.0, 0.0, 0.0, 0.8], device=DEV)             # row 4 is sampled with top_k = 1
    ks = torch.tensor([-1, -1, -1, -1, 1], dtype=torch.int64, device=DEV)
    ones, zeros = torch.ones(B, device=DEV), torch.zeros(B, device=DEV)
    tp1 = H.sample(full, temps, ks, ones, zeros, seed=3)
    assert tp1.tolist() == want
    bufs = [H.topc_pack(full[:, r * vpr:(r + 1) * vpr], r, world, B) for r in range(world)]
    cl, ci = H.topc_unpack(torch.stack([b.float() for b in bufs]).sum(0).to(dtype), B)
    pos = H.sample(cl, temps, ks, ones, zeros, seed=3)
    toks = H.topc_pick(cl, ci, pos, temps)
    assert toks.dtype == torch.int64 and toks.tolist() == want
    # without the temperatures every row is "sampled": the drawn candidate's own index
    assert H.topc_pick(cl, ci, pos).tolist() == ci.gather(1, pos.unsqueeze(1)).squeeze(1).tolist()
    # the all-greedy exchange agrees
    local = [H.sample(full[:, r * vpr:(r + 1) * vpr].contiguous(), torch.zeros(B, device=DEV), ks, ones, zeros) for r in range(world)]
    packed = [H.greedy_pack(full[:, r * vpr:(r + 1) * vpr].contiguous(), local[r], r, world, B + (B & 1)) for r in range(world)]
    merged = H.greedy_merge(torch.stack([b.float() for b in packed]).sum(0).to(dtype), B)
    assert merged.tolist()[:4] == want[:4]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("world", [2, 8])
def test_mass_two_stage_draw_is_the_full_row_softmax(H, dtype, world):
    """f2 for the default SamplingParams (the whole vocabulary): every rank samples its own shard and reports {shard
    maximum, shard mass, local token}; the merge picks a rank by mass from an independent uniform.  Checked here with
    the ranks emulated on one device (the slots are summed on the host: x + 0, what the all-reduce does):
      * the pack / merge round trip is exact (maximum, mass to 2^-24 relative, token index),
      * greedy rows name the full row's argmax, lowest index on ties (also across shards),
      * chi-square of 24 000 draws against the softmax of the full row, for two temperatures,
      * a shard whose logits are all -inf is never picked."""
    vocab, B = 64 * world, 4
    vpr = vocab // world
    g = torch.Generator().manual_seed(100 + world)
    full = (torch.randn(B, vocab, generator=g) * 2).to(dtype)
    full[2, 5] = full[2, vpr + 3] = float(full[2].float().max()) + 1.0       # greedy row: a tie across two shards
    full[3, vpr:2 * vpr] = float("-inf")                                       # a dead shard in a sampled row
    full = full.to(DEV)
    temps = torch.tensor([1.0, 0.6, 0.0, 1.3], device=DEV)
    ks = torch.full((B,), -1, dtype=torch.int64, device=DEV)
    ones, zeros = torch.ones(B, device=DEV), torch.zeros(B, device=DEV)

    def draw(offset):
        bufs = []
        for r in range(world):
            shard = full[:, r * vpr:(r + 1) * vpr].contiguous()
            local, stats = H.sample(shard, temps, ks, ones, zeros, seed=9, offset=offset, return_stats=True)
            bufs.append((H.mass_pack(stats, local, r, world, vpr, B, dtype), stats, local))
        summed = torch.stack([b.float() for b, _, _ in bufs]).sum(0).to(dtype)
        return H.mass_merge(summed, B, temps, seed=9, offset=offset), bufs, summed

    toks, bufs, summed = draw(1)
    # round trip of the statistics through the exact 16-bit pieces
    for r, (buf, stats, local) in enumerate(bufs):
        mine = summed[r].float()
        assert torch.equal(mine[:, 0], stats[:, 0].to(dtype).float())
        mant = mine[:, 2] * 65536 + mine[:, 3] * 256 + mine[:, 4]
        mass = torch.ldexp(mant / 2 ** 24, (mine[:, 1] - 64).int())
        torch.testing.assert_close(mass, stats[:, 1], rtol=2 ** -23, atol=0)
        gidx = (mine[:, 5] * 65536 + mine[:, 6] * 256 + mine[:, 7]).long()
        assert torch.equal(gidx, local + r * vpr)
        want_mass = ((shard_f := full[:, r * vpr:(r + 1) * vpr].float()) - stats[:, :1]).div(temps.clamp_min(1e-8)[:, None]).exp().sum(1)
        ok = temps > 0
        ok[3] = ok[3] and r != 1
        torch.testing.assert_close(stats[ok, 1], want_mass[ok], rtol=2e-3, atol=0)
    assert toks[2].item() == 5                                                  # greedy, tie: the lowest token id
    n_draws = 24000
    counts = torch.zeros(B, vocab, dtype=torch.float64)
    for off in range(n_draws // 8):                                             # 8 sampled rows' worth per call? no: per row
        t = draw(off + 2)[0].cpu()
        counts[torch.arange(B), t] += 1
    n = counts.sum(1)
    for b in (0, 1, 3):
        p = torch.softmax(full[b].double().cpu() / float(temps[b]), dim=-1)
        assert counts[b][p == 0].sum() == 0                                     # (the dead shard of row 3)
        keep = p * n[b] >= 5                                                    # chi-square on cells with enough mass
        exp = torch.cat([p[keep] * n[b], (p[~keep].sum() * n[b]).view(1)])
        obs = torch.cat([counts[b][keep], counts[b][~keep].sum().view(1)])
        chi2 = (((obs - exp) ** 2) / exp.clamp_min(1e-9)).sum().item()
        dof = int(keep.sum())
        assert chi2 < dof + 5 * (2 * dof) ** 0.5, (b, chi2, dof)
    assert (counts[2] > 0).sum() == 1                                           # the greedy row never moves


def _emulated_nucleus_floor(H, full, world, temps, top_ps, min_ps):
    """The cross-rank walk of tp_nucleus_tokens with the ranks emulated on one device (the all-reduces are host sums)."""
    B, vocab = full.shape
    vpr = vocab // world
    shards = [full[:, r * vpr:(r + 1) * vpr].contiguous() for r in range(world)]
    row_max = torch.stack([s.amax(dim=1).float() for s in shards]).amax(dim=0)
    h0 = sum(H.shard_key_hist(s, row_max, temps, None) for s in shards)
    state, prefix = H.nucleus_pick(h0, 0, top_ps, min_ps, temps, row_max, full.dtype)
    h1 = sum(H.shard_key_hist(s, row_max, temps, prefix) for s in shards)
    floor = H.nucleus_pick(h1, 1, top_ps, min_ps, temps, row_max, full.dtype, state, prefix)
    return shards, row_max, h0, floor


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("world", [2, 8])
def test_nucleus_floor_across_shards_is_the_full_row_keep_mask(H, dtype, world):
    """f2 for nucleus / min-p rows: the floor the emulated ranks agree on keeps exactly the set mv_sample keeps on the
    full row (its keep mask is the oracle-checked one), for peaked and flat rows, top-p alone, min-p alone and both;
    the level-0 histogram sums to the row's mass; greedy and unfiltered rows get -inf."""
    vocab, B = 1024 * world, 8
    g = torch.Generator().manual_seed(300 + world)
    full = torch.randn(B, vocab, generator=g)
    full[:4] *= 3.0                                   # peaked rows
    full[4:] *= 0.3                                   # flat rows: the edge lies deep in the row
    full = full.to(dtype).to(DEV)
    temps = torch.tensor([1.0, 0.7, 1.0, 0.0, 1.0, 1.5, 1.0, 1.0], device=DEV)
    top_ps = torch.tensor([0.9, 0.5, 1.0, 0.9, 0.95, 0.3, 1.0, 0.999], device=DEV)
    min_ps = torch.tensor([0.0, 0.0, 0.05, 0.0, 0.0, 0.2, 0.0, 0.5], device=DEV)
    ks = torch.full((B,), -1, dtype=torch.int64, device=DEV)
    shards, row_max, h0, floor = _emulated_nucleus_floor(H, full, world, temps, top_ps, min_ps)
    assert torch.equal(row_max, full.float().amax(dim=1))
    z = ((full.double() - row_max.double()[:, None]) / temps.double().clamp_min(1e-8)[:, None]).exp().sum(1)
    live = temps > 0
    torch.testing.assert_close(h0.sum(1).double()[live] / 2.0 ** 40, z[live], rtol=1e-5, atol=0)
    assert (h0[~live] == 0).all()
    assert floor[3].item() == float("-inf") and floor[6].item() == float("-inf")
    _, mask = H.sample(full, temps, ks, top_ps, min_ps, seed=1, offset=1, return_mask=True)
    for b in range(B):
        if temps[b] == 0:
            continue
        mine = full[b].float() >= floor[b]
        assert torch.equal(mine, mask[b].bool()), (b, int(mine.sum()), int(mask[b].sum()))
    masked = torch.cat([H.mask_below(s, floor) for s in shards], dim=1)
    assert torch.equal(masked, torch.where(full.float() >= floor[:, None], full, torch.full_like(full, float("-inf"))))


@pytest.mark.parametrize("world", [2, 8])
def test_nucleus_floor_with_ties_at_the_edge(H, world):
    """rows made of a few distinct values, so that the nucleus edge falls ON a tie group, with p just below
    and just above the group boundaries -- the case where the cross-rank floor (budget = trunc(p * Z) on 2^-40 fixed-point
    masses, `mv_nucleus_pick`) could part ways with `mv_sample` (fp32 masses) and with the reference's sort-based cut.
    Pinned here: (1) the emulated ranks keep EXACTLY `mv_sample`'s set, at every p; (2) both keep the edge's tie group
    WHOLE, where the reference (functional.py:137-191: sort, cumsum, shift by one) cuts it at whatever position its sort
    leaves the equal values in -- so the reference's kept set is a subset that differs only inside that one group (the
    documented, intended difference: DESIGN.md section 3); (3) an all-equal row keeps everything."""
    dtype, vocab = torch.bfloat16, 256 * world
    levels = [(4.0, 3), (2.5, 17), (1.0, 100)]                       # (value, count); the rest of the row is 0.0
    row = torch.zeros(vocab)
    perm = torch.randperm(vocab, generator=torch.Generator().manual_seed(7))   # tie groups scattered over all shards
    o = 0
    for val, cnt in levels:
        row[perm[o:o + cnt]] = val
        o += cnt
    mass = [(v, c * torch.tensor(v).exp().item()) for v, c in levels] + [(0.0, float(vocab - o))]
    z = sum(m for _, m in mass)
    cum, ps = 0.0, []
    for _, m in mass[:-1]:
        cum += m
        ps += [cum / z - 2e-3, cum / z + 2e-3]                       # just inside / just past every group boundary
    ps += [0.02, 0.9999]
    B = len(ps) + 1
    full = row.repeat(B, 1)
    full[-1] = 1.5                                                   # the all-equal row
    full = full.to(dtype).to(DEV)
    temps = torch.ones(B, device=DEV)
    top_ps = torch.tensor(ps + [0.5], device=DEV)
    min_ps = torch.zeros(B, device=DEV)
    ks = torch.full((B,), -1, dtype=torch.int64, device=DEV)
    shards, _, _, floor = _emulated_nucleus_floor(H, full, world, temps, top_ps, min_ps)
    _, mask = H.sample(full, temps, ks, top_ps, min_ps, seed=1, offset=1, return_mask=True)
    ref_keep = O.sampler_probs(full.float().cpu(), temps.cpu(), ks.cpu(), top_ps.cpu(), min_ps.cpu()) > 0
    vals = full.float().cpu()
    for b in range(B):
        mine = (full[b].float() >= floor[b]).cpu()
        assert torch.equal(mine, mask[b].bool().cpu()), (b, ps[b] if b < len(ps) else "flat", int(mine.sum()), int(mask[b].sum()))
        edge = vals[b][mine].min()
        assert torch.equal(mine, vals[b] >= edge)                    # whole tie groups, nothing else
        ref = ref_keep[b]
        assert bool((ref & ~mine).sum() == 0), b                     # the reference keeps nothing this build drops
        assert vals[b][ref].min() == edge                            # ... and stops inside the same tie group
        assert torch.equal(ref | (vals[b] == edge), mine)
    assert mask[-1].all() and floor[-1].item() <= 1.5
    # the expected group count at each constructed p: below a boundary the group that crosses it is the edge
    kept_groups = [int((vals[b][(full[b].float() >= floor[b]).cpu()]).unique().numel()) for b in range(len(ps))]
    assert kept_groups[:6] == [1, 2, 2, 3, 3, 4], kept_groups


@pytest.mark.parametrize("world", [2, 4])
def test_nucleus_draw_across_shards_follows_the_filtered_softmax(H, world):
    """The whole gather-free nucleus sequence (floor -> masked shards -> local draws with statistics -> rank pick) with
    the ranks emulated on one device: every token lies in the kept set and 6 000 draws per row pass a chi-square
    against the renormalised softmax over it."""
    dtype = torch.bfloat16
    vocab, B = 96 * world, 3
    vpr = vocab // world
    g = torch.Generator().manual_seed(400 + world)
    full = (torch.randn(B, vocab, generator=g) * 2).to(dtype).to(DEV)
    temps = torch.tensor([1.0, 0.8, 1.2], device=DEV)
    top_ps = torch.tensor([0.8, 1.0, 0.6], device=DEV)
    min_ps = torch.tensor([0.0, 0.1, 0.02], device=DEV)
    ks = torch.full((B,), -1, dtype=torch.int64, device=DEV)
    shards, _, _, floor = _emulated_nucleus_floor(H, full, world, temps, top_ps, min_ps)
    masked = [H.mask_below(s, floor) for s in shards]
    n_draws = 6000
    counts = torch.zeros(B, vocab, dtype=torch.float64)
    for off in range(n_draws):
        bufs = []
        for r in range(world):
            local, stats = H.sample(masked[r], temps, None, None, None, seed=5, offset=off, return_stats=True)
            bufs.append(H.mass_pack(stats, local, r, world, vpr, B, dtype))
        summed = torch.stack([b.float() for b in bufs]).sum(0).to(dtype)
        counts[torch.arange(B), H.mass_merge(summed, B, temps, seed=5, offset=off).cpu()] += 1
    _, mask = H.sample(full, temps, ks, top_ps, min_ps, seed=1, offset=1, return_mask=True)
    for b in range(B):
        keep = mask[b].bool().cpu()
        assert counts[b][~keep].sum() == 0
        p = torch.softmax(full[b].double().cpu() / float(temps[b]), dim=-1) * keep
        p = p / p.sum()
        cells = p * n_draws >= 5
        exp = torch.cat([p[cells] * n_draws, (p[~cells].sum() * n_draws).view(1)])
        obs = torch.cat([counts[b][cells], counts[b][~cells].sum().view(1)])
        chi2 = (((obs - exp) ** 2) / exp.clamp_min(1e-9)).sum().item()
        dof = int(cells.sum())
        assert chi2 < dof + 5 * (2 * dof) ** 0.5, (b, chi2, dof)


# ------------------------------------------------------------------ K10
def _ulp_ok(got, exact, dtype, ulps):
    """|got - exact| <= ulps units in the last place of ``dtype`` at the magnitude of ``exact``
    (+ one ulp of the accumulation scale for results that cancel to ~0)."""
    eps = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    err = (got.double() - exact).abs()
    bound = ulps * eps * exact.abs() + eps * exact.abs().mean()
    return bool((err <= bound).all()), float((err / bound).max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k,bias", [
    (1, 2048, 2048, False), (1, 4608, 3584, True), (2, 12288, 2048, False), (3, 40, 256, True),
    (7, 2048, 6144, False), (16, 4096, 2048, True), (17, 2064, 288, False), (33, 1000, 2048 + 32, True),
    (64, 16384 + 16, 512, False), (5, 151936, 1024, False),
])
def test_linear_small_m(H, dtype, m, n, k, bias):
    # fp32 accumulation of 16-bit products, one rounding: within 1 ulp of the exact (float64) result
    x = _randn(m, k, dtype=dtype, seed=m + n).to(DEV)
    w = (_randn(n, k, dtype=torch.float32, seed=n + k) * k ** -0.5).to(dtype).to(DEV)
    b = _randn(n, dtype=dtype, seed=3).to(DEV) if bias else None
    exact = O.linear(x.cpu(), w.cpu(), None if b is None else b.cpu())
    got = H.linear_small_m(x, w, b)
    assert got.shape == (m, n) and got.dtype == dtype
    ok, worst = _ulp_ok(got.cpu(), exact, dtype, 1.0)
    assert ok, worst
    # strided input rows (a column block of a wider activation), as q/k/v views are
    wide = torch.zeros(m, k + 64, dtype=dtype, device=DEV)
    wide[:, 32:32 + k] = x
    got2 = H.linear_small_m(wide[:, 32:32 + k], w, b)
    assert torch.equal(got2, got)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k,bias", [
    (1, 2048, 2048, False), (1, 4608, 3584, True), (2, 2048, 6144, False), (3, 40, 256, True), (4, 3584, 18944, False),
    (16, 4096, 2048, True), (17, 2064, 288, False), (33, 1000, 2048 + 32, True), (5, 151936, 1024, False),
])
def test_linear_small_m_on_the_packed_weight(H, dtype, m, n, k, bias):
    """mv_linear_small_m_packed: the same projection streamed from the fragment-ordered copy (mv_pack_weight) —
    within 1 ulp of the exact result like the row-major kernel, ragged N (zero-padded tiles) included."""
    x = _randn(m, k, dtype=dtype, seed=m + n).to(DEV)
    w = (_randn(n, k, dtype=torch.float32, seed=n + k) * k ** -0.5).to(dtype).to(DEV)
    b = _randn(n, dtype=dtype, seed=3).to(DEV) if bias else None
    wp = H.pack_weight(w)
    exact = O.linear(x.cpu(), w.cpu(), None if b is None else b.cpu())
    got = H.linear_small_m(x, w, b, packed=wp)
    assert got.shape == (m, n) and got.dtype == dtype
    ok, worst = _ulp_ok(got.cpu(), exact, dtype, 1.0)
    assert ok, worst
    # against the row-major kernel: equal but for results whose fp32 sums were cut into different K slices
    plain = H.linear_small_m(x, w, b)
    assert (got != plain).float().mean().item() < 0.02
    # a weight that is not the packed copy of `weight` is refused by size, not silently used
    assert torch.equal(H.linear_small_m(x, w, b, packed=wp.view(-1)[:-8]), plain)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,inter,k", [(1, 6144, 2048), (4, 18944, 3584), (16, 1008, 512)])
def test_gate_up_silu_small_m_on_the_packed_weight(H, dtype, m, inter, k):
    x = _randn(m, k, dtype=dtype, seed=m).to(DEV)
    w = (_randn(2 * inter, k, dtype=torch.float32, seed=inter) * k ** -0.5).to(dtype).to(DEV)
    got = H.gate_up_silu_small_m(x, w, packed=H.pack_weight(w))
    want = O.gate_up_silu(x.cpu(), w.cpu(), dtype).double()
    # (a GEMM output that lands on the other side of a rounding boundary moves the product by up to ~3 ulp; 4 with
    #  the K slices cut differently from the row-major kernel's)
    ok, worst = _ulp_ok(got.cpu(), want, dtype, 4.0)
    assert ok, worst
    ok, worst = _ulp_ok(got.cpu(), H.gate_up_silu_small_m(x, w).cpu().double(), dtype, 4.0)
    assert ok, worst


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k,bias,gated", [
    (1, 4096, 2048, False, False), (4, 4096, 2048, True, False), (2, 4608, 3584, True, False),
    (1, 1280, 8192, True, False), (3, 3072, 1024, False, False), (1, 6144, 2048, False, True),
    (2, 18944, 3584, False, True), (4, 3072, 1024, False, True), (1, 152064, 3584, False, False),
    # round 4: up to 8 rows, in passes of 4 / PT rows (K = 2048 / 1024: 4 per pass; K = 3584: 2 per pass -> 4 rows at most)
    (5, 4096, 2048, False, False), (8, 4096, 2048, True, False), (7, 6144, 2048, False, True), (8, 6144, 2048, False, True),
    (6, 3072, 1024, False, False), (4, 4608, 3584, True, False), (3, 18944, 3584, False, True),
])
@pytest.mark.parametrize("with_residual", [True, False])
def test_norm_folded_into_the_small_m_projection_is_bit_identical(H, dtype, m, n, k, bias, gated, with_residual, monkeypatch):
    """mv_norm_linear_small_m_packed / mv_norm_gate_up_silu_small_m_packed: RMSNorm (+ residual add) in the
    prologue of the projection == mv_(add_)rmsnorm followed by the projection on the packed weight, bit for bit
    (outputs AND the new residual), and the residual input is left untouched.  (The engine folds one pass of rows by
    default; the two-pass cases run with the limit raised, as MV_NORM_FOLD_MAX_ROWS=8 does.)"""
    monkeypatch.setattr(H, "NORM_FUSED_MAX_ROWS", 8)
    x = _randn(m, k, dtype=dtype, seed=m + n).to(DEV)
    res = _randn(m, k, dtype=dtype, seed=7).to(DEV) if with_residual else None
    nw = (1.0 + 0.1 * _randn(k, dtype=torch.float32, seed=3)).to(dtype).to(DEV)
    rows_w = 2 * n if gated else n
    w = (_randn(rows_w, k, dtype=torch.float32, seed=n + k) * k ** -0.5).to(dtype).to(DEV)
    b = _randn(n, dtype=dtype, seed=5).to(DEV) if bias else None
    wp = H.pack_weight(w)
    assert H.norm_linear_small_m_ok(x, rows_w, wp)
    eps = 1e-6
    if with_residual:
        h_ref, res_ref = H.add_rmsnorm(x, res, nw, eps, inplace=False)
    else:
        h_ref, res_ref = H.rmsnorm(x, nw, eps), x
    want = H.gate_up_silu_small_m(h_ref, w, packed=wp) if gated else H.linear_small_m(h_ref, w, b, packed=wp)
    res_before = None if res is None else res.clone()
    got = H.norm_linear_small_m(x, res, nw, eps, wp, n, b, gated)
    assert got is not None
    out, res_out = got
    assert torch.equal(out, want)
    assert torch.equal(res_out, res_ref)
    if res is not None:
        assert torch.equal(res, res_before) and res_out.data_ptr() != res.data_ptr()
    else:
        assert res_out.data_ptr() == x.data_ptr()
    # and against the oracle's norm + linear (the two-launch path is itself checked against it elsewhere)
    if not gated:
        exact = O.linear(h_ref.cpu(), w.cpu(), None if b is None else b.cpu())
        ok, worst = _ulp_ok(out.cpu(), exact, dtype, 1.0)
        assert ok, worst


def test_norm_folding_refuses_an_aliased_residual(H):
    """The new residual is written by ONE workgroup while the others still read the old one: the C entry point refuses
    residual_out == residual instead of racing."""
    import ctypes

    dtype = torch.bfloat16
    x = torch.ones(2, 256, dtype=dtype, device=DEV)
    res = torch.ones(2, 256, dtype=dtype, device=DEV)
    nw = torch.ones(256, dtype=dtype, device=DEV)
    w = torch.zeros(64, 256, dtype=dtype, device=DEV)
    wp = H.pack_weight(w)
    out = torch.empty(2, 64, dtype=dtype, device=DEV)
    lib = H._lib
    rc = lib.mv_norm_linear_small_m_packed(x.data_ptr(), 256, res.data_ptr(), 256, res.data_ptr(), 256, nw.data_ptr(),
                                           ctypes.c_float(1e-6), wp.data_ptr(), None, out.data_ptr(), 64, 2, 64, 256,
                                           0, 0, None)
    assert rc == -7 and b"alias" in lib.mv_strerror(rc).lower()
    rc = lib.mv_norm_linear_small_m_packed(x.data_ptr(), 256, res.data_ptr(), 256, None, 0, nw.data_ptr(),
                                           ctypes.c_float(1e-6), wp.data_ptr(), None, out.data_ptr(), 64, 2, 64, 256,
                                           0, 0, None)
    assert rc < 0                                    # a residual without a place for the new one


def test_norm_folding_declines_what_it_cannot_hold(H):
    dtype = torch.bfloat16
    w = torch.zeros(256, 3584, dtype=dtype, device=DEV)
    wp = H.pack_weight(w)
    assert H.NORM_FUSED_MAX_ROWS == 4 or os.environ.get("MV_NORM_FOLD_MAX_ROWS")
    assert not H.norm_linear_small_m_ok(torch.zeros(5, 3584, dtype=dtype, device=DEV), 256, wp)   # > 4 rows
    assert not H.norm_linear_small_m_ok(torch.zeros(3, 3584, dtype=dtype, device=DEV), 256, wp)   # 3 rows x 2 vectors
    assert H.norm_linear_small_m_ok(torch.zeros(2, 3584, dtype=dtype, device=DEV), 256, wp)
    # the kernel itself takes two passes (round 4) and declines a third: 5 rows x 2 vectors per norm thread
    x = torch.zeros(5, 3584, dtype=dtype, device=DEV)
    assert H.norm_linear_small_m(x, None, torch.ones(3584, dtype=dtype, device=DEV), 1e-6, wp, 256) is None


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,inter,k", [(1, 6144, 2048), (4, 18944, 3584), (16, 1000, 512), (40, 16400, 256)])
def test_gate_up_silu_small_m(H, dtype, m, inter, k):
    x = _randn(m, k, dtype=dtype, seed=m).to(DEV)
    w = (_randn(2 * inter, k, dtype=torch.float32, seed=inter) * k ** -0.5).to(dtype).to(DEV)
    got = H.gate_up_silu_small_m(x, w)
    assert got.shape == (m, inter)
    # the unfused pair on the GPU (same rounding points; summation order may differ by slicing)
    unfused = H.silu_and_mul(H.linear_small_m(x, w))
    exact_gu = O.linear(x.cpu(), w.cpu())
    want = O.gate_up_silu(x.cpu(), w.cpu(), dtype).double()
    # a GEMM output within 1 ulp of a rounding boundary may round either way: allow 3 ulp end to end
    ok, worst = _ulp_ok(got.cpu(), want, dtype, 3.0)
    assert ok, worst
    ok, worst = _ulp_ok(got.cpu(), unfused.cpu().double(), dtype, 3.0)
    assert ok, worst
    assert exact_gu.shape == (m, 2 * inter)


# ------------------------------------------------------------------ K11
def test_pack_weight_is_the_documented_permutation(H):
    # [N,K] -> [ceil(N/16)][K/32][64 lanes][8]: lane l holds W[16*i + (l & 15)][32*j + 8*(l >> 4) ..+8]; rows >= N zero
    for n, k in ((48, 64), (40, 96), (1280, 256)):
        w = _randn(n, k, dtype=torch.bfloat16, seed=n).to(DEV)
        p = H.pack_weight(w)
        npad = (n + 15) // 16 * 16
        assert p.shape == (npad // 16, k // 32, 64, 8)
        back = p.view(npad // 16, k // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(npad, k)
        assert torch.equal(back[:n], w) and back[n:].abs().sum() == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k,bias", [
    (256, 4096, 2048, False), (256, 4608, 3584, True), (256, 2048, 6144, False), (200, 1000, 512, True),
    (64, 12288, 1024, False), (17, 2064, 128, False), (129, 8192 + 16, 64, True), (512, 1280, 8192, True),
])
def test_linear_mid_m_every_tile(H, dtype, m, n, k, bias):
    # fp32 accumulation of 16-bit products, one rounding: within 1 ulp of the exact (float64) result, for every
    # workgroup tile of the table (ragged row / column tiles, K loops shorter than the rings, rotated K order)
    x = _randn(m, k, dtype=dtype, seed=m + n).to(DEV)
    w = (_randn(n, k, dtype=torch.float32, seed=n + k) * k ** -0.5).to(dtype).to(DEV)
    b = _randn(n, dtype=dtype, seed=3).to(DEV) if bias else None
    exact = O.linear(x.cpu(), w.cpu(), None if b is None else b.cpu())
    wp = H.pack_weight(w)
    outs = {}
    for cfg in range(0, 28):
        got = outs[cfg] = H.linear_mid_m(x, wp, n, b, cfg=cfg)
        assert got.shape == (m, n) and got.dtype == dtype
        ok, worst = _ulp_ok(got.cpu(), exact, dtype, 1.0)
        assert ok, (cfg, worst)
    # tiles 20..27 are tiles 14 / 11 / 15 / 2 / 16 / 4 / 17 / 9 with another instruction ORDER (LDS reads between the
    # MFMAs): the same sums in the same order, bit for bit
    for new, old in ((20, 14), (21, 11), (22, 15), (23, 2), (24, 16), (25, 4), (26, 17), (27, 9)):
        assert torch.equal(outs[new], outs[old]), (new, old)
    # strided input rows (a column block of a wider activation)
    wide = torch.zeros(m, k + 64, dtype=dtype, device=DEV)
    wide[:, 32:32 + k] = x
    assert torch.equal(H.linear_mid_m(wide[:, 32:32 + k], wp, n, b, cfg=1), H.linear_mid_m(x, wp, n, b, cfg=1))


def test_timing_ablation_switches_are_not_reachable_in_the_production_build(H):
    """the bench-only switches of mv_linear_mid_m (cfg + 100 x flags: DMA ring only, no x / W DMA, clock
    stamps INTO `bias`, no output stores) and of mv_prefill_varlen (no K/V refresh, no output stores) leave garbage or
    write out of bounds.  They are compiled only with `make ABLATE=1`; the shipped library answers MV_E_RANGE."""
    x = _randn(64, 256, dtype=torch.bfloat16, seed=1).to(DEV)
    w = _randn(128, 256, dtype=torch.bfloat16, seed=2).to(DEV)
    b = _randn(128, dtype=torch.bfloat16, seed=3).to(DEV)
    wp = H.pack_weight(w)
    good = H.linear_mid_m(x, wp, 128, b, cfg=4)
    for flag in (100, 200, 400, 800, 1600, 3200):
        with pytest.raises(H.MvHipError, match="range"):
            H.linear_mid_m(x, wp, 128, b, cfg=4 + flag)
    assert torch.equal(H.linear_mid_m(x, wp, 128, b, cfg=4), good)
    q = _randn(70, 4, 128, dtype=torch.bfloat16, seed=4).to(DEV)
    k = _randn(70, 2, 128, dtype=torch.bfloat16, seed=5).to(DEV)
    v = _randn(70, 2, 128, dtype=torch.bfloat16, seed=6).to(DEV)
    cu = torch.tensor([0, 70], dtype=torch.int32, device=DEV)
    ok = H.prefill_varlen(q, k, v, cu, cu, 70, 70, 128 ** -0.5)
    for variant in (16, 1 << 8, 2 << 8, 1 << 10, 5 | 16):
        with pytest.raises(H.MvHipError, match="range"):
            H.prefill_varlen(q, k, v, cu, cu, 70, 70, 128 ** -0.5, flags=variant << 4)
    assert torch.equal(H.prefill_varlen(q, k, v, cu, cu, 70, 70, 128 ** -0.5), ok)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,inter,k", [(256, 6144, 2048), (128, 18944, 512), (40, 1008, 256), (256, 3696, 1024)])
def test_gate_up_silu_mid_m(H, dtype, m, inter, k):
    x = _randn(m, k, dtype=dtype, seed=m).to(DEV)
    w = (_randn(2 * inter, k, dtype=torch.float32, seed=inter) * k ** -0.5).to(dtype).to(DEV)
    wp = H.pack_weight(w)
    want = O.gate_up_silu(x.cpu(), w.cpu(), dtype).double()
    for cfg in (0, 1, 3, 5, 6, 7, 8, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 24, 26):
        got = H.gate_up_silu_mid_m(x, wp, inter, cfg=cfg)
        assert got.shape == (m, inter)
        # the fused epilogue IS the unfused pair: same GEMM (checked to 1 ulp of the exact product above), same
        # rounding points -> bit-identical to mv_linear_mid_m + mv_silu_and_mul
        assert torch.equal(got, H.silu_and_mul(H.linear_mid_m(x, wp, 2 * inter, cfg=cfg))), cfg
        # against the oracle (GEMM rounded from the exact product): a gate within 1 ulp of a rounding boundary
        # may round the other way, and silu's elasticity 1 + x(1 - sigmoid(x)) amplifies that where the gate is
        # negative; outputs there are tiny, so the bound is a few ulp of the output scale
        ok, worst = _ulp_ok(got.cpu(), want, dtype, 8.0)
        assert ok, (cfg, worst)
    with pytest.raises(H.MvHipError):
        H.gate_up_silu_mid_m(x, wp, inter, cfg=2)   # a tile whose waves do not hold gate/up pairs


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k,splits", [(256, 2048, 6144, 2), (256, 3584, 2048, 4), (100, 1000, 1024, 4), (64, 512, 192, 2),
                                            (256, 3584, 9472, 8)])
def test_split_k_partials_and_the_norm_behind_them(H, dtype, m, n, k, splits):
    x = _randn(m, k, dtype=dtype, seed=m).to(DEV)
    w = (_randn(n, k, dtype=torch.float32, seed=n) * k ** -0.5).to(dtype).to(DEV)
    res = _randn(m, n, dtype=dtype, seed=5).to(DEV)
    nw = (1 + 0.1 * _randn(n, dtype=torch.float32, seed=6)).to(dtype).to(DEV)
    wp = H.pack_weight(w)
    exact = O.linear(x.cpu(), w.cpu())
    for cfg in (0, 1, 6, 10, 11, 15, 17, 21, 22, 26):
        part = H.linear_mid_m_partial(x, wp, n, splits, cfg=cfg)
        assert part.shape == (splits, m, n) and part.dtype == torch.float32
        assert (part.double().sum(0).cpu() - exact).abs().max() < 1e-4 * max(1.0, exact.abs().max())
        # the consumer: sum in split order, ONE rounding to dtype (the GEMM's), then add + norm exactly as
        # mv_add_rmsnorm does on that rounded tensor
        acc = part[0]
        for i in range(1, splits):
            acc = acc + part[i]                 # split order, fp32
        y_ref, r_ref = H.add_rmsnorm(acc.to(dtype), res.clone(), nw, 1e-6)
        y, r = H.add_rmsnorm_partial(part, res.clone(), nw, 1e-6)
        assert torch.equal(r, r_ref)          # the summed, rounded projection + residual: bit-exact
        # the normed output: same formula on the same values in another instantiation of the kernel template;
        # measured: 24 of 917 504 elements (f16, 3584 columns) land on the other side of a rounding tie whose
        # exact value is within 6e-8 relative of the midpoint, i.e. inside fp32 noise -> 1 ulp on < 0.1 %
        diff = y != y_ref
        assert diff.float().mean().item() < 1e-3
        ok, worst = _ulp_ok(y.cpu(), y_ref.cpu().double(), dtype, 2.0)   # one unit in the last place = up to 2 x eps x |y|
        assert ok, worst


def test_linear_small_m_argument_errors(H):
    x = torch.zeros(4, 256, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(64, 256, dtype=torch.bfloat16, device=DEV)
    assert H.linear_small_m(x, w).abs().sum() == 0
    assert H.linear_small_m(x[:0], w).shape == (0, 64)
    for bad_x, bad_w in [(torch.zeros(65, 256, dtype=torch.bfloat16, device=DEV), w),            # M > 64
                         (torch.zeros(4, 224, dtype=torch.bfloat16, device=DEV), w[:, :224].contiguous()),  # K < 256
                         (x.float(), w.float())]:                                                  # fp32
        with pytest.raises(H.MvHipError):
            H.linear_small_m(bad_x, bad_w)
        assert not H.linear_small_m_ok(bad_x, bad_w)
    assert H.linear_small_m_ok(x, w)


# ------------------------------------------------------------------ f2
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_greedy_pack_merge_equals_argmax_of_gathered_row(H, dtype):
    """Two vocab shards packed by 'rank 0' and 'rank 1', summed (what the all-reduce does: x + 0), merged:
    the argmax of the concatenated row, lowest index on ties, indices above 2^16 included."""
    world, rows, vpr = 2, 5, 75968
    full = _randn(rows, world * vpr, dtype=dtype, seed=9).to(DEV)
    full[1, 70000], full[1, 140000] = 9.0, 9.0          # tie across shards -> 70000
    full[2, 151000], full[2, 151900] = 8.0, 8.0         # tie inside shard 1 -> 151000
    full[3] = 0.5                                        # everything ties -> 0
    full[4, 151935] = 11.0                               # last entry
    zeros, ones = torch.zeros(rows, device=DEV), torch.ones(rows, device=DEV)
    nok = torch.full((rows,), -1, dtype=torch.int64, device=DEV)
    bufs = []
    for rank in range(world):
        shard = full[:, rank * vpr:(rank + 1) * vpr]    # a column block: row-strided view
        local = H.sample(shard.contiguous(), zeros, nok, ones, zeros, seed=0)
        bufs.append(H.greedy_pack(shard, local, rank, world, rows + 1))
        assert bufs[-1].shape == (world, rows + 1, 4) and (bufs[-1][1 - rank] == 0).all()
    summed = (bufs[0].float() + bufs[1].float()).to(dtype)
    assert torch.equal(summed[0], bufs[0][0]) and torch.equal(summed[1], bufs[1][1])   # exact
    got = H.greedy_merge(summed, rows)
    want = torch.stack([torch.nonzero(r == r.max())[0, 0] for r in full.float()])
    assert torch.equal(got, want), (got, want)
    assert got.tolist()[1:] == [70000, 151000, 0, 151935]
