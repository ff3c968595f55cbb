"""Seeded random-shape sweeps of the round-2 kernels against the oracle (C-ABI through hip_ops): head layouts, block
sizes, ragged contexts and holes for the paged-decode kernels (both mappings of the vector kernel, the matrix-core
kernel); row counts, ragged N and K for the small-M projections on row-major and fragment-packed weights and their
norm-folded forms.  Sizes are kept small enough for the CPU oracle; every case is reproducible from its seed."""
import math
import random

import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def H():
    from minivllm_amd import hip_ops
    return hip_ops


def _tol(dtype):
    return {torch.float32: dict(rtol=1e-4, atol=1e-5), torch.bfloat16: dict(rtol=2 ** -7, atol=2 ** -7),
            torch.float16: dict(rtol=2 ** -10, atol=2 ** -9)}[dtype]


@pytest.mark.parametrize("seed", range(12))
def test_paged_decode_random_layouts(H, seed):
    rng = random.Random(1000 + seed)
    g = torch.Generator().manual_seed(seed)
    dtype = rng.choice([torch.bfloat16, torch.float16, torch.float32])
    D = rng.choice([64, 128])
    hkv = rng.choice([1, 2, 4, 8])
    grp = rng.choice([1, 2, 3, 4, 7, 8])
    bs = rng.choice([4, 16, 32, 64, 128])
    B = rng.randint(1, 6)
    ctxs = [rng.choice([1, 2, bs - 1, bs, bs + 1, 3 * bs + 5, rng.randint(1, 900)]) for _ in range(B)]
    ctxs = [max(1, c) for c in ctxs]
    nblk = [-(-c // bs) for c in ctxs]
    width = max(nblk) + rng.randint(0, 2)
    nb = sum(nblk) + 3
    perm = torch.randperm(nb, generator=g).tolist()
    table = torch.full((B, width), -1, dtype=torch.int32)
    it = iter(perm)
    for b in range(B):
        for j in range(nblk[b]):
            table[b, j] = next(it)
    if rng.random() < 0.4:                                     # a hole inside some context (contributes nothing)
        b = rng.randrange(B)
        if nblk[b] > 1:
            table[b, rng.randrange(nblk[b])] = -1
    kc = torch.randn(nb, bs, hkv, D, generator=g).to(dtype)
    vc = torch.randn(nb, bs, hkv, D, generator=g).to(dtype)
    q = torch.randn(B, hkv * grp, D, generator=g).to(dtype)
    ctx = torch.tensor(ctxs, dtype=torch.int32)
    scale = 1.0 / math.sqrt(D)
    ref = O.paged_decode(q, kc, vc, table, ctx, scale)
    args = (q.to(DEV), kc.to(DEV), vc.to(DEV), table.to(DEV), ctx.to(DEV), scale)
    for flags in (0, H.MV_DECODE_PER_HEAD, H.MV_DECODE_VECTOR):
        out = H.paged_decode(*args, flags=flags)
        torch.testing.assert_close(out.float().cpu(), ref, msg=lambda m: f"seed {seed} flags {flags} {dtype} D={D} "
                                   f"hkv={hkv} g={grp} bs={bs} ctx={ctxs}: {m}", **_tol(dtype))


@pytest.mark.parametrize("seed", range(10))
def test_small_m_projections_random_shapes(H, seed):
    rng = random.Random(2000 + seed)
    g = torch.Generator().manual_seed(seed)
    dtype = rng.choice([torch.bfloat16, torch.float16])
    m = rng.choice([1, 2, 3, 4, 5, 8, 13, 16])
    k = 64 * rng.randint(4, 60)
    n = 4 * rng.randint(4, 700)
    x = torch.randn(m, k, generator=g).to(dtype).to(DEV)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(dtype).to(DEV)
    b = torch.randn(n, generator=g).to(dtype).to(DEV) if rng.random() < 0.5 else None
    exact = O.linear(x.cpu(), w.cpu(), None if b is None else b.cpu())
    eps = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    bound = eps * exact.abs() + eps * exact.abs().mean()
    wp = H.pack_weight(w)
    for got in (H.linear_small_m(x, w, b), H.linear_small_m(x, w, b, packed=wp)):
        assert ((got.double().cpu() - exact).abs() <= bound).all(), (seed, m, n, k)
    # the norm-folded form where it takes the shape: bit-identical to norm, then projection
    if H.norm_linear_small_m_ok(x, n, wp):
        nw = (1.0 + 0.1 * torch.randn(k, generator=g)).to(dtype).to(DEV)
        res = torch.randn(m, k, generator=g).to(dtype).to(DEV)
        h_ref, r_ref = H.add_rmsnorm(x, res, nw, 1e-6, inplace=False)
        out, r_out = H.norm_linear_small_m(x, res, nw, 1e-6, wp, n, b)
        assert torch.equal(out, H.linear_small_m(h_ref, w, b, packed=wp)) and torch.equal(r_out, r_ref), (seed, m, n, k)


@pytest.mark.parametrize("seed", range(6))
def test_gated_small_m_random_shapes(H, seed):
    rng = random.Random(3000 + seed)
    g = torch.Generator().manual_seed(seed)
    dtype = rng.choice([torch.bfloat16, torch.float16])
    m = rng.choice([1, 2, 4, 7, 16])
    k = 64 * rng.randint(4, 40)
    inter = 16 * rng.randint(2, 300)
    x = torch.randn(m, k, generator=g).to(dtype).to(DEV)
    w = (torch.randn(2 * inter, k, generator=g) * k ** -0.5).to(dtype).to(DEV)
    wp = H.pack_weight(w)
    want = O.gate_up_silu(x.cpu(), w.cpu(), dtype).double()
    eps = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    bound = 4.0 * eps * want.abs() + eps * want.abs().mean()
    for got in (H.gate_up_silu_small_m(x, w), H.gate_up_silu_small_m(x, w, packed=wp)):
        assert ((got.double().cpu() - want).abs() <= bound).all(), (seed, m, inter, k)
    if H.norm_linear_small_m_ok(x, 2 * inter, wp):
        nw = (1.0 + 0.1 * torch.randn(k, generator=g)).to(dtype).to(DEV)
        h_ref = H.rmsnorm(x, nw, 1e-6)
        out, r_out = H.norm_linear_small_m(x, None, nw, 1e-6, wp, inter, None, gated=True)
        assert torch.equal(out, H.gate_up_silu_small_m(h_ref, w, packed=wp)) and r_out.data_ptr() == x.data_ptr()


@pytest.mark.parametrize("seed", range(14))
def test_prefill_mfma_random_shapes(H, seed):
    """The flash prefill kernel with hand-placed LDS reads (round 3): random sequence counts / lengths / head layouts,
    contiguous K/V and the paged (prefix-cached) form with ragged uncached suffixes, 4- and 8-wave workgroups, against
    the oracle; SPIKED scores (one key row of every sequence scaled up, so that a row maximum jumps by far more than the
    deferred-rescale threshold in the middle of a row's tiles: cdna_hip_programming.md §5.4 rule 26) — and every launch
    repeated: a hand-counted wait that is one too small shows up as run-to-run differences."""
    rng = random.Random(3000 + seed)
    g = torch.Generator().manual_seed(seed)
    dtype = rng.choice([torch.bfloat16, torch.float16])
    hkv, grp = rng.choice([(8, 2), (4, 7), (1, 8), (2, 1), (4, 4)])
    paged = seed % 2 == 1
    # every legal engine block size (Config: multiples of 64; reference config.py:178-185) for the paged-prefix form
    hq, D, bs = hkv * grp, 128, ((64, 128, 256)[(seed // 2) % 3] if paged else 64)
    n = rng.randint(1, 4)
    lens = [rng.choice([1, 31, 32, 33, 63, 64, 65, 127, 129, 200, 257, rng.randint(1, 700), rng.randint(300, 1200)]) for _ in range(n)]
    cached = [(rng.randint(0, (ln - 1) // bs) * bs if paged else 0) for ln in lens]
    T = sum(lens)
    q = torch.randn(T, hq, D, generator=g)
    k = torch.randn(T, hkv, D, generator=g)
    v = torch.randn(T, hkv, D, generator=g)
    off = 0
    for ln in lens:                                            # the spike: one key row per sequence, late in it
        if ln > 8:
            k[off + rng.randint(ln // 2, ln - 1)] *= 6.0
        off += ln
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    cu_k = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    ref = O.prefill_varlen(q, k, v, cu_k, cu_k, D ** -0.5)
    tol = {torch.bfloat16: dict(rtol=2 ** -6, atol=2 ** -6), torch.float16: dict(rtol=2 ** -9, atol=2 ** -8)}[dtype]
    if not paged:
        args = (q.to(DEV), k.to(DEV), v.to(DEV), cu_k.to(DEV), cu_k.to(DEV), max(lens), max(lens), D ** -0.5)
        kw = {}
        want = ref
    else:
        nblk = [-(-ln // bs) for ln in lens]
        nb = sum(nblk) + 2
        perm = torch.randperm(nb, generator=g).tolist()
        table = torch.full((n, max(nblk)), -1, dtype=torch.int32)
        kc = torch.zeros(nb, bs, hkv, D, dtype=dtype)
        vc = torch.zeros_like(kc)
        it = iter(perm)
        for i, ln in enumerate(lens):
            base = int(cu_k[i])
            for j in range(nblk[i]):
                blk = next(it)
                table[i, j] = blk
                m = min(bs, ln - j * bs)
                kc[blk, :m] = k[base + j * bs: base + j * bs + m]
                vc[blk, :m] = v[base + j * bs: base + j * bs + m]
        qs = [ln - c for ln, c in zip(lens, cached)]
        cu_q = torch.tensor([0] + list(torch.tensor(qs).cumsum(0)), dtype=torch.int32)
        sel = torch.cat([torch.arange(int(cu_k[i]) + cached[i], int(cu_k[i + 1])) for i in range(n)])
        args = (q[sel].to(DEV), None, None, cu_q.to(DEV), cu_k.to(DEV), max(qs), max(lens), D ** -0.5)
        kw = dict(k_cache=kc.to(DEV), v_cache=vc.to(DEV), block_tables=table.to(DEV))
        want = ref[sel]
    for variant in (1, 2, 5):                                  # 4 waves / 8 waves (hand-placed reads) / 64 rows per wave
        outs = [H.prefill_varlen(*args, flags=variant << 4, **kw) for _ in range(3)]
        torch.testing.assert_close(outs[0].float().cpu(), want, msg=lambda m: f"seed {seed} variant {variant} {dtype} "
                                   f"heads {hq}/{hkv} lens {lens} cached {cached} block {bs}: {m}", **tol)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (seed, variant)
