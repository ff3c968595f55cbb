"""GPU parity at BASELINE.json's full sizes, through size-independent properties (the CPU oracle cannot
finish these shapes in seconds): cross-kernel consistency, causality, permutation / round-trip
invariance, identities at position 0, agreement with the library GEMM.  All calls go through the C-ABI."""
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def H():
    from minivllm_amd import hip_ops

    return hip_ops


def _rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=DEV, dtype=torch.float32) * scale).to(dtype)


# config 4: Qwen2-7B heads, 4 prompts x 4096 tokens (the longest prefill batch of SURVEY §8 d)
def test_prefill_4x4096_causal_and_consistent_with_paged_decode(H):
    dtype, n, S, Hq, Hkv, D, bs = torch.bfloat16, 4, 4096, 28, 4, 128, 64
    T = n * S
    q, k, v = _rand((T, Hq, D), dtype, 1), _rand((T, Hkv, D), dtype, 2), _rand((T, Hkv, D), dtype, 3)
    cu = torch.arange(0, T + 1, S, dtype=torch.int32, device=DEV)
    scale = D ** -0.5
    out = H.prefill_varlen(q, k, v, cu, cu, S, S, scale)
    assert out.shape == (T, Hq, D) and torch.isfinite(out.float()).all()
    # (1) the first token of every sequence attends to itself only: out == its own V row (per kv group)
    first = cu[:-1].long()
    want0 = v[first].repeat_interleave(Hq // Hkv, dim=1)
    torch.testing.assert_close(out[first].float(), want0.float(), rtol=2 ** -7, atol=2 ** -7)
    # (2) causality, bit for bit: rewriting the keys/values of the LAST 1000 positions of each sequence
    # leaves every earlier row untouched
    k2, v2 = k.clone(), v.clone()
    for i in range(n):
        k2[i * S + S - 1000:(i + 1) * S].neg_()
        v2[i * S + S - 1000:(i + 1) * S].mul_(3)
    out2 = H.prefill_varlen(q, k2, v2, cu, cu, S, S, scale)
    keep = torch.ones(T, dtype=torch.bool, device=DEV)
    for i in range(n):
        keep[i * S + S - 1000:(i + 1) * S] = False
    assert torch.equal(out[keep], out2[keep]) and not torch.equal(out[~keep], out2[~keep])
    # (3) the paged-decode kernel is the same mathematics, one query at a time: put K/V into a paged cache
    # (shuffled blocks) and ask it for sampled query rows with context = position + 1
    nblk = S // bs
    perm = torch.randperm(n * nblk, generator=torch.Generator().manual_seed(5)).to(DEV)
    kc = torch.empty(n * nblk, bs, Hkv, D, dtype=dtype, device=DEV)
    vc = torch.empty_like(kc)
    kc[perm] = k.view(n * nblk, bs, Hkv, D)
    vc[perm] = v.view(n * nblk, bs, Hkv, D)
    rows = [(0, 0), (0, 63), (1, 64), (1, 1000), (2, 2047), (2, 2048), (3, 4000), (3, 4095)]
    tables = torch.stack([perm[i * nblk:(i + 1) * nblk] for i, _ in rows]).to(torch.int32)
    ctx = torch.tensor([p + 1 for _, p in rows], dtype=torch.int32, device=DEV)
    qs = torch.stack([q[i * S + p] for i, p in rows])
    dec = H.paged_decode(qs, kc, vc, tables, ctx, scale)
    got = torch.stack([out[i * S + p] for i, p in rows])
    torch.testing.assert_close(got.float(), dec.float(), rtol=2 ** -7, atol=2 ** -7)
    # and one row against the oracle itself
    ref = O.paged_decode(qs[3:4].cpu(), kc.cpu(), vc.cpu(), tables[3:4].cpu(), ctx[3:4].cpu(), scale)
    torch.testing.assert_close(got[3:4].float().cpu(), ref, rtol=2 ** -7, atol=2 ** -7)


# config 3/4: the largest prefill step (16384 tokens) through the fused qk-norm + RoPE + KV-store kernel
def test_rope_store_16384_tokens_round_trip_and_norm(H):
    dtype, T, Hq, Hkv, D, bs = torch.bfloat16, 16384, 16, 8, 128, 64
    from minivllm_amd.layers.rotary_embedding import build_cos_sin

    cos, sin = build_cos_sin(D, 40960, 1e6, None)
    cos, sin = cos.to(DEV), sin.to(DEV)
    qkv = _rand((T, (Hq + 2 * Hkv) * D), dtype, 7)
    q = qkv[:, :Hq * D].view(T, Hq, D)
    k = qkv[:, Hq * D:(Hq + Hkv) * D].view(T, Hkv, D)
    v = qkv[:, (Hq + Hkv) * D:].view(T, Hkv, D)
    pos = torch.arange(T, dtype=torch.int64, device=DEV)
    pos[:64] = 0                                              # position 0: the rotation is the identity
    nb = T // bs
    slots = torch.randperm(nb * bs, generator=torch.Generator().manual_seed(8)).to(torch.int32).to(DEV)
    kc = torch.zeros(nb, bs, Hkv, D, dtype=dtype, device=DEV)
    vc = torch.zeros_like(kc)
    q_rot, k_rot = H.rope_qk_store(q, k, v, pos, cos, sin, None, None, 1e-6, kc, vc, slots, want_k_out=True)
    assert torch.equal(q_rot[:64], q[:64]) and torch.equal(k_rot[:64], k[:64])       # identity, bit-exact
    # scatter -> gather round trip: the cache holds exactly the rotated keys and the raw values
    flat_k, flat_v = kc.view(nb * bs, Hkv, D), vc.view(nb * bs, Hkv, D)
    assert torch.equal(flat_k[slots.long()], k_rot) and torch.equal(flat_v[slots.long()], v)
    # rotations preserve each (i, i + D/2) pair's length (up to the two bf16 roundings)
    def pair_norm(x):
        x = x.float()
        return (x[..., :D // 2] ** 2 + x[..., D // 2:] ** 2).sqrt()
    torch.testing.assert_close(pair_norm(q_rot), pair_norm(q), rtol=2 ** -6, atol=2 ** -6)
    # sampled rows against the oracle
    rows = torch.tensor([0, 100, 5000, T - 1], device=DEV)
    rq, rk = O.rope_qk_store(q[rows].cpu(), k[rows].cpu(), v[rows].cpu(), pos[rows].cpu(), cos.cpu(), sin.cpu(),
                             None, None, 1e-6)[:2]
    torch.testing.assert_close(q_rot[rows].float().cpu(), rq.float(), rtol=2 ** -7, atol=2 ** -7)
    torch.testing.assert_close(k_rot[rows].float().cpu(), rk.float(), rtol=2 ** -7, atol=2 ** -7)


# config 3: 256 rows x 151936 logits
def test_sampler_256x151936_greedy_and_degenerate_filters(H):
    B, V = 256, 151936
    logits = _rand((B, V), torch.bfloat16, 11, scale=4.0)
    logits[5, 777] = logits[5].max() + 1
    logits[5, 150000] = logits[5, 777]                          # tie: the lowest index wins
    zeros, ones = torch.zeros(B, device=DEV), torch.ones(B, device=DEV)
    nok = torch.full((B,), -1, dtype=torch.int64, device=DEV)
    want = torch.stack([torch.nonzero(r == r.max())[0, 0] for r in logits.float()])
    greedy = H.sample(logits, zeros, nok, ones, zeros, seed=1)
    assert torch.equal(greedy, want) and greedy[5] == 777
    # top-k = 1 with any temperature, and a vanishing nucleus, can only return the maximum; a tie is
    # resolved towards either maximal entry
    for kw in (dict(top_ks=torch.ones(B, dtype=torch.int64, device=DEV), top_ps=ones),
               dict(top_ks=nok, top_ps=torch.full((B,), 1e-6, device=DEV))):
        tok = H.sample(logits, ones * 0.7, kw["top_ks"], kw["top_ps"], zeros, seed=2, offset=3)
        picked = logits.gather(1, tok[:, None])[:, 0]
        assert torch.equal(picked, logits.max(dim=1).values)
    # the kept set of top-k = 40 has exactly 40 members (ties at the threshold value are cut to k, like
    # torch.topk + scatter in the reference) and they are the largest
    toks, mask = H.sample(logits, ones, torch.full((B,), 40, dtype=torch.int64, device=DEV), ones, zeros, seed=3,
                          return_mask=True)
    rows = [0, 1, 100, 255]
    for r in rows:
        kept = torch.nonzero(mask[r])[:, 0]
        assert kept.numel() == 40
        row = logits[r].float()
        thr = row.topk(40).values[-1]
        assert (row[kept] >= thr).all() and mask[r][row > thr].all()
        assert mask[r, toks[r]]


# config 3: the LM head of a 16-token decode step, against the library GEMM
def test_lm_head_small_m_matches_library_gemm(H):
    import torch.nn.functional as F

    m, n, k = 16, 151936, 2048
    x, w = _rand((m, k), torch.bfloat16, 21), _rand((n, k), torch.bfloat16, 22, scale=k ** -0.5)
    got, lib = H.linear_small_m(x, w).float(), F.linear(x, w).float()
    # both are fp32 accumulations (in different orders) rounded once: they may differ by one bf16 ulp where
    # the sums straddle a rounding boundary -- one ulp at the output scale (~1) for sums that cancel to ~0
    ulp = 2.0 ** -7 * lib.abs().clamp_min(0.25)   # bf16 spacing is <= 2^-7 |x|
    assert ((got - lib).abs() <= ulp).all()
    assert (got == lib).float().mean() > 0.9
    # linearity in x (exact for a power-of-two factor)
    assert torch.equal(H.linear_small_m(x * 2, w).float(), got * 2)


# config 3 / 5 sampler shape under tensor parallelism: 256 rows x 151 936 logits over 8 emulated ranks
def test_two_stage_draw_at_full_size_is_consistent_with_the_full_row(H):
    """The gather-free draw for the default SamplingParams (mv_mass_pack / mv_mass_merge) at B = 256, V = 151 936, TP = 8,
    through properties that need no second implementation:
      * greedy rows: the token is the full row's argmax,
      * sampled rows: the token is a valid id, and its logit lies inside the numerically reachable set — with the row's
        uniform u the inverse-CDF draw cannot land on a token whose probability is < 2^-24 of the row's mass,
      * the shard masses add up to the full row's mass (logsumexp identity, fp32),
      * the same (seed, offset) gives the same tokens; another offset moves most sampled rows,
      * a row concentrated on one token (logit + 40) always returns that token."""
    dtype, B, V, world = torch.bfloat16, 256, 151936, 8
    vpr = V // world
    full = _rand((B, V), dtype, 11, 2.0)
    spike = torch.arange(0, B, 7, device=DEV)
    spike_tok = (spike * 593 + 17) % V
    full[spike, spike_tok] = 40.0
    temps = torch.ones(B, device=DEV)
    temps[::4] = 0.0                                      # every fourth row greedy
    temps[1::4] = 0.7
    ks = torch.full((B,), -1, dtype=torch.int64, device=DEV)
    ones, zeros = torch.ones(B, device=DEV), torch.zeros(B, device=DEV)

    def draw(offset):
        bufs, masses, maxes = [], [], []
        for r in range(world):
            shard = full[:, r * vpr:(r + 1) * vpr]
            local, stats = H.sample(shard, temps, ks, ones, zeros, seed=21, offset=offset, return_stats=True)
            bufs.append(H.mass_pack(stats, local, r, world, vpr, B, dtype))
            maxes.append(stats[:, 0])
            masses.append(stats[:, 1])
        summed = torch.stack([b.float() for b in bufs]).sum(0).to(dtype)
        return H.mass_merge(summed, B, temps, seed=21, offset=offset), torch.stack(maxes, 1), torch.stack(masses, 1)

    toks, maxes, masses = draw(5)
    assert toks.shape == (B,) and int(toks.min()) >= 0 and int(toks.max()) < V
    greedy = temps == 0
    assert torch.equal(toks[greedy], full[greedy].float().argmax(-1))
    assert torch.equal(toks[spike], spike_tok)                                   # e^-30 of the mass elsewhere
    # shard statistics: sum_r S_r exp((m_r - M)/T) == sum_j exp((l_j - M)/T)
    sampled = ~greedy
    t = temps[sampled][:, None]
    M = maxes[sampled].max(dim=1, keepdim=True).values
    z_shards = (masses[sampled] * torch.exp((maxes[sampled] - M) / t)).sum(1)
    z_full = torch.exp((full[sampled].float() - M) / t).sum(1)
    torch.testing.assert_close(z_shards, z_full, rtol=2e-3, atol=0)
    # the drawn token is not a numerically unreachable one
    logit = full[sampled].float().gather(1, toks[sampled][:, None]).squeeze(1)
    assert bool(((logit - M.squeeze(1)) / t.squeeze(1) > -24 * 0.6931 - 1).all())
    again, _, _ = draw(5)
    other, _, _ = draw(6)
    assert torch.equal(again, toks)
    plain = sampled.clone()
    plain[spike] = False
    assert (other[plain] != toks[plain]).float().mean() > 0.9


def test_nucleus_floor_at_full_size_equals_the_full_row_keep_mask(H):
    """The cross-rank nucleus / min-p floor (mv_shard_key_hist / mv_nucleus_pick) at B = 256, V = 151 936, TP = 8 (ranks
    emulated: the all-reduces are host sums of the int64 histograms), against the one thing that must hold at any size:
    the kept set {x >= floor} of every row IS mv_sample's keep mask on the full row (whose own parity with the reference
    sampler is pinned at fixture size), for peaked and flat rows and a mix of top-p, min-p, both, none and greedy rows;
    plus: integer histograms are reproducible (two evaluations bit-identical) and their level-0 total is the row's mass."""
    dtype, B, V, world = torch.bfloat16, 256, 151936, 8
    vpr = V // world
    full = _rand((B, V), dtype, 12, 1.0)
    full[:128] *= 3.0                                       # peaked half / flat half
    full[128:] *= 0.25
    temps = torch.ones(B, device=DEV)
    temps[::5] = 0.0
    temps[1::5] = 0.6
    top_ps = torch.ones(B, device=DEV)
    top_ps[::2] = 0.9
    top_ps[1::6] = 0.5
    min_ps = torch.zeros(B, device=DEV)
    min_ps[::3] = 0.02
    ks = torch.full((B,), -1, dtype=torch.int64, device=DEV)
    shards = [full[:, r * vpr:(r + 1) * vpr].contiguous() for r in range(world)]
    row_max = torch.stack([s.amax(dim=1).float() for s in shards]).amax(dim=0)

    def floor_of():
        h0 = sum(H.shard_key_hist(s, row_max, temps, None) for s in shards)
        state, prefix = H.nucleus_pick(h0, 0, top_ps, min_ps, temps, row_max, dtype)
        h1 = sum(H.shard_key_hist(s, row_max, temps, prefix) for s in shards)
        return h0, H.nucleus_pick(h1, 1, top_ps, min_ps, temps, row_max, dtype, state, prefix)

    h0, floor = floor_of()
    h0b, floor_b = floor_of()
    assert torch.equal(h0, h0b) and torch.equal(floor, floor_b)
    live = temps > 0
    z = ((full[live].double() - row_max[live].double()[:, None]) / temps[live].double()[:, None]).exp().sum(1)
    torch.testing.assert_close(h0[live].sum(1).double() / 2.0 ** 40, z, rtol=2e-5, atol=0)
    plain = live & (top_ps >= 1) & (min_ps <= 0)
    assert bool((floor[plain] == float("-inf")).all()) and bool((floor[~live] == float("-inf")).all())
    _, mask = H.sample(full, temps, ks, top_ps, min_ps, seed=1, offset=1, return_mask=True)
    mine = full.float() >= floor[:, None]
    bad = ((mine != mask.bool()) & live[:, None]).sum(1)
    assert int(bad.sum()) == 0, (bad.nonzero().flatten().tolist()[:8], int(bad.max()))
