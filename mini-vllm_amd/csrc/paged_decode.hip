// K3 — paged-KV decode attention for gfx950
// (reference: minivllm/models/layers/page_attention.py:35-133 and the
//  flash_attn_with_kvcache call site, layers/attention.py:436-447).
//
// HBM-bound: every K/V byte of a (sequence, kv-head) pair is read exactly
// once, 16 bytes per lane, LPT = D/VEC adjacent lanes covering one token
// row (256 contiguous bytes for bf16 D=128), straight into VGPRs.  One
// workgroup (4 waves) owns one (sequence, kv-head[, head-chunk], split); all
// G query heads of the kv-head are served from the same K/V registers.
// Each LPT-lane group runs an independent online softmax over its tokens
// (no cross-lane traffic in the loop besides the in-row DPP dot-product
// reduction); groups, waves and splits are merged at the end through
// shuffles, LDS and (only when split) an fp32 workspace.
//
// WIDE (n_kv_heads a multiple of TPL = 64 / LPT): the TPL lane groups of a wave take TPL adjacent kv heads of
// ONE token instead of TPL tokens of one kv head, so a wave-wide load is 1 KiB of contiguous cache (a token
// row is n_kv_heads x D contiguous elements) instead of TPL pieces of 256 B at a row stride, a workgroup owns
// (sequence, TPL kv heads, split), and the token -> cache row arithmetic is scalar.  Measured as pure reads
// (tools/dbg/ingest_probe3.hip): 6.85 TB/s against 6.5 TB/s for the per-head mapping.
// (The same idea for the matrix-core kernel — a wave per kv head — was measured SLOWER, 94 vs 88 us on Qwen2-7B
//  heads at B = 256: that kernel is bound by the bytes its 8 resident waves per CU keep in flight, and a wave
//  that walks twice as many tiles serially keeps no more of them in flight.)
#include "paged_decode.h"
#include <cstdlib>


template <typename Tag>
__device__ __forceinline__ float dot16B(const u32x4 &a, const u32x4 &b) {
  float acc = 0.f;
  if constexpr (__is_same(Tag, bf16_tag)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc = __builtin_amdgcn_fdot2_f32_bf16(u2bf2(a[i]),
                                            u2bf2(b[i]), acc, false);
  } else if constexpr (__is_same(Tag, f16_tag)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc = __builtin_amdgcn_fdot2(u2h2(a[i]),
                                   u2h2(b[i]), acc, false);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc = fmaf(u2f(a[i]), u2f(b[i]), acc);
  }
  return acc;
}

template <typename Tag>
__device__ __forceinline__ void unpack_vec(const u32x4 &raw, float (&f)[elem<Tag>::VEC]) {
  if constexpr (elem<Tag>::VEC == 8) {
    unpack8<Tag>(raw, f);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = u2f(raw[i]);
  }
}

// Sum over the LPT adjacent lanes that share one token (LPT in {8,16,32}).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int LPT>
__device__ __forceinline__ float token_sum(float x) {
  x += dpp_mov<0xB1>(x);   // quad_perm [1,0,3,2]  (xor 1)
  x += dpp_mov<0x4E>(x);   // quad_perm [2,3,0,1]  (xor 2)
  x += dpp_mov<0x141>(x);  // row_half_mirror      (other quad of the 8)
  if constexpr (LPT >= 16) x += dpp_mov<0x140>(x);  // row_mirror (other half of the row)
  if constexpr (LPT >= 32) x += __shfl_xor(x, 16, 64);
  return x;
}

template <typename Tag, int D, int G, int NLOAD, bool UNIFORM, bool WIDE, bool ROPE = false>
__global__ void __launch_bounds__(256) paged_decode_kernel(DecodeParams p) {
  typedef typename elem<Tag>::T T;
  constexpr int VEC = elem<Tag>::VEC;
  constexpr int LPT = D / VEC;       // lanes per token row
  constexpr int TPL = 64 / LPT;      // tokens (WIDE: kv heads) per wave-wide load
  constexpr int TILE = WIDE ? NLOAD : NLOAD * TPL;  // tokens per wave per iteration
  constexpr int NW = 4;
  constexpr int UNIT = NW * TILE;
  constexpr int HG = WIDE ? TPL : 1;  // kv heads per workgroup
  constexpr float NEG_BIG = -1e30f;

  __shared__ float lds_ml[NW][HG][G][2];
  __shared__ float lds_acc[NW][HG][G][D];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane % LPT;  // which 16-byte piece of the row
  const int grp = lane / LPT;  // which token (WIDE: kv head) of a wave-wide load

  const int n_chunks = WIDE ? 1 : (p.g_total + G - 1) / G;  // q-head chunks per kv head
  const int hkv0 = WIDE ? blockIdx.x * TPL : blockIdx.x / n_chunks;  // first kv head of this workgroup
  const int hkv = WIDE ? hkv0 + grp : hkv0;                          // this lane's kv head
  const int chunk = WIDE ? 0 : blockIdx.x - hkv0 * n_chunks;
  const int split = blockIdx.y;
  const int b = blockIdx.z;
  const int h0 = hkv * p.g_total + chunk * G;  // first q head of this lane
  const int g_live = min(G, p.g_total - chunk * G);

  int ctx = p.context_lens[b];
  const int64_t cap = (int64_t)p.max_blocks * p.block_size;
  if (ctx > cap) ctx = (int)cap;
  if (ctx < 0) ctx = 0;
  const int units = (ctx + UNIT - 1) / UNIT;
  const int ups = (units + p.nsplit - 1) / p.nsplit;
  const int u_begin = split * ups;
  const int u_end = min(units, u_begin + ups);

  // q fragments: this lane's 16-byte piece of each of the G heads
  u32x4 qraw[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int gg = g < g_live ? g : 0;
    const T *qp = (const T *)p.q + (int64_t)b * p.q_stride + (int64_t)(h0 + gg) * D + sub * VEC;
    qraw[g] = *reinterpret_cast<const u32x4 *>(qp);
  }
  if constexpr (ROPE) {
    static_assert(!ROPE || elem<Tag>::VEC == 8, "rope fusion: 16-bit dtypes");
    // rotary factors of this sequence's position for this lane's dims (rope_qk_store_kernel's table lookup)
    const int64_t pos = p.positions[b];
    const bool pos_ok = pos >= 0 && pos < p.max_pos;
    float cs[VEC], sn[VEC];
    const int i0 = (sub % (LPT / 2)) * VEC;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      cs[j] = pos_ok ? p.cos[pos * (D / 2) + i0 + j] : 1.0f;
      sn[j] = pos_ok ? p.sin[pos * (D / 2) + i0 + j] : 0.0f;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) qraw[g] = norm_rope_row<Tag, D>(qraw[g], (const T *)p.q_norm_w, p.eps, sub, cs, sn);
    // the workgroup that will read the sequence's last token writes its K / V rows first (one wave; WIDE: its four
    // lane groups are the four kv heads, else lane group 0 has the workgroup's kv head)
    int64_t slot = p.slot_mapping ? (int64_t)p.slot_mapping[b] : -1;
    if (slot >= p.num_blocks * p.block_size) slot = -1;
    const int u_last = ctx > 0 ? (ctx - 1) / UNIT : -1;
    if (slot >= 0 && u_last >= u_begin && u_last < u_end && wave == 0 && (WIDE || grp == 0)) {
      const u32x4 kraw = *reinterpret_cast<const u32x4 *>((const T *)p.k_new + (int64_t)b * p.k_stride + (int64_t)hkv * D + sub * VEC);
      const u32x4 vraw = *reinterpret_cast<const u32x4 *>((const T *)p.v_new + (int64_t)b * p.v_stride + (int64_t)hkv * D + sub * VEC);
      const u32x4 krot = norm_rope_row<Tag, D>(kraw, (const T *)p.k_norm_w, p.eps, sub, cs, sn);
      *reinterpret_cast<u32x4 *>((T *)p.k_cache + (slot * p.n_kv_heads + hkv) * D + sub * VEC) = krot;
      *reinterpret_cast<u32x4 *>((T *)p.v_cache + (slot * p.n_kv_heads + hkv) * D + sub * VEC) = vraw;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the rows are in L2 before any wave of the workgroup asks
    }
    __syncthreads();
  }

  float m[G], l[G], acc[G][VEC];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    m[g] = NEG_BIG;
    l[g] = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[g][j] = 0.f;
  }

  const int32_t *table = p.block_tables + (int64_t)b * p.table_stride;
  const int64_t row_stride = (int64_t)p.n_kv_heads * D;  // elements between token rows
  const T *kbase = (const T *)p.k_cache + (int64_t)hkv * D + sub * VEC;
  const T *vbase = (const T *)p.v_cache + (int64_t)hkv * D + sub * VEC;

  // one tile = NLOAD wave-wide loads of K and of V
  auto load_tile = [&](int u, u32x4 (&kraw)[NLOAD], u32x4 (&vraw)[NLOAD], bool (&valid)[NLOAD]) {
    const int tile_base = (u * NW + wave) * TILE;
    int64_t uni_row = -1;
    if constexpr (UNIFORM) {  // whole tile inside one KV block
      const int bi = p.bs_shift >= 0 ? (tile_base >> p.bs_shift) : (tile_base / p.block_size);
      const int blk = bi < p.max_blocks ? __builtin_amdgcn_readfirstlane(table[bi]) : -1;
      if (blk >= 0 && blk < p.num_blocks)
        uni_row = (int64_t)blk * p.block_size + (tile_base - bi * p.block_size);
    }
    // Loads are unconditional: a token past the context (or in a hole of the table) re-reads the
    // sequence's token 0 / the tile's first token — finite data, masked to probability 0 below.
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int off = WIDE ? i : i * TPL + grp;  // token of this lane within the tile
      const int tok = tile_base + off;
      int64_t row;
      bool ok = tok < ctx;
      if constexpr (UNIFORM) {
        ok = ok && uni_row >= 0;
        row = (uni_row >= 0 ? uni_row : 0) + (ok ? off : 0);
      } else {
        const int tk = ok ? tok : tile_base;  // tile_base < ctx
        const int bi = p.bs_shift >= 0 ? (tk >> p.bs_shift) : (tk / p.block_size);
        const int blk = bi < p.max_blocks ? table[bi] : -1;
        const bool blk_ok = blk >= 0 && blk < p.num_blocks;
        ok = ok && blk_ok;
        row = blk_ok ? (int64_t)blk * p.block_size + (tk - bi * p.block_size) : 0;
      }
      valid[i] = ok;
      // K/V rows are read exactly once per step and 30 GB never fit the 256 MiB Infinity Cache:
      // non-temporal loads keep them from evicting the weights / activations that do get re-used
      kraw[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(kbase + row * row_stride));
      vraw[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(vbase + row * row_stride));
    }
  };

  auto consume_tile = [&](const u32x4 (&kraw)[NLOAD], const u32x4 (&vraw)[NLOAD], const bool (&valid)[NLOAD]) {
    // scores (log2 domain) + online softmax, per query head
    float pr[G][NLOAD];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float mx = m[g];
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) {
        float s = token_sum<LPT>(dot16B<Tag>(qraw[g], kraw[i])) * p.scale_log2;
        s = valid[i] ? s : -INFINITY;
        pr[g][i] = s;
        mx = fmaxf(mx, s);
      }
      const float alpha = __builtin_amdgcn_exp2f(m[g] - mx);
      m[g] = mx;
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) {
        const float e = __builtin_amdgcn_exp2f(pr[g][i] - mx);
        pr[g][i] = e;
        ps += e;
      }
      l[g] = l[g] * alpha + ps;
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[g][j] *= alpha;
    }
    // PV
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      float vf[VEC];
      // row 0 of a foreign block may hold anything (0 * NaN)
      unpack_vec<Tag>(valid[i] ? vraw[i] : u32x4{0u, 0u, 0u, 0u}, vf);
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[g][j] = fmaf(pr[g][i], vf[j], acc[g][j]);
    }
  };

  // a tile at or past the context is skipped whole (wave-uniform)
  auto live = [&](int u) { return u < u_end && (u * NW + wave) * TILE < ctx; };
  // (round 2: a second register set with the next tile's loads in flight during the arithmetic — 256 VGPRs,
  //  2 waves per SIMD — measured no faster than this loop, 166 vs 165 us: the HBM side is the limit)
  for (int u = u_begin; live(u); ++u) {
    u32x4 kraw[NLOAD], vraw[NLOAD];
    bool valid[NLOAD];
    load_tile(u, kraw, vraw, valid);
    consume_tile(kraw, vraw, valid);
  }

  if constexpr (WIDE) {
    // lane groups hold different kv heads: nothing to merge inside the wave
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (sub == 0) {
        lds_ml[wave][grp][g][0] = m[g];
        lds_ml[wave][grp][g][1] = l[g];
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) lds_acc[wave][grp][g][sub * VEC + j] = acc[g][j];
    }
  } else {
  // merge the TPL lane groups of this wave
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float mm = m[g];
    for (int off = LPT; off < 64; off <<= 1) mm = fmaxf(mm, __shfl_xor(mm, off, 64));
    const float f = __builtin_amdgcn_exp2f(m[g] - mm);
    float ll = l[g] * f;
    for (int off = LPT; off < 64; off <<= 1) ll += __shfl_xor(ll, off, 64);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float a = acc[g][j] * f;
      for (int off = LPT; off < 64; off <<= 1) a += __shfl_xor(a, off, 64);
      acc[g][j] = a;
    }
    if (grp == 0) {
      if (sub == 0) {
        lds_ml[wave][0][g][0] = mm;
        lds_ml[wave][0][g][1] = ll;
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) lds_acc[wave][0][g][sub * VEC + j] = acc[g][j];
    }
  }
  }
  __syncthreads();

  // merge the 4 waves; write the output row or this split's partial
  for (int idx = threadIdx.x; idx < HG * G * D; idx += blockDim.x) {
    const int hg = idx / (G * D), g = (idx / D) % G, d = idx % D;
    if (g >= g_live) continue;
    float M = lds_ml[0][hg][g][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, lds_ml[w][hg][g][0]);
    float L = 0.f, A = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float f = __builtin_amdgcn_exp2f(lds_ml[w][hg][g][0] - M);
      L += lds_ml[w][hg][g][1] * f;
      A += lds_acc[w][hg][g][d] * f;
    }
    const int h = (hkv0 + hg) * p.g_total + chunk * G + g;
    if (p.nsplit == 1) {
      const float o = L > 0.f ? A / L : 0.f;
      ((T *)p.out)[((int64_t)b * p.n_heads + h) * D + d] = elem<Tag>::st(o);
    } else {
      const int64_t slot = ((int64_t)b * p.n_heads + h) * p.nsplit + split;
      p.ws_acc[slot * D + d] = A;
      if (d == 0) {
        p.ws_ml[slot * 2 + 0] = M;
        p.ws_ml[slot * 2 + 1] = L;
      }
    }
  }
}

// Split-KV combine: one workgroup (128 threads) per (sequence, head).  The partial rows were written by workgroups on other
// XCDs, so every load is a ~2 us round trip and what counts is how many of them are DEPENDENT: the (m, l) pairs go
// through LDS once; then thread (q, c) — q = threadIdx / (D/4) picks every QUARTER-th split, c the 16-byte column piece —
// keeps all its rows (up to 16) in flight at once, and the four partial sums meet in LDS.  (Round 2: one thread per
// output dim, eight 4-byte loads in flight, four rounds at 32 splits: 6.8 us per launch, 14 % of the B = 1 step.)
template <typename Tag, int D>
__global__ void __launch_bounds__(128) paged_decode_reduce_kernel(const float *__restrict__ ws_acc,
                                                                  const float *__restrict__ ws_ml,
                                                                  typename elem<Tag>::T *__restrict__ out,
                                                                  int nsplit) {
  constexpr int D4 = D / 4, NQ = 128 / D4;    // column pieces; split groups (4 at D = 128, 8 at D = 64)
  constexpr int kMax = 64, PER = kMax / NQ;   // kMaxSplits; rows per thread at most
  __shared__ float sf[kMax];
  __shared__ float s_l;
  __shared__ f32x4 part[NQ][D4];
  const int64_t bh = blockIdx.x;
  const int tid = threadIdx.x, q = tid / D4, c = tid - q * D4;
  // rows first (they do not depend on the factors), then the pairs: one round trip covers both
  f32x4 a[PER];
  const f32x4 *row = reinterpret_cast<const f32x4 *>(ws_acc + bh * nsplit * D) + c;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int sp = q + u * NQ;
    a[u] = row[(int64_t)(sp < nsplit ? sp : 0) * D4];
  }
  float m = -1e30f, l = 0.f;
  if (tid < nsplit) {
    m = ws_ml[(bh * nsplit + tid) * 2];
    l = ws_ml[(bh * nsplit + tid) * 2 + 1];
  }
  // maximum over the splits (<= 64: the first wave holds them all), factors, the sum of l * f
  if (tid < 64) {
    float M = m;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
    const float f = tid < nsplit ? __builtin_amdgcn_exp2f(m - M) : 0.f;
    sf[tid] = f;
    float L = l * f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) L += __shfl_xor(L, off, 64);
    if (tid == 0) s_l = L;
  }
  __syncthreads();
  f32x4 A = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int sp = q + u * NQ;
    const float f = sp < nsplit ? sf[sp] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) A[e] += a[u][e] * f;
  }
  part[q][c] = A;
  __syncthreads();
  if (q == 0) {
#pragma unroll
    for (int g = 1; g < NQ; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) A[e] += part[g][c][e];
    const float L = s_l;
    typename elem<Tag>::T *o = out + bh * D + 4 * c;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = elem<Tag>::st(L > 0.f ? A[e] / L : 0.f);
  }
}

// ---- host side -------------------------------------------------------------
namespace {

struct DecodePlan {
  int g_chunk;  // q heads served per workgroup (template G)
  int n_chunks;
  int nload;
  int tile;
  int unit;
  int nsplit;
  bool uniform;
  bool mfma;  // tensor-core variant (paged_decode_mfma.hip)
  bool wide;  // vector kernel, TPL adjacent kv heads per workgroup (whole-row loads)
};

constexpr int kTargetWorkgroups = 2048;  // ~8 workgroups per CU
constexpr int kMaxSplits = 64;
constexpr int kMfmaMinGroup = 4;

int pick_g(int g_total) {
  if (g_total <= 8) return g_total;
  // widest chunk <= 8 that divides evenly, else 8 with a ragged tail
  for (int g = 8; g >= 4; --g)
    if (g_total % g == 0) return g;
  return 8;
}

bool make_plan(int batch, int n_heads, int n_kv_heads, int head_dim, int block_size,
               int max_blocks, int dtype, int flags, DecodePlan *pl) {
  if (n_kv_heads <= 0 || n_heads % n_kv_heads) return false;
  if (head_dim != 64 && head_dim != 128) return false;
  const int g_total = n_heads / n_kv_heads;
  int kv_groups = n_kv_heads;  // workgroups along the kv-head axis
  pl->wide = false;
  // >= 4 query heads per kv head: QK^T and PV go to the matrix cores (the vector path spends
  // ~G x 20 VALU instructions per key and drops to 2-3.7 TB/s at G = 7..8)
  const int min_group = (flags & MV_DECODE_VECTOR) ? 1 << 30 : ((flags & MV_DECODE_MFMA) ? 1 : kMfmaMinGroup);
  pl->mfma = dtype != MV_F32 && head_dim == 128 && g_total >= min_group && block_size % 32 == 0;
  if (pl->mfma) {
    pl->g_chunk = g_total < 16 ? g_total : 16;
    pl->n_chunks = (g_total + 15) / 16;
    pl->nload = 8;
    pl->tile = 32;
    pl->unit = 128;
    pl->uniform = true;
  } else {
    pl->g_chunk = pick_g(g_total);
    pl->n_chunks = (g_total + pl->g_chunk - 1) / pl->g_chunk;
    const int vec = dtype == MV_F32 ? 4 : 8;
    const int tpl = 64 / (head_dim / vec);
    pl->nload = pl->g_chunk <= 4 ? 8 : 4;
    pl->wide = pl->n_chunks == 1 && pl->g_chunk <= 4 && n_kv_heads % tpl == 0 && !(flags & MV_DECODE_PER_HEAD);
    pl->tile = pl->wide ? pl->nload : pl->nload * tpl;
    pl->unit = 4 * pl->tile;
    pl->uniform = block_size % pl->tile == 0;
    if (pl->wide) kv_groups = n_kv_heads / tpl;
  }
  const int64_t max_ctx = (int64_t)max_blocks * block_size;
  const int max_units = (int)((max_ctx + pl->unit - 1) / pl->unit);
  const int64_t wg = (int64_t)batch * kv_groups * pl->n_chunks;
  // the matrix-core kernel keeps 4 workgroups per CU resident (LDS): one wave of 1024 is enough
  // whole-row loads: HBM likes FEWER concurrent streams (B = 256 x 1024 tokens: 162 us at 512 workgroups,
  // 170 us at 1024, 182 us at 2560; the per-head mapping: 171 us at its best)
  int target = pl->wide ? kTargetWorkgroups / 4 : (pl->mfma ? kTargetWorkgroups / 2 : kTargetWorkgroups);
  // per-head vector mapping with >= 512 workgroups already (e.g. the 4 q / 2 kv heads of a TP = 4 shard at B = 256): no
  // split — 45 us against 50 us with the 4-way split and its combine launch
  if (!pl->mfma && !pl->wide && wg >= 512) target = (int)(wg < (1 << 20) ? wg : (1 << 20));
  if ((flags >> 8) & 0xff) target = ((flags >> 8) & 0xff) * 16;  // tuning aid: bits 8-15 = target workgroups / 16
  int ns = wg > 0 ? (int)((target + wg - 1) / wg) : 1;
  if (ns > max_units) ns = max_units;
  if (ns > kMaxSplits) ns = kMaxSplits;
  if (ns < 1) ns = 1;
  pl->nsplit = ns;
  return true;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

template <typename Tag, int D, int G, int NLOAD, bool ROPE>
void launch_gr(const DecodeParams &p, const DecodePlan &pl, int batch, hipStream_t stream) {
  dim3 block(256);
  if constexpr (G <= 4) {
    if (pl.wide) {
      dim3 grid(p.n_kv_heads / (64 / (D / elem<Tag>::VEC)), pl.nsplit, batch);
      if (pl.uniform)
        hipLaunchKernelGGL((paged_decode_kernel<Tag, D, G, NLOAD, true, true, ROPE>), grid, block, 0, stream, p);
      else
        hipLaunchKernelGGL((paged_decode_kernel<Tag, D, G, NLOAD, false, true, ROPE>), grid, block, 0, stream, p);
      return;
    }
  }
  dim3 grid(p.n_kv_heads * pl.n_chunks, pl.nsplit, batch);
  if (pl.uniform)
    hipLaunchKernelGGL((paged_decode_kernel<Tag, D, G, NLOAD, true, false, ROPE>), grid, block, 0, stream, p);
  else
    hipLaunchKernelGGL((paged_decode_kernel<Tag, D, G, NLOAD, false, false, ROPE>), grid, block, 0, stream, p);
}

template <typename Tag, int D, int G, int NLOAD>
void launch_g(const DecodeParams &p, const DecodePlan &pl, int batch, hipStream_t stream) {
  if constexpr (elem<Tag>::VEC == 8 && G <= 4) {   // (rope fusion: 16-bit dtypes, up to 4 query heads per kv head)
    if (p.positions != nullptr) return launch_gr<Tag, D, G, NLOAD, true>(p, pl, batch, stream);
  }
  launch_gr<Tag, D, G, NLOAD, false>(p, pl, batch, stream);
}

template <typename Tag, int D>
int launch_d(const DecodeParams &p, const DecodePlan &pl, int batch, int dtype, int flags, hipStream_t stream) {
  if (pl.mfma) {
    const int klds = (flags & MV_DECODE_KLDS_ON) ? 1 : ((flags & MV_DECODE_KLDS_OFF) ? 0 : -1);
    const int rc = mv_paged_decode_mfma_launch(p, pl.n_chunks, pl.nsplit, batch, dtype, klds, stream);
    if (rc != MV_OK) return rc;
    if (pl.nsplit > 1) {
      hipLaunchKernelGGL((paged_decode_reduce_kernel<Tag, D>), dim3(batch * p.n_heads), dim3(128), 0,
                         stream, p.ws_acc, p.ws_ml, (typename elem<Tag>::T *)p.out, pl.nsplit);
      MV_LAUNCH_CHECK();
    }
    return MV_OK;
  }
  switch (pl.g_chunk) {
    case 1: launch_g<Tag, D, 1, 8>(p, pl, batch, stream); break;
    case 2: launch_g<Tag, D, 2, 8>(p, pl, batch, stream); break;
    case 3: launch_g<Tag, D, 3, 8>(p, pl, batch, stream); break;
    case 4: launch_g<Tag, D, 4, 8>(p, pl, batch, stream); break;
    case 5: launch_g<Tag, D, 5, 4>(p, pl, batch, stream); break;
    case 6: launch_g<Tag, D, 6, 4>(p, pl, batch, stream); break;
    case 7: launch_g<Tag, D, 7, 4>(p, pl, batch, stream); break;
    case 8: launch_g<Tag, D, 8, 4>(p, pl, batch, stream); break;
    default: return MV_E_SHAPE;
  }
  MV_LAUNCH_CHECK();
  if (pl.nsplit > 1) {
    hipLaunchKernelGGL((paged_decode_reduce_kernel<Tag, D>), dim3(batch * p.n_heads), dim3(128), 0,
                       stream, p.ws_acc, p.ws_ml, (typename elem<Tag>::T *)p.out, pl.nsplit);
    MV_LAUNCH_CHECK();
  }
  return MV_OK;
}

}  // namespace

extern "C" size_t mv_paged_decode_workspace_bytes(int batch, int n_heads, int n_kv_heads,
                                                  int head_dim, int block_size, int max_blocks, int flags) {
  // sized for the widest element type so one buffer serves every dtype
  DecodePlan a, b;
  if (batch <= 0 || !make_plan(batch, n_heads, n_kv_heads, head_dim, block_size, max_blocks, MV_BF16, flags, &a) ||
      !make_plan(batch, n_heads, n_kv_heads, head_dim, block_size, max_blocks, MV_F32, flags, &b))
    return 256;
  const int ns = a.nsplit > b.nsplit ? a.nsplit : b.nsplit;
  if (ns <= 1) return 256;
  const size_t slots = (size_t)batch * n_heads * ns;
  return align256(slots * head_dim * sizeof(float)) + align256(slots * 2 * sizeof(float));
}

namespace {
struct RopeArgs {
  const void *k_new, *v_new;
  int64_t k_stride, v_stride;
  const int64_t *positions;
  const float *cos, *sin;
  int64_t max_pos;
  const void *q_norm_w, *k_norm_w;
  float eps;
  const int32_t *slot_mapping;
};
int paged_decode_impl(const void *q, int64_t q_stride_tok, const void *k_cache,
                      const void *v_cache, const int32_t *block_tables,
                      int64_t table_stride, const int32_t *context_lens, void *out,
                      int batch, int n_heads, int n_kv_heads, int head_dim,
                      int block_size, int max_blocks, int64_t num_blocks, float scale,
                      void *workspace, size_t ws_bytes, int dtype, int flags, void *stream, const RopeArgs *rope);
}  // namespace

extern "C" int mv_paged_decode(const void *q, int64_t q_stride_tok, const void *k_cache,
                               const void *v_cache, const int32_t *block_tables,
                               int64_t table_stride, const int32_t *context_lens, void *out,
                               int batch, int n_heads, int n_kv_heads, int head_dim,
                               int block_size, int max_blocks, int64_t num_blocks, float scale,
                               void *workspace, size_t ws_bytes, int dtype, int flags, void *stream) {
  return paged_decode_impl(q, q_stride_tok, k_cache, v_cache, block_tables, table_stride, context_lens, out, batch,
                           n_heads, n_kv_heads, head_dim, block_size, max_blocks, num_blocks, scale, workspace, ws_bytes,
                           dtype, flags, stream, nullptr);
}

// K7 + K6 (per head) + K1 + K3 in ONE launch for decode steps: q / k / v are the RAW rows of the qkv projection; the
// attention kernel norms and rotates q itself, and the workgroup that reads a sequence's last token first writes that
// token's K (normed, rotated) and V into the cache (see paged_decode_kernel<ROPE>).  Bit-identical to
// mv_rope_qk_store followed by mv_paged_decode.  MV_E_SHAPE for layouts the fused form does not take (fp32; more than
// 16 query heads per kv head): the caller then issues the two launches.
extern "C" int mv_rope_paged_decode(const void *q, int64_t q_stride_tok, const void *k, int64_t k_stride_tok,
                                    const void *v, int64_t v_stride_tok, const int64_t *positions, const float *cos,
                                    const float *sin, int64_t max_pos, const void *q_norm_w, const void *k_norm_w,
                                    float eps, const int32_t *slot_mapping, void *k_cache, void *v_cache,
                                    const int32_t *block_tables, int64_t table_stride, const int32_t *context_lens,
                                    void *out, int batch, int n_heads, int n_kv_heads, int head_dim, int block_size,
                                    int max_blocks, int64_t num_blocks, float scale, void *workspace, size_t ws_bytes,
                                    int dtype, int flags, void *stream) {
  if (batch == 0) return MV_OK;
  if (!k || !v || !positions || !cos || !sin || !slot_mapping) return MV_E_NULLPTR;
  if (dtype != MV_BF16 && dtype != MV_F16) return dtype == MV_F32 ? MV_E_SHAPE : MV_E_DTYPE;
  if (max_blocks == 0 || num_blocks == 0 || n_kv_heads <= 0 || n_heads % n_kv_heads) return MV_E_SHAPE;
  if (!mv_aligned16(k) || !mv_aligned16(v) || (k_stride_tok & 7) || (v_stride_tok & 7) ||
      (q_norm_w && !mv_aligned16(q_norm_w)) || (k_norm_w && !mv_aligned16(k_norm_w)))
    return MV_E_ALIGN;
  DecodePlan pl;
  if (!make_plan(batch, n_heads, n_kv_heads, head_dim, block_size, max_blocks, dtype, flags, &pl)) return MV_E_SHAPE;
  // vector kernel: up to 4 query heads per kv head; matrix-core kernel: one chunk of <= 16 heads
  if (pl.n_chunks != 1) return MV_E_SHAPE;
  if (pl.mfma ? head_dim != 128 : pl.g_chunk > 4) return MV_E_SHAPE;
  const RopeArgs rope{k, v, k_stride_tok, v_stride_tok, positions, cos, sin, max_pos, q_norm_w, k_norm_w, eps, slot_mapping};
  return paged_decode_impl(q, q_stride_tok, k_cache, v_cache, block_tables, table_stride, context_lens, out, batch,
                           n_heads, n_kv_heads, head_dim, block_size, max_blocks, num_blocks, scale, workspace, ws_bytes,
                           dtype, flags, stream, &rope);
}

namespace {
int paged_decode_impl(const void *q, int64_t q_stride_tok, const void *k_cache,
                      const void *v_cache, const int32_t *block_tables,
                      int64_t table_stride, const int32_t *context_lens, void *out,
                      int batch, int n_heads, int n_kv_heads, int head_dim,
                      int block_size, int max_blocks, int64_t num_blocks, float scale,
                      void *workspace, size_t ws_bytes, int dtype, int flags, void *stream, const RopeArgs *rope) {
  if (batch == 0) return MV_OK;
  if (!q || !out || !context_lens) return MV_E_NULLPTR;
  if (batch < 0 || n_heads <= 0 || block_size <= 0 || max_blocks < 0 || num_blocks < 0)
    return MV_E_SHAPE;
  if (dtype != MV_BF16 && dtype != MV_F16 && dtype != MV_F32) return MV_E_DTYPE;
  const int esz = dtype == MV_F32 ? 4 : 2;
  if (max_blocks == 0 || num_blocks == 0) {
    // reference: zero-column block tables (warm-up) give zeros, page_attention.py:62-65
    return mv_check_hip(hipMemsetAsync(out, 0, (size_t)batch * n_heads * head_dim * esz,
                                       (hipStream_t)stream));
  }
  if (!k_cache || !v_cache || !block_tables) return MV_E_NULLPTR;
  DecodePlan pl;
  if (!make_plan(batch, n_heads, n_kv_heads, head_dim, block_size, max_blocks, dtype, flags, &pl))
    return MV_E_SHAPE;
  if (!mv_aligned16(q) || !mv_aligned16(k_cache) || !mv_aligned16(v_cache) || !mv_aligned16(out) ||
      (q_stride_tok * esz) % 16)
    return MV_E_ALIGN;

  DecodeParams p;
  p.q = q;
  p.k_cache = k_cache;
  p.v_cache = v_cache;
  p.block_tables = block_tables;
  p.context_lens = context_lens;
  p.out = out;
  p.q_stride = q_stride_tok;
  p.table_stride = table_stride;
  p.num_blocks = num_blocks;
  p.n_heads = n_heads;
  p.n_kv_heads = n_kv_heads;
  p.g_total = n_heads / n_kv_heads;
  p.block_size = block_size;
  p.bs_shift = (block_size & (block_size - 1)) == 0 ? __builtin_ctz((unsigned)block_size) : -1;
  p.max_blocks = max_blocks;
  p.nsplit = pl.nsplit;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.ws_acc = nullptr;
  p.ws_ml = nullptr;
  if (pl.nsplit > 1) {
    const size_t slots = (size_t)batch * n_heads * pl.nsplit;
    const size_t acc_bytes = align256(slots * head_dim * sizeof(float));
    const size_t need = acc_bytes + align256(slots * 2 * sizeof(float));
    if (!workspace) return MV_E_NULLPTR;
    if (ws_bytes < need || !mv_aligned16(workspace)) return MV_E_WORKSPACE;
    p.ws_acc = (float *)workspace;
    p.ws_ml = (float *)((char *)workspace + acc_bytes);
  }
  if (rope) {
    p.k_new = rope->k_new, p.v_new = rope->v_new, p.k_stride = rope->k_stride, p.v_stride = rope->v_stride;
    p.positions = rope->positions, p.cos = rope->cos, p.sin = rope->sin, p.max_pos = rope->max_pos;
    p.q_norm_w = rope->q_norm_w, p.k_norm_w = rope->k_norm_w, p.eps = rope->eps, p.slot_mapping = rope->slot_mapping;
  }
  hipStream_t st = (hipStream_t)stream;
  if (head_dim == 128) {
    MV_DISPATCH_DTYPE(dtype, return (launch_d<Tag, 128>(p, pl, batch, dtype, flags, st)));
  } else {
    MV_DISPATCH_DTYPE(dtype, return (launch_d<Tag, 64>(p, pl, batch, dtype, flags, st)));
  }
  return MV_OK;
}
}  // namespace
