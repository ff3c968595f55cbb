// Bodies of the vector paged-decode kernel and of the split-KV combine kernel (K3), shared by paged_decode.hip (one
// launch each) and decode_persistent.hip (phases of the persistent small-batch decode kernel, over virtual workgroups):
// ONE statement of the arithmetic, so the two paths are bit-identical by construction.
#pragma once
#include "paged_decode.h"

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

// LDS of one (virtual) workgroup of paged_decode_body: [NW][HG][G][2] (m, l) pairs, then [NW][HG][G][D] partial rows
template <typename Tag, int D, int G, bool WIDE>
struct DecodeLds {
  static constexpr int HG = WIDE ? 64 / (D / elem<Tag>::VEC) : 1;
  static constexpr int ML_FLOATS = 4 * HG * G * 2, ACC_FLOATS = 4 * HG * G * D;
  static constexpr int BYTES = (ML_FLOATS + ACC_FLOATS) * 4;
};

// The body of paged_decode_kernel for the 256-thread workgroup (bx, by, bz) of its grid; `tid` = thread of that workgroup.
// The kernel below calls it with its blockIdx / threadIdx; the persistent small-batch decode kernel
// (decode_persistent.hip) runs it for VIRTUAL workgroups: `write` false = a slot without work that still has to take
// part in the workgroup barriers (it recomputes a valid workgroup and stores nothing).
template <typename Tag, int D, int G, int NLOAD, bool UNIFORM, bool WIDE, bool ROPE, bool COH = false>
__device__ __forceinline__ void paged_decode_body(const DecodeParams &p, const int bx, const int by, const int bz, const int tid,
                                                  const bool write, float *lds) {
  typedef typename elem<Tag>::T T;
  constexpr int VEC = elem<Tag>::VEC;
  constexpr int LPT = D / VEC;       // lanes per token row
  constexpr int TPL = 64 / LPT;      // tokens (WIDE: kv heads) per wave-wide load
  constexpr int TILE = WIDE ? NLOAD : NLOAD * TPL;  // tokens per wave per iteration
  constexpr int NW = 4;
  constexpr int UNIT = NW * TILE;
  constexpr int HG = WIDE ? TPL : 1;  // kv heads per workgroup
  constexpr float NEG_BIG = -1e30f;

  typedef MV_AS3 float lds_f;
  lds_f (*lds_ml)[HG][G][2] = reinterpret_cast<lds_f (*)[HG][G][2]>((lds_f *)lds);
  lds_f (*lds_acc)[HG][G][D] = reinterpret_cast<lds_f (*)[HG][G][D]>((lds_f *)lds + NW * HG * G * 2);

  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int sub = lane % LPT;  // which 16-byte piece of the row
  const int grp = lane / LPT;  // which token (WIDE: kv head) of a wave-wide load

  const int n_chunks = WIDE ? 1 : (p.g_total + G - 1) / G;  // q-head chunks per kv head
  const int hkv0 = WIDE ? bx * TPL : bx / n_chunks;  // first kv head of this workgroup
  const int hkv = WIDE ? hkv0 + grp : hkv0;                          // this lane's kv head
  const int chunk = WIDE ? 0 : bx - hkv0 * n_chunks;
  const int split = by;
  const int b = bz;
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
    qraw[g] = ld16<COH>(qp);
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
    if (write && slot >= 0 && u_last >= u_begin && u_last < u_end && wave == 0 && (WIDE || grp == 0)) {
      const u32x4 kraw = ld16<COH>((const T *)p.k_new + (int64_t)b * p.k_stride + (int64_t)hkv * D + sub * VEC);
      const u32x4 vraw = ld16<COH>((const T *)p.v_new + (int64_t)b * p.v_stride + (int64_t)hkv * D + sub * VEC);
      const u32x4 krot = norm_rope_row<Tag, D>(kraw, (const T *)p.k_norm_w, p.eps, sub, cs, sn);
      *(MV_AS1 u32x4 *)((T *)p.k_cache + (slot * p.n_kv_heads + hkv) * D + sub * VEC) = krot;
      *(MV_AS1 u32x4 *)((T *)p.v_cache + (slot * p.n_kv_heads + hkv) * D + sub * VEC) = vraw;
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
  const MV_AS1 T *kbase = (const MV_AS1 T *)p.k_cache + (int64_t)hkv * D + sub * VEC;
  const MV_AS1 T *vbase = (const MV_AS1 T *)p.v_cache + (int64_t)hkv * D + sub * VEC;

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
      kraw[i] = __builtin_nontemporal_load((const MV_AS1 u32x4 *)(kbase + row * row_stride));
      vraw[i] = __builtin_nontemporal_load((const MV_AS1 u32x4 *)(vbase + row * row_stride));
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
  for (int idx = tid; idx < HG * G * D; idx += NW * 64) {
    const int hg = idx / (G * D), g = (idx / D) % G, d = idx % D;
    if (g >= g_live || !write) continue;
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
      if constexpr (sizeof(T) == 2) st2<COH>((uint16_t *)p.out + ((int64_t)b * p.n_heads + h) * D + d, elem<Tag>::st(o));
      else ((T *)p.out)[((int64_t)b * p.n_heads + h) * D + d] = elem<Tag>::st(o);
    } else {
      const int64_t slot = ((int64_t)b * p.n_heads + h) * p.nsplit + split;
      stf<COH>(p.ws_acc + slot * D + d, A);
      if (d == 0) {
        stf<COH>(p.ws_ml + slot * 2 + 0, M);
        stf<COH>(p.ws_ml + slot * 2 + 1, L);
      }
    }
  }
}

template <typename Tag, int D, int G, int NLOAD, bool UNIFORM, bool WIDE, bool ROPE = false>
__global__ void __launch_bounds__(256) paged_decode_kernel(DecodeParams p) {
  __shared__ float lds[DecodeLds<Tag, D, G, WIDE>::ML_FLOATS + DecodeLds<Tag, D, G, WIDE>::ACC_FLOATS];
  paged_decode_body<Tag, D, G, NLOAD, UNIFORM, WIDE, ROPE>(p, blockIdx.x, blockIdx.y, blockIdx.z, threadIdx.x, true, lds);
}

// Split-KV combine: one workgroup (128 threads) per (sequence, head).  The partial rows were written by workgroups on other
// XCDs, so every load is a ~2 us round trip and what counts is how many of them are DEPENDENT: the (m, l) pairs go
// through LDS once; then thread (q, c) — q = threadIdx / (D/4) picks every QUARTER-th split, c the 16-byte column piece —
// keeps all its rows (up to 16) in flight at once, and the four partial sums meet in LDS.  (Round 2: one thread per
// output dim, eight 4-byte loads in flight, four rounds at 32 splits: 6.8 us per launch, 14 % of the B = 1 step.)
// LDS of one (virtual) workgroup of paged_decode_reduce_body, in floats: part [NQ][D4] f32x4, then sf [64], then s_l
template <int D>
struct ReduceLds {
  static constexpr int FLOATS = 128 * 4 + 64 + 4;
  static constexpr int BYTES = FLOATS * 4;
};

// body for the 128-thread workgroup `bh` = (sequence, head); see paged_decode_body for `write`
template <typename Tag, int D, bool COH = false>
__device__ __forceinline__ void paged_decode_reduce_body(const float *ws_acc, const float *ws_ml, typename elem<Tag>::T *out,
                                                         const int nsplit, const int64_t bh, const int tid, const bool write,
                                                         float *lds) {
  constexpr int D4 = D / 4, NQ = 128 / D4;    // column pieces; split groups (4 at D = 128, 8 at D = 64)
  constexpr int kMax = 64, PER = kMax / NQ;   // kMaxSplits; rows per thread at most
  typedef MV_AS3 f32x4 lds_f4;
  typedef MV_AS3 float lds_f;
  lds_f4 (*part)[D4] = reinterpret_cast<lds_f4 (*)[D4]>((lds_f4 *)lds);
  lds_f *sf = (lds_f *)lds + 128 * 4;
  lds_f &s_l = ((lds_f *)lds)[128 * 4 + 64];
  const int q = tid / D4, c = tid - q * D4;
  // rows first (they do not depend on the factors), then the pairs: one round trip covers both
  f32x4 a[PER];
  const f32x4 *row = reinterpret_cast<const f32x4 *>(ws_acc + bh * nsplit * D) + c;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int sp = q + u * NQ;
    a[u] = ldf4<COH>(reinterpret_cast<const float *>(row + (int64_t)(sp < nsplit ? sp : 0) * D4));
  }
  float m = -1e30f, l = 0.f;
  if (tid < nsplit) {
    m = ldf<COH>(ws_ml + (bh * nsplit + tid) * 2);
    l = ldf<COH>(ws_ml + (bh * nsplit + tid) * 2 + 1);
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
  if (q == 0 && write) {
#pragma unroll
    for (int g = 1; g < NQ; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) A[e] += part[g][c][e];
    const float L = s_l;
    typename elem<Tag>::T *o = out + bh * D + 4 * c;
    if constexpr (COH && sizeof(typename elem<Tag>::T) == 2) {
      uint32_t r[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = elem<Tag>::st(L > 0.f ? A[e] / L : 0.f);
      st8<true>(o, u32x2{r[0] | (r[1] << 16), r[2] | (r[3] << 16)});
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = elem<Tag>::st(L > 0.f ? A[e] / L : 0.f);
    }
  }
}

template <typename Tag, int D>
__global__ void __launch_bounds__(128) paged_decode_reduce_kernel(const float *__restrict__ ws_acc,
                                                                  const float *__restrict__ ws_ml,
                                                                  typename elem<Tag>::T *__restrict__ out,
                                                                  int nsplit) {
  __shared__ __attribute__((aligned(16))) float lds[ReduceLds<D>::FLOATS];
  paged_decode_reduce_body<Tag, D>(ws_acc, ws_ml, out, nsplit, blockIdx.x, threadIdx.x, true, lds);
}

