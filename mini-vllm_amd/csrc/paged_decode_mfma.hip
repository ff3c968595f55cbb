// K3 (matrix-core variant) — paged-KV decode attention for GQA groups of >= 4 query heads
// (Qwen2-7B: 7 per kv head, Qwen2-72B @ TP8: 8), bf16/f16, D = 128.
// Reference: minivllm/models/layers/page_attention.py:35-133, layers/attention.py:436-447.
//
// Still HBM-bound and still "every K/V byte once", but the per-key arithmetic leaves the vector
// ALU: with G query heads sharing a kv head the dot products are a [G x D] x [D x keys] product.
// One workgroup (4 waves) owns one (sequence, kv head, <=16 query heads, split); each wave walks
// 32-key tiles on v_mfma_f32_16x16x32:
//   S^T[key][q] = K Q^T     A = K, either straight from HBM into the A-operand layout (16 keys x 64 B per
//                           load instruction; small grids) or, KLDS, by LDS-DMA in whole 256-B rows
//                           (swizzle on the source address) + ds_read_b128; B = Q^T resident in registers
//   online softmax          a query head lives on lanes {q, q+16, q+32, q+48}: two shuffles per tile
//   O^T[d][q] += V^T P^T    V is staged through a per-wave 8 KiB LDS tile (coalesced 256-B rows in,
//                           ds_read_b64_tr_b16 transposed reads out); P^T is the S^T accumulator
//                           reused in place (its key index is the register index)
// No workgroup barrier in the loop: a wave only reads the LDS rows it wrote itself.
#include <stdlib.h>

#include "paged_decode.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int D = 128;
constexpr int TILE = 32;  // keys per wave per step
constexpr int NW = 4;
constexpr int UNIT = NW * TILE;
constexpr int ROW_BYTES = D * 2;
constexpr int V_TILE_BYTES = TILE * ROW_BYTES;       // 8 KiB per wave
constexpr int MERGE_STRIDE = D + 2;                  // acc[D], m, l
constexpr int SMEM_BYTES = NW * 16 * MERGE_STRIDE * 4;  // 33,280 B >= NW * V_TILE_BYTES
constexpr int SMEM_BYTES_KLDS = 2 * NW * V_TILE_BYTES;  // K tile + V tile per wave: 64 KiB

template <typename Tag>
__device__ __forceinline__ f32x4 mfma16(const u32x4 &a, const u32x4 &b, const f32x4 &c) {
  if constexpr (__is_same(Tag, bf16_tag))
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <typename Tag>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  // one v_cvt_pk_{bf16,f16}_f32 (round to nearest even), not two converts + a merge
  if constexpr (__is_same(Tag, bf16_tag))
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2_t));
  else
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, f16x2_t));
}

// LDS row of key kk (0..31) of a tile: the four 4-key blocks of a 16-key half are placed so that the
// two blocks one 32-lane half reads transposed sit 8 rows apart (conflict free, guide T10).
__device__ __forceinline__ int v_lrow(int kk) {
  const int blk = (kk >> 2) & 3;
  return (kk & 16) + 8 * (blk & 1) + 4 * (blk >> 1) + (kk & 3);
}
__device__ __forceinline__ int v_off(int lrow, int ch) {
  return lrow * ROW_BYTES + ((ch ^ (((lrow & 3) << 2) | ((lrow >> 2) & 3))) << 4);
}

// KLDS: K also goes through LDS (DMA in whole 256-B rows, XOR-swizzled on the source side, A operand by
// ds_read_b128) instead of 16 rows x 64 B per load instruction straight into the A-operand registers.
// ROPE (mv_rope_paged_decode): q arrives as the raw rows of the qkv projection and is rotated in place —
// in the Q^T fragment layout a lane holds dims 32s + 8 quad + j, so the rotation partner d + 64 is fragment s + 2 of the
// SAME lane — and the workgroup (chunk 0) whose tiles hold the sequence's last token first writes that token's rotated K
// and raw V rows into the cache (wave 0, in rope_qk_store_kernel's 16-lanes-per-row layout).
template <typename Tag, bool KLDS, bool ROPE = false>
__global__ void __launch_bounds__(256, KLDS ? 2 : 4) paged_decode_mfma_kernel(DecodeParams p, int n_chunks) {
  typedef typename elem<Tag>::T T;
  __shared__ __attribute__((aligned(16))) char smem[KLDS ? SMEM_BYTES_KLDS : SMEM_BYTES];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, quad = lane >> 4;
  const int hkv = blockIdx.x / n_chunks;
  const int chunk = blockIdx.x - hkv * n_chunks;
  const int split = blockIdx.y, b = blockIdx.z;
  const int h0 = hkv * p.g_total + chunk * 16;
  const int g_live = min(16, p.g_total - chunk * 16);

  int ctx = p.context_lens[b];
  const int64_t cap = (int64_t)p.max_blocks * p.block_size;
  if (ctx > cap) ctx = (int)cap;
  if (ctx < 0) ctx = 0;
  const int units = (ctx + UNIT - 1) / UNIT;
  const int ups = (units + p.nsplit - 1) / p.nsplit;
  const int u_begin = split * ups;
  const int u_end = min(units, u_begin + ups);

  // Q^T fragments (B operand): lane (q = i16, quad) holds Q[q][32s + 8 quad .. +8)
  u32x4 qf[4];
  {
    const bool q_ok = i16 < g_live;
    const T *qp = (const T *)p.q + (int64_t)b * p.q_stride + (int64_t)(h0 + (q_ok ? i16 : 0)) * D + 8 * quad;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const u32x4 zero = {0u, 0u, 0u, 0u};
      qf[s] = q_ok ? *reinterpret_cast<const u32x4 *>(qp + 32 * s) : zero;
    }
  }
  if constexpr (ROPE) {
    const int64_t pos = p.positions[b];
    const bool pos_ok = pos >= 0 && pos < p.max_pos;
    if (p.q_norm_w != nullptr) {
      // per-head RMSNorm of q, rope_qk_store_kernel's sum order: there a head row sits on 16 lanes "sub" (8 dims each,
      // fmaf chain over them) and the lane sums meet in a butterfly over sub ^ 1, 2, 4, 8.  Here sub = 4 s + quad:
      // sub ^ 1 / ^ 2 are the lanes quad ^ 1 / ^ 2 (16 / 32 lanes away), sub ^ 4 / ^ 8 are fragments s ^ 1 / s ^ 2
      float x[4][8], ss[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        unpack8<Tag>(qf[s], x[s]);
        ss[s] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) ss[s] = fmaf(x[s][j], x[s][j], ss[s]);
        ss[s] += __shfl_xor(ss[s], 16, 64);
        ss[s] += __shfl_xor(ss[s], 32, 64);
      }
      const float a0 = ss[0] + ss[1], a1 = ss[1] + ss[0], a2 = ss[2] + ss[3], a3 = ss[3] + ss[2];  // sub ^ 4
      const float tot[4] = {a0 + a2, a1 + a3, a2 + a0, a3 + a1};                                       // sub ^ 8
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float rs = 1.0f / sqrtf(tot[s] / (float)D + p.eps);
        float w[8];
        unpack8<Tag>(*reinterpret_cast<const u32x4 *>((const T *)p.q_norm_w + 32 * s + 8 * quad), w);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[s][j] = elem<Tag>::ld(elem<Tag>::st(__fmul_rn(__fmul_rn(x[s][j], rs), w[j])));
        qf[s] = pack8<Tag>(x[s]);
      }
    }
    // q: dims d = 32 s + 8 quad + j (s = 0, 1) pair with d + 64 (fragment s + 2); products rounded separately, like
    // rope_qk_store_kernel: y[d] = x[d] cos - x[d+64] sin, y[d+64] = x[d+64] cos + x[d] sin
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float lo[8], hi[8], ylo[8], yhi[8];
      unpack8<Tag>(qf[s], lo);
      unpack8<Tag>(qf[s + 2], hi);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = 32 * s + 8 * quad + j;
        const float c = pos_ok ? p.cos[pos * (D / 2) + i] : 1.0f, sn = pos_ok ? p.sin[pos * (D / 2) + i] : 0.0f;
        ylo[j] = __fsub_rn(__fmul_rn(lo[j], c), __fmul_rn(hi[j], sn));
        yhi[j] = __fadd_rn(__fmul_rn(hi[j], c), __fmul_rn(lo[j], sn));
      }
      const bool live = i16 < g_live;   // (rows beyond the live heads stay zero)
      qf[s] = live ? pack8<Tag>(ylo) : u32x4{0u, 0u, 0u, 0u};
      qf[s + 2] = live ? pack8<Tag>(yhi) : u32x4{0u, 0u, 0u, 0u};
    }
    // the new token's K / V rows: written by the workgroup (chunk 0) that will read them, before anybody does
    int64_t slot = p.slot_mapping ? (int64_t)p.slot_mapping[b] : -1;
    if (slot >= p.num_blocks * p.block_size) slot = -1;
    const int u_last = ctx > 0 ? (ctx - 1) / UNIT : -1;
    if (slot >= 0 && chunk == 0 && u_last >= u_begin && u_last < u_end && wave == 0 && lane < 16) {
      const int sub = lane;
      float cs[8], sn[8];
      const int i0 = (sub % 8) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        cs[j] = pos_ok ? p.cos[pos * (D / 2) + i0 + j] : 1.0f;
        sn[j] = pos_ok ? p.sin[pos * (D / 2) + i0 + j] : 0.0f;
      }
      const u32x4 kraw = *reinterpret_cast<const u32x4 *>((const T *)p.k_new + (int64_t)b * p.k_stride + (int64_t)hkv * D + sub * 8);
      const u32x4 vraw = *reinterpret_cast<const u32x4 *>((const T *)p.v_new + (int64_t)b * p.v_stride + (int64_t)hkv * D + sub * 8);
      const u32x4 krot = norm_rope_row<Tag, D>(kraw, (const T *)p.k_norm_w, p.eps, sub, cs, sn);
      *reinterpret_cast<u32x4 *>((T *)p.k_cache + (slot * p.n_kv_heads + hkv) * D + sub * 8) = krot;
      *reinterpret_cast<u32x4 *>((T *)p.v_cache + (slot * p.n_kv_heads + hkv) * D + sub * 8) = vraw;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
  f32x4 oacc[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;

  const int32_t *table = p.block_tables + (int64_t)b * p.table_stride;
  const int64_t row_elems = (int64_t)p.n_kv_heads * D;
  char *vlds = smem + wave * V_TILE_BYTES;
  char *klds = smem + (NW + wave) * V_TILE_BYTES;  // KLDS only
  const int q4 = i16 >> 2, pp = i16 & 3;

  for (int u = u_begin; u < u_end; ++u) {
    const int tile_base = (u * NW + wave) * TILE;
    if (tile_base >= ctx) break;  // wave-uniform
    const int bi = p.bs_shift >= 0 ? (tile_base >> p.bs_shift) : (tile_base / p.block_size);
    const int blk = bi < p.max_blocks ? __builtin_amdgcn_readfirstlane(table[bi]) : -1;
    const bool blk_ok = blk >= 0 && blk < p.num_blocks;
    if (!blk_ok) continue;  // wave-uniform: a hole in the table contributes nothing
    const int64_t row0 = (int64_t)blk * p.block_size + (tile_base - bi * p.block_size);
    const T *kb = (const T *)p.k_cache + (row0 * p.n_kv_heads + hkv) * D;
    const T *vb = (const T *)p.v_cache + (row0 * p.n_kv_heads + hkv) * D;

    // ---- V: LDS-DMA (global_load_lds_dwordx4), no VGPR staging.  Instruction n fills LDS rows
    //      4n..4n+3 linearly (wave base + lane*16), so the swizzle goes on the SOURCE: lane (quad, i16)
    //      fetches chunk (i16 ^ swz) of the key that lives in LDS row 4n + quad.  Keys past the context
    //      re-read the tile's first (valid) row: their probability is exactly 0 and the data is finite.
    const int n_valid = min(TILE, ctx - tile_base);
    if constexpr (KLDS) {
      // K first (it is needed first; vmcnt retires in order): LDS row = key, chunk XORed with (key & 15)
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        const int key = 4 * n + quad;
        const int kk = key < n_valid ? key : 0;
        const int ch = i16 ^ (key & 15);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(kb + kk * row_elems + 8 * ch),
                                         (__attribute__((address_space(3))) void *)(klds + n * 1024), 16, 0, 2);
      }
    }
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const int lrow = 4 * n + quad;
      // inverse of v_lrow: LDS row -> key of the tile
      const int kk_raw = (lrow & 16) + 4 * (((lrow >> 3) & 1) | (((lrow >> 2) & 1) << 1)) + (lrow & 3);
      const int kk = kk_raw < n_valid ? kk_raw : 0;
      const int ch = i16 ^ (((lrow & 3) << 2) | ((lrow >> 2) & 3));
      const T *src = vb + kk * row_elems + 8 * ch;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(vlds + n * 1024), 16, 0, 2 /* nt: whole rows, read once */);
    }
    // ---- K: A operand, lane (key = i16, quad) reads 16 B at d = 32s + 8 quad of key 16kt + i16
    u32x4 kf[2][4];
    if constexpr (KLDS) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the 8 K DMAs have landed, the 8 V DMAs may still fly
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const int key = 16 * kt + i16;
#pragma unroll
        for (int s = 0; s < 4; ++s)
          kf[kt][s] = *reinterpret_cast<const u32x4 *>(klds + key * ROW_BYTES + (((4 * s + quad) ^ (key & 15)) << 4));
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const int kk = min(16 * kt + i16, n_valid - 1);  // past the context: re-read a valid key (masked below)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          // plain loads: the four 64-B pieces of a 256-B row arrive in four instructions, the cache must
          // keep each 128-B line until its second half is read (non-temporal K loads measured 8 % slower)
          kf[kt][s] = *reinterpret_cast<const u32x4 *>(kb + kk * row_elems + 32 * s + 8 * quad);
      }
    }
    // (K loads are issued AFTER the V DMA below: vmcnt retires in order, so "K has landed" implies
    //  "V is in LDS")
    // ---- S^T = K Q^T
    f32x4 sacc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      sacc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) sacc[kt] = mfma16<Tag>(kf[kt][s], qf[s], sacc[kt]);
    }
    // ---- online softmax over the 32 keys (log2 domain); keys of this lane: 16kt + 4 quad + reg
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = tile_base + 16 * kt + 4 * quad + r;
        float s = sacc[kt][r] * p.scale_log2;
        s = (blk_ok && key < ctx) ? s : -INFINITY;
        sacc[kt][r] = s;
        mx = fmaxf(mx, s);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float rs = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(sacc[kt][r] - m_new);
        sacc[kt][r] = e;
        rs += e;
      }
    l_run = l_run * alpha + rs;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) oacc[dt] *= alpha;
    // P^T as B operand: k-slot (quad, j): j < 4 -> key 4 quad + j of the first 16, else of the second 16
    u32x4 pf;
    pf[0] = pack2<Tag>(sacc[0][0], sacc[0][1]);
    pf[1] = pack2<Tag>(sacc[0][2], sacc[0][3]);
    pf[2] = pack2<Tag>(sacc[1][0], sacc[1][1]);
    pf[3] = pack2<Tag>(sacc[1][2], sacc[1][3]);

    // ---- O^T += V^T P^T : A operand rows d = 16dt + i16 via transposed LDS reads
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the V DMA of this tile (the compiler cannot see it)
    const int lrow_a = 8 * (quad & 1) + 4 * (quad >> 1) + q4;  // = v_lrow(4 quad + q4)
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      const int ch = 2 * dt + (pp >> 1);
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(vlds + v_off(lrow_a, ch) + 8 * (pp & 1)));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(vlds + v_off(lrow_a + 16, ch) + 8 * (pp & 1)));
      // the transposed read already delivers packed 16-bit pairs: reinterpret, do not repack
      const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
      const u32x4 a = {lo2.x, lo2.y, hi2.x, hi2.y};
      oacc[dt] = mfma16<Tag>(a, pf, oacc[dt]);
    }
  }

  // ---- merge the 4 waves (the merge area aliases the V tiles: barrier first)
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  __syncthreads();
  float *merge = reinterpret_cast<float *>(smem);
  if (i16 < g_live) {
    float *row = merge + (wave * 16 + i16) * MERGE_STRIDE;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) row[16 * dt + 4 * quad + r] = oacc[dt][r];
    if (quad == 0) {
      row[D] = m_run;
      row[D + 1] = l_tot;
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < g_live * D; idx += blockDim.x) {
    const int g = idx / D, d = idx - g * D;
    float M = merge[(0 * 16 + g) * MERGE_STRIDE + D];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, merge[(w * 16 + g) * MERGE_STRIDE + D]);
    float L = 0.f, A = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float *row = merge + (w * 16 + g) * MERGE_STRIDE;
      const float f = __builtin_amdgcn_exp2f(row[D] - M);
      L += row[D + 1] * f;
      A += row[d] * f;
    }
    const int h = h0 + g;
    if (p.nsplit == 1) {
      ((T *)p.out)[((int64_t)b * p.n_heads + h) * D + d] = elem<Tag>::st(L > 0.f ? A / L : 0.f);
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

}  // namespace

int mv_paged_decode_mfma_launch(const DecodeParams &p, int n_chunks, int nsplit, int batch, int dtype,
                                int force, hipStream_t stream) {
  dim3 grid(p.n_kv_heads * n_chunks, nsplit, batch), block(256);
  // K through LDS pays once the (sequence, kv head) pairs alone fill the chip (whole-row DMA instead of
  // 16 rows x 64 B per load: 5.5 -> 5.9 TB/s on Qwen2-7B heads at B=256, 5.8 -> 6.4 at ctx 4096); with fewer
  // workgroups its 2-per-CU residency (64 KiB of LDS) costs more than it gains.  The caller's flags can force.
  const bool klds = force >= 0 ? force == 1 : (int64_t)grid.x * grid.z >= 1024;
  const bool rope = p.positions != nullptr;   // mv_rope_paged_decode (the caller has checked: no head norms, one chunk)
#define MV_MFMA_LAUNCH(TAG)                                                                                        \
  do {                                                                                                             \
    if (rope && klds) hipLaunchKernelGGL((paged_decode_mfma_kernel<TAG, true, true>), grid, block, 0, stream, p, n_chunks);   \
    else if (rope) hipLaunchKernelGGL((paged_decode_mfma_kernel<TAG, false, true>), grid, block, 0, stream, p, n_chunks);     \
    else if (klds) hipLaunchKernelGGL((paged_decode_mfma_kernel<TAG, true>), grid, block, 0, stream, p, n_chunks);            \
    else hipLaunchKernelGGL((paged_decode_mfma_kernel<TAG, false>), grid, block, 0, stream, p, n_chunks);                     \
  } while (0)
  if (dtype == MV_BF16) MV_MFMA_LAUNCH(bf16_tag);
  else if (dtype == MV_F16) MV_MFMA_LAUNCH(f16_tag);
  else return MV_E_DTYPE;
#undef MV_MFMA_LAUNCH
  MV_LAUNCH_CHECK();
  return MV_OK;
}
