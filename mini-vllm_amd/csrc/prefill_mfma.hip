// K2 (MFMA path) — packed varlen causal flash attention for gfx950, bf16/f16, D = 128
// (reference call site: flash_attn_varlen_func, minivllm/models/layers/attention.py:422-433;
//  semantics of the fallback :467-504,554-597 with bottom-right aligned causal mask).
//
// One workgroup = 4 waves = 128 query rows of one (sequence, query head); each wave owns 32
// rows.  K/V are streamed in 64-key tiles through LDS (next tile's global loads are issued
// before the current tile's MFMAs and written to LDS after the barrier).
// Per wave and tile, all on v_mfma_f32_32x32x16:
//   S^T[key][q] = K Q^T      A = K rows from LDS (ds_read_b128, XOR-swizzled), B = Q^T in registers
//   online softmax           a query row lives on lane r and lane r+32 (16 keys each per 32-key tile):
//                            row max / sum need ONE cross-lane exchange, not a tree
//   O^T[d][q] += V^T P^T     the S^T accumulator is reused in place as the B operand (its key index
//                            is the register index), A = V^T via ds_read_b64_tr_b16 transposed reads
// so P never goes through LDS and the rescale factor of a query row is a per-lane scalar.
#include "prefill.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int BM = 128;  // query rows per workgroup
constexpr int BN = 64;   // keys per tile
constexpr int D = 128;
constexpr int NS = D / 16;      // k-steps of the QK^T product
constexpr int NDT = D / 32;     // 32-wide d tiles of the output
constexpr int ROW_BYTES = D * 2;
constexpr int TILE_BYTES = BN * ROW_BYTES;  // 16 KiB

template <typename Tag>
__device__ __forceinline__ f32x16 mfma(const u32x4 &a, const u32x4 &b, const f32x16 &c) {
  if constexpr (__is_same(Tag, bf16_tag))
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <typename Tag>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  return (uint32_t)elem<Tag>::st(lo) | ((uint32_t)elem<Tag>::st(hi) << 16);
}

// K tile: row-major [key][D], 16-byte chunk index XORed with (key & 15): the A-operand read
// (32 different keys, same chunk) is bank-conflict free.
__device__ __forceinline__ int k_off(int row, int ch) { return row * ROW_BYTES + ((ch ^ (row & 15)) << 4); }
// V tile: row-major [key][D] with the "one image for row and transposed reads" swizzle
// (cdna_hip_programming.md T10 (b)): 4 consecutive keys x 16 columns blocks read conflict free.
__device__ __forceinline__ int v_off(int row, int ch) {
  return row * ROW_BYTES + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

template <typename Tag>
__global__ void __launch_bounds__(256, 2) prefill_mfma_kernel(PrefillParams p) {
  typedef typename elem<Tag>::T T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *ldsK = smem;
  char *ldsV = smem + TILE_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int g = (lane >> 4) & 1, i16 = lane & 15;

  const int seq = blockIdx.z, head = blockIdx.y;
  // heaviest (latest) query blocks first: the causal tail finishes earlier
  const int qb = (int)gridDim.x - 1 - (int)blockIdx.x;
  const int q0 = p.cu_q[seq], sq = p.cu_q[seq + 1] - q0;
  const int k0 = p.cu_k[seq], sk = p.cu_k[seq + 1] - k0;
  if (qb * BM >= sq) return;
  const int hkv = head / (p.n_heads / p.n_kv_heads);
  const int shift = sk - sq;  // bottom-right alignment of the causal mask

  // ---- Q^T fragments (B operand): lane (r,h) holds Q[row r][16s + 8h .. +8)
  const int qidx = qb * BM + wave * 32 + r;
  const bool q_ok = qidx < sq;
  u32x4 qf[NS];
  {
    const T *qp = (const T *)p.q + (int64_t)(q0 + (q_ok ? qidx : 0)) * p.q_stride + (int64_t)head * D + 8 * h;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const u32x4 zero = {0u, 0u, 0u, 0u};
      qf[s] = q_ok ? *reinterpret_cast<const u32x4 *>(qp + 16 * s) : zero;
    }
  }

  f32x16 oacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[dt][i] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float scale_log2 = p.scale * 1.4426950408889634f;

  // keys needed by this query block: everything up to its last row's diagonal
  const int q_last = min(sq, (qb + 1) * BM) - 1;
  const int kv_len = min(sk, q_last + shift + 1);
  const int n_tiles = (kv_len + BN - 1) / BN;
  const int wave_qpos_min = shift + qb * BM + wave * 32;  // smallest key limit among this wave's rows

  // ---- tile staging: each thread moves 4 x 16 B of K and of V per tile
  u32x4 kreg[4], vreg[4];
  auto fetch = [&](int t) {
    const int kv0 = t * BN;
    int64_t base_row;  // row index into the K/V source of key kv0
    bool tile_ok = true;
    if (p.block_tables) {
      const int bi = kv0 / p.block_size;
      const int blk = p.block_tables[(int64_t)seq * p.table_stride + bi];
      tile_ok = blk >= 0 && blk < p.num_blocks;
      base_row = (int64_t)blk * p.block_size + (kv0 - bi * p.block_size);
    } else {
      base_row = k0 + kv0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pc = tid + 256 * i;
      const int row = pc >> 4, ch = pc & 15;
      const bool ok = tile_ok && (kv0 + row) < sk;
      const u32x4 zero = {0u, 0u, 0u, 0u};
      if (p.block_tables) {
        const int64_t e = ((base_row + row) * p.n_kv_heads + hkv) * D + ch * 8;
        kreg[i] = ok ? *reinterpret_cast<const u32x4 *>((const T *)p.k_cache + e) : zero;
        vreg[i] = ok ? *reinterpret_cast<const u32x4 *>((const T *)p.v_cache + e) : zero;
      } else {
        kreg[i] = ok ? *reinterpret_cast<const u32x4 *>((const T *)p.k + (base_row + row) * p.k_stride + hkv * D + ch * 8) : zero;
        vreg[i] = ok ? *reinterpret_cast<const u32x4 *>((const T *)p.v + (base_row + row) * p.v_stride + hkv * D + ch * 8) : zero;
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pc = tid + 256 * i;
      const int row = pc >> 4, ch = pc & 15;
      *reinterpret_cast<u32x4 *>(ldsK + k_off(row, ch)) = kreg[i];
      *reinterpret_cast<u32x4 *>(ldsV + v_off(row, ch)) = vreg[i];
    }
  };

  if (n_tiles > 0) fetch(0);
  for (int t = 0; t < n_tiles; ++t) {
    __syncthreads();  // every wave is done reading the previous tile
    stage();
    __syncthreads();
    if (t + 1 < n_tiles) fetch(t + 1);  // in flight during this tile's MFMAs
    const int kv0 = t * BN;

    // ---- S^T = K Q^T : two 32-key tiles
    f32x16 sacc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sacc[kt][i] = 0.f;
      const int krow = kt * 32 + r;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const u32x4 a = *reinterpret_cast<const u32x4 *>(ldsK + k_off(krow, 2 * s + h));
        sacc[kt] = mfma<Tag>(a, qf[s], sacc[kt]);
      }
    }

    // ---- scale, mask, online softmax (log2 domain)
    const bool need_mask = (kv0 + BN - 1 > wave_qpos_min) || (kv0 + BN > sk);
    const int klim = min(qidx + shift, sk - 1);  // last key this query row may see
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float s = sacc[kt][i] * scale_log2;
        if (need_mask) {
          const int key = kv0 + kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          s = key <= klim ? s : -INFINITY;
        }
        sacc[kt][i] = s;
        mx = fmaxf(mx, s);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float rs = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float e = __builtin_amdgcn_exp2f(sacc[kt][i] - m_new);
        sacc[kt][i] = e;
        rs += e;
      }
    l_run = l_run * alpha + rs;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) oacc[dt][i] *= alpha;

    // ---- P^T as MFMA B operand: registers 8*s2 .. 8*s2+7 of a 32-key tile are k-step s2
    u32x4 pf[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          pf[kt][s2][j] = pack2<Tag>(sacc[kt][8 * s2 + 2 * j], sacc[kt][8 * s2 + 2 * j + 1]);

    // ---- O^T += V^T P^T
    const int q4 = i16 >> 2, pp = i16 & 3;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int ch = 4 * dt + 2 * g + (pp >> 1);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int rb = kt * 32 + 16 * s2 + 4 * h + q4;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (lds_s16x4 *)(ldsV + v_off(rb, ch) + 8 * (pp & 1)));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (lds_s16x4 *)(ldsV + v_off(rb + 8, ch) + 8 * (pp & 1)));
          u32x4 a;
          a[0] = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
          a[1] = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
          a[2] = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
          a[3] = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
          oacc[dt] = mfma<Tag>(a, pf[kt][s2], oacc[dt]);
        }
    }
  }

  // ---- epilogue: O[q][d] = O^T[d][q] / l ; lane (r,h) owns d = 32dt + 8c + 4h + {0..3}
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (q_ok) {
    T *op = (T *)p.out + ((int64_t)(q0 + qidx) * p.n_heads + head) * D;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        u32x2 w;
        w[0] = pack2<Tag>(oacc[dt][4 * c + 0] * inv, oacc[dt][4 * c + 1] * inv);
        w[1] = pack2<Tag>(oacc[dt][4 * c + 2] * inv, oacc[dt][4 * c + 3] * inv);
        *reinterpret_cast<u32x2 *>(op + 32 * dt + 8 * c + 4 * h) = w;
      }
  }
}

}  // namespace

int mv_prefill_mfma_launch(const PrefillParams &p, int dtype, hipStream_t stream) {
  if (p.head_dim != D || (dtype != MV_BF16 && dtype != MV_F16)) return MV_E_SHAPE;
  if (p.block_tables && (p.block_size % BN) != 0) return MV_E_SHAPE;  // a 64-key tile must sit in one block
  const int q_blocks = (p.max_seqlen_q + BM - 1) / BM;
  if (q_blocks == 0 || p.n_seqs == 0) return MV_OK;
  dim3 grid(q_blocks, p.n_heads, p.n_seqs), block(256);
  const size_t lds = 2 * TILE_BYTES;
  if (dtype == MV_BF16)
    hipLaunchKernelGGL((prefill_mfma_kernel<bf16_tag>), grid, block, lds, stream, p);
  else
    hipLaunchKernelGGL((prefill_mfma_kernel<f16_tag>), grid, block, lds, stream, p);
  MV_LAUNCH_CHECK();
  return MV_OK;
}
