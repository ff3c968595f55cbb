// K2 (MFMA path) — packed varlen causal flash attention for gfx950, bf16/f16, D = 128
// (reference call site: flash_attn_varlen_func, minivllm/models/layers/attention.py:422-433;
//  semantics of the fallback :467-504,554-597 with bottom-right aligned causal mask).
//
// One workgroup = 4 waves = 128 query rows of one (sequence, query head); each wave owns 32
// rows.  K/V are streamed in 64-key tiles through LDS, double-buffered, by LDS-DMA (the next tile's
// transfers are issued before the current tile's MFMAs and waited for right before the tile's barrier).
// Per wave and tile, all on v_mfma_f32_32x32x16:
//   S^T[key][q] = K Q^T      A = K rows from LDS (ds_read_b128, XOR-swizzled), B = Q^T in registers
//   online softmax           a query row lives on lane r and lane r+32 (16 keys each per 32-key tile):
//                            row max / sum need ONE cross-lane exchange, not a tree
//   O^T[d][q] += V^T P^T     the S^T accumulator is reused in place as the B operand (its key index
//                            is the register index), A = V^T via ds_read_b64_tr_b16 transposed reads
// so P never goes through LDS and the rescale factor of a query row is a per-lane scalar.
#include <type_traits>

#include "prefill.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

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
  // one v_cvt_pk_{bf16,f16}_f32 (round to nearest even), not two converts + a merge
  if constexpr (__is_same(Tag, bf16_tag))
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2_t));
  else
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, f16x2_t));
}

// K tile: row-major [key][D], 16-byte chunk index XORed with (key & 15): the A-operand read
// (32 different keys, same chunk) is bank-conflict free.
__device__ __forceinline__ int k_off(int row, int ch) { return row * ROW_BYTES + ((ch ^ (row & 15)) << 4); }
// V tile: row-major [key][D] with the "one image for row and transposed reads" swizzle
// (cdna_hip_programming.md T10 (b)): 4 consecutive keys x 16 columns blocks read conflict free.
__device__ __forceinline__ int v_off(int row, int ch) {
  return row * ROW_BYTES + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

// ---- round 3: hand-placed LDS reads --------------------------------------------------------------------------
// What hipcc makes of the plain form of this loop (ROCm 7.2, -save-temps): every pair of QK^T MFMAs sits behind its own
// `ds_read_b128 x2; s_waitcnt lgkmcnt(0)` (eight fully exposed LDS round trips per tile), and the first transposed V
// read is preceded by a compiler-inserted `s_waitcnt vmcnt(0)`: the backend cannot tell that the LDS-DMA of tile t+1
// (in flight since the top of the iteration) writes the OTHER buffer, so the whole DMA latency lands in front of PV.
// With PIPE the K / V fragment reads are inline assembly (invisible to that analysis), issued two k-steps (K) and one
// 16-key group (V; the first group before the softmax) ahead of the MFMAs that consume them, and retired by counted
// `s_waitcnt lgkmcnt(N)` statements that name the fragment registers as read-write operands, so no MFMA can be
// scheduled above the wait that covers its operands (cdna_hip_programming.md §5.7 form (ii)).  LDS operations complete
// in order, so a count of "my newer reads" stays safe when the compiler issues LDS operations of its own in between;
// scalar loads (which share the counter and return out of order) must not be in flight: the block-table entry of the
// paged form is therefore fetched with a vector load.
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}
template <int OFF>
__device__ __forceinline__ void ds_r128(u32x4 &d, uint32_t a) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "i"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void ds_r64tr(u32x2 &d, uint32_t a) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(a), "i"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_wait2(u32x4 &a, u32x4 &b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lgkm_wait1(u32x2 &a, u32x2 &b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lgkm_wait8(u32x2 (&lo)[4], u32x2 (&hi)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3])
               : "n"(N));
}

// Round 2 (profiles/r02_prefill_kernel.md): the softmax arithmetic is scalar f32 (v_fma_f32 / v_exp_f32 / v_add_f32;
// the packed v_pk_* forms cost more issue cycles beside MFMAs than the two ops they replace, and this file is built
// with -fno-slp-vectorize so that the compiler does not pair them either), and a full tile's DMA source addresses
// are one scalar base + per-lane constants instead of ~90 integer VALU instructions per tile and wave:
// 743 -> 850 TFLOP/s on 4 x 4096 tokens, 28/4 heads.
// NW = waves per workgroup (32 query rows each): 4 (two workgroups per CU) or 8 (one; the K/V tile and its DMA
// instructions — ~100 issue cycles apiece — are shared by twice the rows)
template <typename Tag, int NW, bool PIPE>
__global__ void __launch_bounds__(NW * 64, 2) prefill_mfma_kernel(PrefillParams p, float rescale_thr) {
  constexpr int NBUF = 2;   // K|V tile buffers in LDS
  constexpr int BM = NW * 32;   // query rows per workgroup
  constexpr int PP = 16 / NW;   // 1-KiB DMA pieces (4 rows) of a K tile per wave, and as many of the V tile
  typedef typename elem<Tag>::T T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // two K|V tile buffers: tile t+1 is written while tile t is being read -> ONE barrier per tile

  const int tid = threadIdx.x;
  // the wave index as a SCALAR: everything derived from it (the diagonal test below) then branches on
  // scc instead of being if-converted into 100 predicated VALU instructions on every tile
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int g = (lane >> 4) & 1, i16 = lane & 15;

  // Dispatch order is x fastest, z slowest: the query block is the SLOWEST index and runs from the
  // heaviest (latest rows, most KV tiles) to the lightest, across all heads and sequences — otherwise
  // the last (head, sequence) pair starts its 64-tile blocks when everything else has drained
  // (measured: 0.92 resident waves per SIMD on average instead of 2).
  const int head = blockIdx.x, seq = blockIdx.y;
  const int qb = (int)gridDim.z - 1 - (int)blockIdx.z;
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
  // ... and the largest: tiles that start beyond it are masked out for EVERY row of this wave (the workgroup walks the
  // tiles its LAST wave needs): such a wave skips the tile's arithmetic (p = 0 for all its scores: maximum, sum and
  // accumulator would not move) and only takes part in the DMA and the barrier.  Waves 0..NW-2 save (NW-1-w)/2 tiles
  // each: 5 % of the MFMAs at 4096 tokens, 22 % at 1024, more for short prompts.
  const int wave_qpos_max = wave_qpos_min + 31;

  const bool paged = p.block_tables != nullptr;
  const T *ksrc = paged ? (const T *)p.k_cache : (const T *)p.k;
  const T *vsrc = paged ? (const T *)p.v_cache : (const T *)p.v;
  const int64_t kstride = paged ? (int64_t)p.n_kv_heads * D : p.k_stride;
  const int64_t vstride = paged ? (int64_t)p.n_kv_heads * D : p.v_stride;
  // ---- tile staging: K/V tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4), no VGPR staging and no
  //      ds_write pass (+3..15 % over register staging).  The DMA of tile t+1 is issued at the top of iteration
  //      t into the buffer whose readers all passed the previous barrier.  A DMA instruction fills 1 KiB of LDS
  //      linearly per wave (4 rows x 256 B), so both swizzles go on the SOURCE address: lane -> (row, stored
  //      chunk) fetches the logical chunk that belongs there.  Rows past the sequence end re-read the last valid
  //      key (finite data whose probability is exactly 0 after masking); loads are unconditional.
  // per-lane parts of the DMA source addresses that do not change from tile to tile (non-paged: everything but
  // the tile's first row; the ragged last tile clamps rows, so it takes the general path below)
  int64_t krel[PP], vrel[PP];
#pragma unroll
  for (int i = 0; i < PP; ++i) {
    const int row = 4 * (NW * i + wave) + (lane >> 4), pc = lane & 15;
    krel[i] = (int64_t)row * kstride + (pc ^ (row & 15)) * 8;
    vrel[i] = (int64_t)row * vstride + (pc ^ (((row & 3) << 2) | ((row >> 2) & 3))) * 8;
  }
  bool tile_ok_next = true;
  // (PIPE: through the vector memory path — mbcnt is 0 in every lane but opaque to the compiler — so that no scalar
  // load is outstanding beside the hand-counted LDS waits)
  auto table_entry = [&](int bi) -> int {
    const int32_t *e = p.block_tables + (int64_t)seq * p.table_stride + bi;
    if constexpr (PIPE) return __builtin_amdgcn_readfirstlane(e[__builtin_amdgcn_mbcnt_lo(0u, 0u)]);
    else return *e;
  };
  auto dma = [&](int t, int buf) {
    {
      const int kv0f = t * BN;
      if (kv0f + BN <= sk) {   // a full tile: one scalar base + the per-lane constants
        int64_t base_row = k0 + kv0f;
        tile_ok_next = true;
        if (paged) {
          const int bi = kv0f / p.block_size;
          const int blk = table_entry(bi);
          tile_ok_next = blk >= 0 && blk < p.num_blocks;
          base_row = (int64_t)(tile_ok_next ? blk : 0) * p.block_size + (kv0f - bi * p.block_size);
        }
        const T *kt = ksrc + base_row * kstride + hkv * D;
        const T *vt = vsrc + base_row * vstride + hkv * D;
        char *dk = smem + buf * 2 * TILE_BYTES, *dv = dk + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < PP; ++i) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(kt + krel[i]),
                                           (__attribute__((address_space(3))) void *)(dk + (NW * i + wave) * 1024), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vt + vrel[i]),
                                           (__attribute__((address_space(3))) void *)(dv + (NW * i + wave) * 1024), 16, 0, 0);
        }
        return;
      }
    }
    const int kv0 = t * BN;
    int64_t base_row = k0 + kv0;
    tile_ok_next = true;
    if (paged) {
      const int bi = kv0 / p.block_size;
      const int blk = table_entry(bi);
      tile_ok_next = blk >= 0 && blk < p.num_blocks;  // a hole: read block 0, mask the whole tile
      base_row = (int64_t)(tile_ok_next ? blk : 0) * p.block_size + (kv0 - bi * p.block_size);
    }
    const T *kt = ksrc + base_row * kstride + hkv * D;
    const T *vt = vsrc + base_row * vstride + hkv * D;
    const int last = sk - 1 - kv0;
    char *dk = smem + buf * 2 * TILE_BYTES, *dv = dk + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < PP; ++i) {
      // wave `wave`, instruction i fill LDS rows 4 (NW i + wave) .. +3 linearly: lane -> (row, stored chunk)
      const int wv = wave;
      const int row = 4 * (NW * i + wv) + (lane >> 4), pc = lane & 15;
      const int srow = min(row, last);
      const int kch = pc ^ (row & 15);
      const int vch = pc ^ (((row & 3) << 2) | ((row >> 2) & 3));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(kt + (int64_t)srow * kstride + kch * 8),
                                       (__attribute__((address_space(3))) void *)(dk + (NW * i + wv) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vt + (int64_t)srow * vstride + vch * 8),
                                       (__attribute__((address_space(3))) void *)(dv + (NW * i + wv) * 1024), 16, 0, 0);
    }
  };

  // ---- loop-invariant LDS read offsets (the tile-relative part is an instruction immediate)
  // K, A operand of k-step s: row r (+32 kt), chunk (2s+h) ^ (r & 15)
  int koff[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) koff[s] = r * ROW_BYTES + (((2 * s + h) ^ (r & 15)) << 4);
  // V^T, A operand of d-tile dt: rows 4h+q4 (+8) (+32 kt + 16 s2), chunk (4dt + 2g + pp/2) ^ swizzle(row)
  const int q4 = i16 >> 2, pp = i16 & 3;
  int voff_lo[NDT], voff_hi[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) {
    const int ch = 4 * dt + 2 * g + (pp >> 1);
    voff_lo[dt] = v_off(4 * h + q4, ch) + 8 * (pp & 1);
    voff_hi[dt] = v_off(4 * h + q4 + 8, ch) + 8 * (pp & 1);
  }

  bool tile_ok = true;
  if (n_tiles > 0) {
    dma(0, 0);
    tile_ok = tile_ok_next;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler does not count LDS-DMA completions)
  }
  __syncthreads();
  // (PIPE: the per-lane LDS read addresses once, relative to buffer 0; the loop below is unrolled over the two buffers
  // so that the buffer, the 32-key half and the 16-key group are instruction immediates: no address arithmetic per tile)
  uint32_t ka0[NS], va_lo[NDT], va_hi[NDT];
  if constexpr (PIPE) {
    const uint32_t base = lds_addr(smem);
#pragma unroll
    for (int s = 0; s < NS; ++s) ka0[s] = base + koff[s];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      va_lo[dt] = base + TILE_BYTES + voff_lo[dt];
      va_hi[dt] = base + TILE_BYTES + voff_hi[dt];
    }
  }
  // ---- the three phases of a tile.  BO = byte offset of the tile's K|V buffer (compile time: the loop is unrolled
  //      over the two buffers), add = a run-time addend of the read addresses (0 in the shipped loop)
  f32x16 sacc[2];
  u32x2 vlo[1][NDT], vhi[1][NDT];     // PIPE: the V^T fragments of one 16-key group (re-filled behind each MFMA)
  u32x4 pf[2][2];                     // P^T fragments of the tile
  auto qk_phase = [&](auto BOc, const uint32_t add) __attribute__((always_inline)) {
    constexpr int BO = decltype(BOc)::value;
    // ---- S^T = K Q^T : two 32-key tiles (raw, unscaled scores; first k-step starts from an inline 0)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (PIPE) {
      u32x4 kf[2][2];                   // two sets of K fragments: reads run one k-step (two MFMAs) ahead
      uint32_t ka[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) ka[s] = ka0[s] + add;
      ds_r128<BO>(kf[0][0], ka[0]);
      ds_r128<BO + 32 * ROW_BYTES>(kf[0][1], ka[0]);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) {
          ds_r128<BO>(kf[(s + 1) & 1][0], ka[s + 1]);
          ds_r128<BO + 32 * ROW_BYTES>(kf[(s + 1) & 1][1], ka[s + 1]);
          lgkm_wait2<2>(kf[s & 1][0], kf[s & 1][1]);
        } else {
          // the first group of V^T fragments goes out before the last wait: it lands under the softmax
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) {
            ds_r64tr<BO>(vlo[0][dt], va_lo[dt] + add);
            ds_r64tr<BO>(vhi[0][dt], va_hi[dt] + add);
          }
          lgkm_wait2<2 * NDT>(kf[s & 1][0], kf[s & 1][1]);
        }
        sacc[0] = mfma<Tag>(kf[s & 1][0], qf[s], s == 0 ? zero16 : sacc[0]);
        sacc[1] = mfma<Tag>(kf[s & 1][1], qf[s], s == 0 ? zero16 : sacc[1]);
      }
    } else {
      // the two 32-key accumulators alternate: a dependent MFMA issued right behind its predecessor waits
      // for the whole 16-pass latency, an independent one issues as soon as the pipe is free
      const char *k0_ = smem + BO, *k1 = k0_ + 32 * ROW_BYTES;  // (row + 32) & 15 == row & 15: same swizzle
      sacc[0] = mfma<Tag>(*reinterpret_cast<const u32x4 *>(k0_ + koff[0]), qf[0], zero16);
      sacc[1] = mfma<Tag>(*reinterpret_cast<const u32x4 *>(k1 + koff[0]), qf[0], zero16);
#pragma unroll
      for (int s = 1; s < NS; ++s) {
        sacc[0] = mfma<Tag>(*reinterpret_cast<const u32x4 *>(k0_ + koff[s]), qf[s], sacc[0]);
        sacc[1] = mfma<Tag>(*reinterpret_cast<const u32x4 *>(k1 + koff[s]), qf[s], sacc[1]);
      }
    }
  };
  auto softmax_phase = [&](const int t, const bool tile_ok) __attribute__((always_inline)) {
    const int kv0 = t * BN;
    // ---- mask (only tiles that touch the diagonal / the ragged end: wave-uniform branch)
    if ((kv0 + BN - 1 > wave_qpos_min) || (kv0 + BN > sk) || !tile_ok) {
      const int klim = tile_ok ? min(qidx + shift, sk - 1) : -1;  // last key this query row may see
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = kv0 + kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          sacc[kt][i] = key <= klim ? sacc[kt][i] : -INFINITY;
        }
    }
    // ---- online softmax in the log2 domain: p = exp2(s*c - m), one FMA + one exp per score
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[kt][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * scale_log2);  // scale > 0: max commutes with the scaling
    // rescale_thr = 0: rescale whenever some row's maximum moved.  > 0 (log2 units): only when one moved by more than
    // that — the stale maximum then leaves probabilities of up to 2^thr in P (fp32 sums, 16-bit P with the same
    // relative rounding), and the 64 multiplies of the O rescale leave most tiles (cdna_hip_programming.md T13)
    if (!__all(m_new - m_run <= rescale_thr)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[dt][i] *= alpha;
    }
    {
      // one v_fma_f32 + one v_exp_f32 + one v_add_f32 per score (four independent partial row sums); this file is
      // built with -fno-slp-vectorize so that the compiler does not pair them into v_pk_* either
      const float nm = -m_run;
      float rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][i], scale_log2, nm));
          sacc[kt][i] = e;
          rs[i & 3] += e;
        }
      l_run += (rs[0] + rs[1]) + (rs[2] + rs[3]);
    }
    // ---- P^T as MFMA B operand: registers 8*s2 .. 8*s2+7 of a 32-key tile are k-step s2
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          pf[kt][s2][j] = pack2<Tag>(sacc[kt][8 * s2 + 2 * j], sacc[kt][8 * s2 + 2 * j + 1]);
  };
  // ---- O^T += V^T P^T   (rows 32kt + 16s2 + ... keep (row & 3) and ((row >> 2) & 3) of the base row:
  //      the swizzle term is loop invariant, only the row base moves by a multiple of 16 rows)
  auto pv_phase = [&](auto BOc, const uint32_t add) __attribute__((always_inline)) {
    constexpr int BO = decltype(BOc)::value;
    if constexpr (PIPE) {
      // group g = (kt, s2) = 16 keys.  ONE set of fragment registers, reused on the fly: the two reads of (g + 1, dt) go
      // out right behind the MFMA that consumed (g, dt), so eight reads stay in flight and the oldest pair is what the
      // next MFMA needs (group 0 was requested behind QK^T)
      const uint32_t vl[NDT] = {va_lo[0] + add, va_lo[1] + add, va_lo[2] + add, va_lo[3] + add};
      const uint32_t vh[NDT] = {va_hi[0] + add, va_hi[1] + add, va_hi[2] + add, va_hi[3] + add};
      auto pv = [&](auto G) __attribute__((always_inline)) {
        constexpr int g = decltype(G)::value;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          if constexpr (g + 1 < 4) lgkm_wait1<2 * NDT - 2>(vlo[0][dt], vhi[0][dt]);
          else if (dt == 0) lgkm_wait1<6>(vlo[0][dt], vhi[0][dt]);
          else if (dt == 1) lgkm_wait1<4>(vlo[0][dt], vhi[0][dt]);
          else if (dt == 2) lgkm_wait1<2>(vlo[0][dt], vhi[0][dt]);
          else lgkm_wait1<0>(vlo[0][dt], vhi[0][dt]);
          const u32x4 a = {vlo[0][dt].x, vlo[0][dt].y, vhi[0][dt].x, vhi[0][dt].y};
          oacc[dt] = mfma<Tag>(a, pf[g >> 1][g & 1], oacc[dt]);
          if constexpr (g + 1 < 4) {
            ds_r64tr<BO + (g + 1) * 16 * ROW_BYTES>(vlo[0][dt], vl[dt]);
            ds_r64tr<BO + (g + 1) * 16 * ROW_BYTES>(vhi[0][dt], vh[dt]);
          }
        }
      };
      pv(std::integral_constant<int, 0>{});
      pv(std::integral_constant<int, 1>{});
      pv(std::integral_constant<int, 2>{});
      pv(std::integral_constant<int, 3>{});
    } else {
      const char *vt = smem + BO + TILE_BYTES;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const char *vb_ = vt + (kt * 32 + 16 * s2) * ROW_BYTES;
          u32x4 a[NDT];
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(vb_ + voff_lo[dt]));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(vb_ + voff_hi[dt]));
            // the transposed read already delivers packed 16-bit pairs: reinterpret, do not repack
            const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
            a[dt] = u32x4{lo2.x, lo2.y, hi2.x, hi2.y};
          }
          // four independent accumulators in a row (see QK^T above)
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) oacc[dt] = mfma<Tag>(a[dt], pf[kt][s2], oacc[dt]);
        }
    }
  };
  // top of an iteration: the DMA of the next tile into the buffer whose readers all passed the previous barrier
  auto tile_top = [&](const int t) __attribute__((always_inline)) {
    // (PIPE: nothing of the prologue — kernel-argument scalar loads included — may still be counted when the first
    // hand-counted wait runs)
    if constexpr (PIPE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // global -> another LDS buffer, in flight during this tile's MFMAs  (tried in round 3: issued behind QK^T, in the
    // VALU-only stretch where an LDS-DMA instruction is cheaper to issue: 862 vs 876 TFLOP/s at 4 x 4096, dropped)
    if (t + 1 < n_tiles && !(rescale_thr < 0.f && t > 0)) dma(t + 1, (t + 1) % NBUF);   // (thr < 0: bench ablation)
  };
  // end of an iteration: this wave's share of tile t+1 has landed, then the single barrier of the tile: it publishes
  // tile t+1 and retires every wave's reads of the buffer the next DMA will overwrite
  auto tile_end = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tile_ok = tile_ok_next;
    __syncthreads();
  };
  // Tried in round 3 and removed (profiles/r03_prefill_kernel.md): the two waves of a SIMD run half a tile apart (waves
  // 4-7 defer the PV of a tile to the start of the next iteration, three K|V buffers) so that one wave's softmax runs
  // beside the other's MFMAs — correct, bit-identical, and SLOWER: 875-895 against 935-942 TFLOP/s at 4 x 4096.
  // (also tried and removed: two tiles per barrier — four K|V buffers, half the barriers: 958-960 against 970-988)
  auto tile = [&](const int t, auto BUF) __attribute__((always_inline)) {
    constexpr int BO = decltype(BUF)::value * 2 * TILE_BYTES;   // byte offset of this tile's K|V buffer
    tile_top(t);
    if (t * BN <= wave_qpos_max) {
      qk_phase(std::integral_constant<int, BO>{}, 0u);
      softmax_phase(t, tile_ok);
      pv_phase(std::integral_constant<int, BO>{}, 0u);
    }
    tile_end();
  };
  for (int t = 0; t < n_tiles; t += 2) {
    tile(t, std::integral_constant<int, 0>{});
    if (t + 1 < n_tiles) tile(t + 1, std::integral_constant<int, 1>{});
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

int mv_prefill_mfma_launch(const PrefillParams &p, int dtype, int variant, hipStream_t stream) {
  if (p.head_dim != D || (dtype != MV_BF16 && dtype != MV_F16)) return MV_E_SHAPE;
  if (p.block_tables && (p.block_size % BN) != 0) return MV_E_SHAPE;  // a 64-key tile must sit in one block
  // 8 waves (256 query rows) per workgroup share a K/V tile where the sequences are long enough that the larger
  // blocks still fill the chip evenly; short prompts keep 4 waves x 2 workgroups per CU (measured per shape,
  // profiles/r02_prefill_kernel.md).  variant (bench switch, flags >> 4): 1 = force 4 waves, 2 = force 8
  // round 3: variants 3 / 4 = the round-2 loop (compiler-placed LDS reads) with 4 / 8 waves, for comparison
  const bool no_defer = (variant & 8) != 0;   // bench switch: rescale on every tile whose maximum moved
  const bool no_dma = (variant & 16) != 0;    // bench ablation: K/V tiles are not refreshed (results are garbage)
  const float thr_pick = 6.0f + 2.0f * (float)((variant >> 6) & 3);   // bench switch: rescale threshold 6 / 8 / 10 / 12
  variant &= 7;
  const bool pipe = variant != 3 && variant != 4;
  // rescale threshold (log2 units): the O accumulator is rescaled only when a row maximum grows by more than a factor of
  // 2^6 (+5.5 % at 4 x 4096; P then holds values up to 64, well inside bf16 / f16 range; same relative rounding)
  const float thr = no_dma ? -1.0f : ((pipe && !no_defer) ? thr_pick : 0.0f);
  const int nw = (variant == 1 || variant == 3) ? 4 : ((variant == 2 || variant == 4) ? 8 : (p.max_seqlen_q >= 2048 ? 8 : 4));
  const int bm = nw * 32;
  const int q_blocks = (p.max_seqlen_q + bm - 1) / bm;
  if (q_blocks == 0 || p.n_seqs == 0) return MV_OK;
  dim3 grid(p.n_heads, p.n_seqs, q_blocks), block(nw * 64);
  const size_t lds = 4 * TILE_BYTES;   // 2 x (K tile + V tile) = 64 KiB -> two workgroups per CU
#define MV_PF_LAUNCH(NWAVES, PIPED)                                                                            \
  do {                                                                                                        \
    if (dtype == MV_BF16) hipLaunchKernelGGL((prefill_mfma_kernel<bf16_tag, NWAVES, PIPED>), grid, block, lds, stream, p, thr); \
    else hipLaunchKernelGGL((prefill_mfma_kernel<f16_tag, NWAVES, PIPED>), grid, block, lds, stream, p, thr);   \
  } while (0)
  if (nw == 8 && pipe) MV_PF_LAUNCH(8, true);
  else if (nw == 8) MV_PF_LAUNCH(8, false);
  else if (pipe) MV_PF_LAUNCH(4, true);
  else MV_PF_LAUNCH(4, false);
#undef MV_PF_LAUNCH
  MV_LAUNCH_CHECK();
  return MV_OK;
}
