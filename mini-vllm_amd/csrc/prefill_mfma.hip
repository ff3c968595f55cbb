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
#include <utility>

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

// max / sum across the two half-waves that share a query row (lane r and lane r + 32): one v_permlane32_swap (a VALU
// instruction) instead of a ds_bpermute, whose compiler-inserted lgkmcnt(0) would also wait for every hand-counted
// fragment read in flight
__device__ __forceinline__ void half_swap(float x, float &lo, float &hi) {
  // as inline assembly: with the builtin and the same value in both operands hipcc (ROCm 7.2) treats the two results as
  // one value (ISA: `v_permlane32_swap v178, v0; v_add_f32 v0, v178, v178` for lo + hi).  `s_nop 1`: a VALU write of
  // an operand right in front of v_permlane* needs wait states the compiler does not insert inside a statement.
  lo = x;
  hi = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
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
    if constexpr (PIPE) {   // (no LDS operation here: a ds_bpermute's lgkmcnt(0) would drain the V^T reads meant to land under the exponentials)
      float lo, hi;
      half_swap(mx, lo, hi);
      mx = fmaxf(lo, hi);
    } else {
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    }
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

  // ---- epilogue: O[q][d] = O^T[d][q] / l ; lane (r,h) owns d = 32dt + 8c + 4h + {0..3} of row r: sixteen 8-byte
  //      pieces, each at another place of a 256-byte output row.  Stored like that (round 1-2) a wave instruction
  //      wrote 32 partial lines at a stride of n_heads x 256 B (a power of two for 16 heads: the same channels over and
  //      over): 5 % of the kernel at 4 x 4096, 9 % at 16 x 1024 (28/4), 30 % at 16 x 1024 (16/8) and 33 % at 64 x 256
  //      (timing ablation, profiles/r03_prefill_kernel.md).  Now the wave's 32 x 128 tile goes through its own 8 KiB of
  //      the (no longer needed) K|V buffers — 16-byte chunks XOR-swizzled by row so that both the 8-byte writes and the
  //      16-byte reads spread over the banks — and leaves as whole rows: one instruction = four contiguous 256-B rows.
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  // (every wave passed the barrier that ends the last tile: nobody reads the K|V buffers any more; a wave then only
  //  touches its own 8 KiB, so no further barrier)
  char *stage = smem + wave * (32 * ROW_BYTES);
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      u32x2 w;
      w[0] = pack2<Tag>(oacc[dt][4 * c + 0] * inv, oacc[dt][4 * c + 1] * inv);
      w[1] = pack2<Tag>(oacc[dt][4 * c + 2] * inv, oacc[dt][4 * c + 3] * inv);
      *reinterpret_cast<u32x2 *>(stage + r * ROW_BYTES + (((4 * dt + c) ^ (r & 15)) << 4) + 8 * h) = w;
    }
  {
    const int rr = lane >> 4, cc = lane & 15;   // four rows per instruction, sixteen 16-byte chunks per row
    const int row0 = qb * BM + wave * 32;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = 4 * it + rr;
      const u32x4 v = *reinterpret_cast<const u32x4 *>(stage + row * ROW_BYTES + ((cc ^ (row & 15)) << 4));
      if (row0 + row < sq && !(rescale_thr <= -2.f && lane != 0)) {   // (thr <= -2: bench ablation, timing only)
        T *op = (T *)p.out + ((int64_t)(q0 + row0 + row) * p.n_heads + head) * D;
        *reinterpret_cast<u32x4 *>(op + 8 * cc) = v;
      }
    }
  }
}


// ---- the 64-rows-per-wave kernel OWNS the accumulator half of the register file by name:
//        a[0:127]    O^T accumulators: block b, d-tile dt -> a[64 b + 16 dt .. + 15]
//        a[128:191]  Q^T fragments:    block b, k-step s  -> a[128 + 32 b + 4 s .. + 3]
//      while the score accumulators, P, the K / V^T fragments and all softmax arithmetic stay in the architectural half.
//      hipcc cannot express that split (one MFMA form per function; as "a"-constrained C++ values the accumulators were
//      copied between two AGPR sets on every tile and Q was spilled to scratch), so every instruction that touches
//      those registers is inline assembly naming them literally, and ONE statement (the zero fill) lists them as
//      clobbers so that the kernel descriptor allocates them.  hipcc itself must then never use an AGPR: the build is
//      audited for that (no v_accvgpr_* / a[...] outside ASMSTART..ASMEND, no scratch; tools/dbg/audit_pf64.sh).
//      Hazards the compiler no longer sees (cdna_hip_programming.md, inline assembly, item 2): a score accumulator is
//      read by VALU code only behind a later MFMA statement that names it as an operand (>= 3 MFMA issue slots after
//      its last writer); P fragments fresh from v_cvt_pk are covered by the `s_nop 1` that opens the PV statement; O is
//      read by VALU instructions only in the rescale branch / the epilogue (far behind the last PV MFMA; the epilogue
//      pads explicitly).
#define MV_PF64_ZERO_O() asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\tv_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0\n\tv_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0\n\tv_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0\n\tv_accvgpr_write_b32 a96, 0\n\tv_accvgpr_write_b32 a97, 0\n\tv_accvgpr_write_b32 a98, 0\n\tv_accvgpr_write_b32 a99, 0\n\tv_accvgpr_write_b32 a100, 0\n\tv_accvgpr_write_b32 a101, 0\n\tv_accvgpr_write_b32 a102, 0\n\tv_accvgpr_write_b32 a103, 0\n\tv_accvgpr_write_b32 a104, 0\n\tv_accvgpr_write_b32 a105, 0\n\tv_accvgpr_write_b32 a106, 0\n\tv_accvgpr_write_b32 a107, 0\n\tv_accvgpr_write_b32 a108, 0\n\tv_accvgpr_write_b32 a109, 0\n\tv_accvgpr_write_b32 a110, 0\n\tv_accvgpr_write_b32 a111, 0\n\tv_accvgpr_write_b32 a112, 0\n\tv_accvgpr_write_b32 a113, 0\n\tv_accvgpr_write_b32 a114, 0\n\tv_accvgpr_write_b32 a115, 0\n\tv_accvgpr_write_b32 a116, 0\n\tv_accvgpr_write_b32 a117, 0\n\tv_accvgpr_write_b32 a118, 0\n\tv_accvgpr_write_b32 a119, 0\n\tv_accvgpr_write_b32 a120, 0\n\tv_accvgpr_write_b32 a121, 0\n\tv_accvgpr_write_b32 a122, 0\n\tv_accvgpr_write_b32 a123, 0\n\tv_accvgpr_write_b32 a124, 0\n\tv_accvgpr_write_b32 a125, 0\n\tv_accvgpr_write_b32 a126, 0\n\tv_accvgpr_write_b32 a127, 0\n\ts_nop 1" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255")
template <typename Tag, int QLO>
__device__ __forceinline__ void mfma_qk0(f32x16 &d, const u32x4 &a) {
  if constexpr (__is_same(Tag, bf16_tag)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], 0" : "=&v"(d) : "v"(a), "i"(QLO), "i"(QLO + 3));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%2:%3], 0" : "=&v"(d) : "v"(a), "i"(QLO), "i"(QLO + 3));
}
template <typename Tag, int QLO>
__device__ __forceinline__ void mfma_qk(f32x16 &d, const u32x4 &a) {
  if constexpr (__is_same(Tag, bf16_tag)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0" : "+v"(d) : "v"(a), "i"(QLO), "i"(QLO + 3));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%2:%3], %0" : "+v"(d) : "v"(a), "i"(QLO), "i"(QLO + 3));
}
// ... the same, additionally naming two other accumulators: VALU code that reads THEM cannot be scheduled above this
template <typename Tag, int QLO>
__device__ __forceinline__ void mfma_qk_fence(f32x16 &d, const u32x4 &a, f32x16 &f0, f32x16 &f1) {
  if constexpr (__is_same(Tag, bf16_tag))
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %3, a[%4:%5], %0" : "+v"(d), "+v"(f0), "+v"(f1) : "v"(a), "i"(QLO), "i"(QLO + 3));
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %3, a[%4:%5], %0" : "+v"(d), "+v"(f0), "+v"(f1) : "v"(a), "i"(QLO), "i"(QLO + 3));
}
template <typename Tag, int OLO>
__device__ __forceinline__ void mfma_pv(const u32x4 &a, const u32x4 &b) {
  if constexpr (__is_same(Tag, bf16_tag))
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, a[%2:%3]" : : "v"(a), "v"(b), "i"(OLO), "i"(OLO + 15));
  else
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, a[%2:%3]" : : "v"(a), "v"(b), "i"(OLO), "i"(OLO + 15));
}
template <typename Tag, int OLO>
__device__ __forceinline__ void mfma_pv_fence(const u32x4 &a, const u32x4 &b, f32x16 &f0, f32x16 &f1) {
  if constexpr (__is_same(Tag, bf16_tag))
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%4:%5], %2, %3, a[%4:%5]" : "+v"(f0), "+v"(f1) : "v"(a), "v"(b), "i"(OLO), "i"(OLO + 15));
  else
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%4:%5], %2, %3, a[%4:%5]" : "+v"(f0), "+v"(f1) : "v"(a), "v"(b), "i"(OLO), "i"(OLO + 15));
}
// K fragments in a[192:255] (k-step s, key half kt -> a[192 + 8 s + 4 kt .. + 3]): read from LDS once per tile, A operand of
// both blocks' QK^T MFMAs (an LDS read may name AGPR destinations, and both MFMA inputs may come from the accumulator file)
template <int KLO, int OFF>
__device__ __forceinline__ void k_read_acc(uint32_t addr) {
  asm volatile("ds_read_b128 a[%1:%2], %0 offset:%3" : : "v"(addr), "i"(KLO), "i"(KLO + 3), "i"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_wait_plain() {
  asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N) : "memory");
}
template <typename Tag, int KLO, int QLO>
__device__ __forceinline__ void mfma_qk0_kk(f32x16 &d) {
  if constexpr (__is_same(Tag, bf16_tag)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%1:%2], a[%3:%4], 0" : "=v"(d) : "i"(KLO), "i"(KLO + 3), "i"(QLO), "i"(QLO + 3));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%1:%2], a[%3:%4], 0" : "=v"(d) : "i"(KLO), "i"(KLO + 3), "i"(QLO), "i"(QLO + 3));
}
template <typename Tag, int KLO, int QLO>
__device__ __forceinline__ void mfma_qk_kk(f32x16 &d) {
  if constexpr (__is_same(Tag, bf16_tag)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%1:%2], a[%3:%4], %0" : "+v"(d) : "i"(KLO), "i"(KLO + 3), "i"(QLO), "i"(QLO + 3));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%1:%2], a[%3:%4], %0" : "+v"(d) : "i"(KLO), "i"(KLO + 3), "i"(QLO), "i"(QLO + 3));
}
template <typename Tag, int KLO, int QLO>
__device__ __forceinline__ void mfma_qk_kk_fence(f32x16 &d, f32x16 &f0, f32x16 &f1) {
  if constexpr (__is_same(Tag, bf16_tag))
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%3:%4], a[%5:%6], %0" : "+v"(d), "+v"(f0), "+v"(f1) : "i"(KLO), "i"(KLO + 3), "i"(QLO), "i"(QLO + 3));
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%3:%4], a[%5:%6], %0" : "+v"(d), "+v"(f0), "+v"(f1) : "i"(KLO), "i"(KLO + 3), "i"(QLO), "i"(QLO + 3));
}
template <int QLO, typename P>
__device__ __forceinline__ void q_load(const P *ptr) {
  asm volatile("global_load_dwordx4 a[%1:%2], %0, off" : : "v"(ptr), "i"(QLO), "i"(QLO + 3) : "memory");
}
template <int I>
__device__ __forceinline__ void o_scale_one(float alpha) {
  float t;
  asm volatile("v_accvgpr_read_b32 %0, a[%2]\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a[%2], %0" : "=&v"(t) : "v"(alpha), "i"(I));
}
template <int BASE, int... I>
__device__ __forceinline__ void o_scale(float alpha, std::integer_sequence<int, I...>) {
  (o_scale_one<BASE + I>(alpha), ...);
}
template <int I>
__device__ __forceinline__ float o_read() {
  float t;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(t) : "i"(I));
  return t;
}

// ---- round 3: 64 query rows per wave, one wave per SIMD (the whole 512-register file) --------------------------------
// A workgroup = 4 waves = 256 query rows of one (sequence, query head); a wave owns TWO 32-row query blocks A and B.
// Per 64-key tile a wave issues 64 MFMAs in four phases, and the softmax of one block runs beside the MFMAs of the
// other instead of in front of its own:
//   QK_A  (16 MFMAs; the 16 K fragments are read from LDS once and stay in registers)
//   QK_B  (16 MFMAs on the same fragments)          || max / exp / sum / pack of A's scores
//   PV_A  (16 MFMAs; the 16 V^T fragments are read once and stay in registers)   || max / exp / sum / pack of B's
//   PV_B  (16 MFMAs on the same fragments)
// so a K / V fragment read and a K|V tile (and its LDS-DMA instructions) serve twice the rows of the 32-row kernel
// above.  The (rare) deferred rescale of O is a wave-uniform branch between a block's maximum and its exponentials.
// Layouts (S^T = K Q^T, P^T as B operand, V^T by transposed reads) are those of the kernel above.
template <typename Tag, int ABL>
__global__ void __launch_bounds__(256, 1) prefill_mfma64_kernel(PrefillParams p, float rescale_thr) {
  constexpr int ablate = ABL;   // bench switches (timing only): 1 no K/V refresh, 2 no exponentials, 4 no maxima
  constexpr int NW = 4, BM = 256, PP = 16 / NW;
  typedef typename elem<Tag>::T T;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int g = (lane >> 4) & 1, i16 = lane & 15;
  const int head = blockIdx.x, seq = blockIdx.y;
  const int qb = (int)gridDim.z - 1 - (int)blockIdx.z;   // heaviest query blocks first (see above)
  const int q0 = p.cu_q[seq], sq = p.cu_q[seq + 1] - q0;
  const int k0 = p.cu_k[seq], sk = p.cu_k[seq + 1] - k0;
  if (qb * BM >= sq) return;
  const int hkv = head / (p.n_heads / p.n_kv_heads);
  const int shift = sk - sq;

  const int qidx0 = qb * BM + wave * 64 + r;   // block A's row of this lane; block B's is 32 further
  const bool q_ok[2] = {qidx0 < sq, qidx0 + 32 < sq};
  // Q^T fragments (B operand of QK^T), loaded STRAIGHT into a[128:191] (a VMEM load may name AGPR destinations on
  // gfx90a+): they stay there for the whole kernel.  Rows past the end read row 0 (their results are never stored).
  MV_PF64_ZERO_O();
  {
    const T *qa = (const T *)p.q + (int64_t)(q0 + (q_ok[0] ? qidx0 : 0)) * p.q_stride + (int64_t)head * D + 8 * h;
    const T *qb_ = (const T *)p.q + (int64_t)(q0 + (q_ok[1] ? qidx0 + 32 : 0)) * p.q_stride + (int64_t)head * D + 8 * h;
    q_load<128>(qa);       q_load<132>(qa + 16);  q_load<136>(qa + 32);  q_load<140>(qa + 48);
    q_load<144>(qa + 64);  q_load<148>(qa + 80);  q_load<152>(qa + 96);  q_load<156>(qa + 112);
    q_load<160>(qb_);      q_load<164>(qb_ + 16); q_load<168>(qb_ + 32); q_load<172>(qb_ + 48);
    q_load<176>(qb_ + 64); q_load<180>(qb_ + 80); q_load<184>(qb_ + 96); q_load<188>(qb_ + 112);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};
  const float scale_log2 = p.scale * 1.4426950408889634f;

  const int q_last = min(sq, (qb + 1) * BM) - 1;
  const int kv_len = min(sk, q_last + shift + 1);
  const int n_tiles = (kv_len + BN - 1) / BN;
  const int wave_qpos_min = shift + qb * BM + wave * 64;
  const int wave_qpos_max = wave_qpos_min + 63;

  const bool paged = p.block_tables != nullptr;
  const T *ksrc = paged ? (const T *)p.k_cache : (const T *)p.k;
  const T *vsrc = paged ? (const T *)p.v_cache : (const T *)p.v;
  const int64_t kstride = paged ? (int64_t)p.n_kv_heads * D : p.k_stride;
  const int64_t vstride = paged ? (int64_t)p.n_kv_heads * D : p.v_stride;
  int64_t krel[PP], vrel[PP];
#pragma unroll
  for (int i = 0; i < PP; ++i) {
    const int row = 4 * (NW * i + wave) + (lane >> 4), pc = lane & 15;
    krel[i] = (int64_t)row * kstride + (pc ^ (row & 15)) * 8;
    vrel[i] = (int64_t)row * vstride + (pc ^ (((row & 3) << 2) | ((row >> 2) & 3))) * 8;
  }
  bool tile_ok_next = true;
  auto table_entry = [&](int bi) -> int {   // through the vector memory path: no scalar load beside the counted LDS waits
    const int32_t *e = p.block_tables + (int64_t)seq * p.table_stride + bi;
    return __builtin_amdgcn_readfirstlane(e[__builtin_amdgcn_mbcnt_lo(0u, 0u)]);
  };
  // the LDS-DMA of a tile in two parts: dma_prep (scalar: where the tile lives) and eight dma_piece calls (one 1-KiB
  // piece of K or V each), so that the pieces of tile t+1 can sit between the MFMAs of tile t's first phase instead of
  // in front of them (an LDS-DMA instruction holds the issue port for ~60 cycles: two MFMAs cover it)
  const T *kt_n = nullptr, *vt_n = nullptr;
  char *dk_n = nullptr;
  int last_n = 0;
  bool full_n = true;
  auto dma_prep = [&](int t, int buf) {
    const int kv0 = t * BN;
    int64_t base_row = k0 + kv0;
    tile_ok_next = true;
    if (paged) {
      const int bi = kv0 / p.block_size;
      const int blk = table_entry(bi);
      tile_ok_next = blk >= 0 && blk < p.num_blocks;   // a hole: read block 0, mask the whole tile
      base_row = (int64_t)(tile_ok_next ? blk : 0) * p.block_size + (kv0 - bi * p.block_size);
    }
    kt_n = ksrc + base_row * kstride + hkv * D;
    vt_n = vsrc + base_row * vstride + hkv * D;
    dk_n = smem + buf * 2 * TILE_BYTES;
    full_n = kv0 + BN <= sk;
    last_n = sk - 1 - kv0;   // the ragged last tile: rows past the end re-read the last valid key
  };
  auto dma_piece = [&](auto I, auto ISV) __attribute__((always_inline)) {
    constexpr int i = decltype(I)::value;
    constexpr bool is_v = decltype(ISV)::value;
    const T *src;
    if (full_n) {   // a full tile: one scalar base + the per-lane constants
      src = is_v ? vt_n + vrel[i] : kt_n + krel[i];
    } else {
      const int row = 4 * (NW * i + wave) + (lane >> 4), pc = lane & 15;
      const int srow = min(row, last_n);
      const int ch = is_v ? (pc ^ (((row & 3) << 2) | ((row >> 2) & 3))) : (pc ^ (row & 15));
      src = (is_v ? vt_n + (int64_t)srow * vstride : kt_n + (int64_t)srow * kstride) + ch * 8;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)(dk_n + (is_v ? TILE_BYTES : 0) + (NW * i + wave) * 1024),
                                     16, 0, 0);
  };
  auto dma_all = [&]() __attribute__((always_inline)) {
    dma_piece(std::integral_constant<int, 0>{}, std::false_type{}); dma_piece(std::integral_constant<int, 0>{}, std::true_type{});
    dma_piece(std::integral_constant<int, 1>{}, std::false_type{}); dma_piece(std::integral_constant<int, 1>{}, std::true_type{});
    dma_piece(std::integral_constant<int, 2>{}, std::false_type{}); dma_piece(std::integral_constant<int, 2>{}, std::true_type{});
    dma_piece(std::integral_constant<int, 3>{}, std::false_type{}); dma_piece(std::integral_constant<int, 3>{}, std::true_type{});
  };

  bool tile_ok = true;
  if (n_tiles > 0) {
    dma_prep(0, 0);
    dma_all();
    tile_ok = tile_ok_next;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // per-lane LDS read addresses, relative to buffer 0 (buffer / key half / key group are instruction immediates)
  uint32_t ka0[NS], va_lo[NDT], va_hi[NDT];
  {
    const uint32_t base = lds_addr(smem);
    const int q4 = i16 >> 2, pp = i16 & 3;
#pragma unroll
    for (int s = 0; s < NS; ++s) ka0[s] = base + r * ROW_BYTES + (((2 * s + h) ^ (r & 15)) << 4);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int ch = 4 * dt + 2 * g + (pp >> 1);
      va_lo[dt] = base + TILE_BYTES + v_off(4 * h + q4, ch) + 8 * (pp & 1);
      va_hi[dt] = base + TILE_BYTES + v_off(4 * h + q4 + 8, ch) + 8 * (pp & 1);
    }
  }

  auto tile = [&](const int t, auto BUF, auto MASKED, const bool has_next) __attribute__((always_inline)) {
    constexpr int BO = decltype(BUF)::value * 2 * TILE_BYTES;
    constexpr bool MASK = decltype(MASKED)::value;
    const int kv0 = t * BN;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 sacc[2][2];
    u32x4 pf[2][2][2];
    float m_new[2];
    // ---- the softmax of block b in three pieces
    auto sm_max = [&](auto Bc) __attribute__((always_inline)) -> bool {
      constexpr int b = decltype(Bc)::value;
      if constexpr ((ablate & 4) != 0) return false;
      if constexpr (MASK) {
        const int klim = tile_ok ? min(qidx0 + 32 * b + shift, sk - 1) : -1;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = kv0 + kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            sacc[b][kt][i] = key <= klim ? sacc[b][kt][i] : -INFINITY;
          }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[b][kt][i]);
      {
        float lo, hi;
        half_swap(mx, lo, hi);
        mx = fmaxf(lo, hi);
      }
      m_new[b] = fmaxf(m_run[b], mx * scale_log2);
      return !__all(m_new[b] - m_run[b] <= rescale_thr);
    };
    auto sm_rescale = [&](auto Bc) __attribute__((always_inline)) {
      constexpr int b = decltype(Bc)::value;
      const float alpha = __builtin_amdgcn_exp2f(m_run[b] - m_new[b]);
      m_run[b] = m_new[b];
      l_run[b] *= alpha;
      o_scale<64 * b>(alpha, std::make_integer_sequence<int, 64>{});
    };
    // the exponentials of block b in sixteen chunks of two scores (2 fma + 2 exp + 2 add + 1 cvt_pk), one behind an MFMA
    // of the other block's phase; chunk c covers scores 2c, 2c+1 of the 32 this lane holds (key group c / 4);
    // rs = two partial row sums carried across the chunks
    float rs[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    auto sm_exp = [&](auto Bc, auto Cc) __attribute__((always_inline)) {
      constexpr int b = decltype(Bc)::value, c = decltype(Cc)::value;
      constexpr int kt = c >> 3, i0 = 2 * (c & 7), s2 = (c & 7) >> 2, j = c & 3;
      if constexpr ((ablate & 2) != 0) return;
      const float nm = -m_run[b];
      const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[b][kt][i0], scale_log2, nm));
      const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[b][kt][i0 + 1], scale_log2, nm));
      rs[b][0] += e0;
      rs[b][1] += e1;
      pf[b][kt][s2][j] = pack2<Tag>(e0, e1);
      // (an empty volatile statement that names the results: LLVM otherwise SINKS a chunk into the later basic block that
      // holds its consumer -- found in the ISA: eight of A's chunks sat in one cluster behind B's rescale branch)
      asm volatile("" : "+v"(pf[b][kt][s2][j]), "+v"(rs[b][0]), "+v"(rs[b][1]));
      if constexpr (c == 15) l_run[b] += rs[b][0] + rs[b][1];
    };
    constexpr std::integral_constant<int, 0> A{};
    constexpr std::integral_constant<int, 1> B{};
#define MV_PIN() __builtin_amdgcn_sched_barrier(0)
#define MV_IC(n) std::integral_constant<int, (n)>{}

    // ---- QK_A then QK_B as ONE run of 16 k-steps.  The K fragments go LDS -> a[192:255] once per tile, three k-steps
    //      ahead of A's MFMAs (one wave per SIMD: nobody else hides an LDS round trip), and serve B's MFMAs from there;
    //      the first two groups of V^T fragments go out during B's k-steps.  A's maximum follows B's second k-step; one
    //      chunk of A's exponentials follows each of B's remaining twelve MFMAs (the last four chunks run behind PV_A's
    //      first group, which needs key group 0 only)
    u32x2 vlo[4][NDT], vhi[4][NDT];     // V^T fragments of the tile's four 16-key groups: read for A, kept for B
    auto k_read = [&](auto S) __attribute__((always_inline)) {
      constexpr int sk_ = decltype(S)::value;
      k_read_acc<192 + 8 * sk_, BO>(ka0[sk_]);
      k_read_acc<192 + 8 * sk_ + 4, BO + 32 * ROW_BYTES>(ka0[sk_]);
    };
    auto v_read = [&](auto G, auto DT) __attribute__((always_inline)) {
      constexpr int gi = decltype(G)::value, dt = decltype(DT)::value;
      ds_r64tr<BO + gi * 16 * ROW_BYTES>(vlo[gi][dt], va_lo[dt]);
      ds_r64tr<BO + gi * 16 * ROW_BYTES>(vhi[gi][dt], va_hi[dt]);
    };
    auto v_read_group = [&](auto G) __attribute__((always_inline)) {
      v_read(G, MV_IC(0)); v_read(G, MV_IC(1)); v_read(G, MV_IC(2)); v_read(G, MV_IC(3));
    };
    k_read(MV_IC(0));
    k_read(MV_IC(1));
    k_read(MV_IC(2));
    auto qk_step = [&](auto S16) __attribute__((always_inline)) {
      constexpr int s16 = decltype(S16)::value;
      constexpr int s = s16 & (NS - 1), b = s16 >> 3;
      if constexpr (s16 + 3 < NS) k_read(MV_IC(s16 + 3));
      if constexpr (s16 < NS) lgkm_wait_plain<(s16 + 3 < NS ? 6 : 2 * (NS - 1 - s16))>();   // this step's two reads have landed
      if constexpr (s16 == NS + 2) v_read_group(MV_IC(0));      // (nothing waits on these before PV_A)
      if constexpr (s16 == NS + 5) v_read_group(MV_IC(1));
      constexpr int KLO = 192 + 8 * s, QLO = 128 + 32 * b + 4 * s;
      if constexpr (s == 0) {
        mfma_qk0_kk<Tag, KLO, QLO>(sacc[b][0]);
        mfma_qk0_kk<Tag, KLO + 4, QLO>(sacc[b][1]);
      } else if constexpr (s16 < NS) {
        mfma_qk_kk<Tag, KLO, QLO>(sacc[b][0]);
        mfma_qk_kk<Tag, KLO + 4, QLO>(sacc[b][1]);
      } else if constexpr (s16 == NS + 1) {
        mfma_qk_kk<Tag, KLO, QLO>(sacc[b][0]);
        mfma_qk_kk_fence<Tag, KLO + 4, QLO>(sacc[b][1], sacc[0][0], sacc[0][1]);   // A's scores are final
      } else {
        mfma_qk_kk<Tag, KLO, QLO>(sacc[b][0]);
        if constexpr (s16 >= NS + 2) { MV_PIN(); sm_exp(A, MV_IC(2 * (s16 - NS - 2))); MV_PIN(); }
        mfma_qk_kk<Tag, KLO + 4, QLO>(sacc[b][1]);
        if constexpr (s16 >= NS + 2) { MV_PIN(); sm_exp(A, MV_IC(2 * (s16 - NS - 2) + 1)); MV_PIN(); }
      }
      if constexpr (s16 < NS) {   // one piece of the next tile's K (even steps) / V (odd steps) behind each of A's k-steps
        MV_PIN();
        if (has_next) dma_piece(MV_IC(s16 >> 1), std::integral_constant<bool, (s16 & 1) != 0>{});
        MV_PIN();
      }
      if constexpr (s16 == NS + 1) {
        if (sm_max(A)) sm_rescale(A);
        MV_PIN();
      }
    };
    qk_step(MV_IC(0));  qk_step(MV_IC(1));  qk_step(MV_IC(2));  qk_step(MV_IC(3));
    qk_step(MV_IC(4));  qk_step(MV_IC(5));  qk_step(MV_IC(6));  qk_step(MV_IC(7));
    qk_step(MV_IC(8));  qk_step(MV_IC(9));  qk_step(MV_IC(10)); qk_step(MV_IC(11));
    qk_step(MV_IC(12)); qk_step(MV_IC(13)); qk_step(MV_IC(14)); qk_step(MV_IC(15));
    // ---- PV_A then PV_B (32 MFMAs): A's MFMAs read the V^T fragments two groups ahead and keep them; B's use them from
    //      registers.  Slots: 0-3 A's last four chunks; 4-7 bare, then B's maximum; 8-23 B's sixteen chunks (PV_B's
    //      group g at slot 16 + 4 g needs chunks 0 .. 4 g + 3); 24-31 bare
    auto pv_slot = [&](auto M) __attribute__((always_inline)) {
      constexpr int m = decltype(M)::value;
      constexpr int g8 = m >> 2, dt = m & 3, b = g8 >> 2, gi = g8 & 3;
      if constexpr (b == 0) {
        // reads behind (gi, dt)'s own pair: the rest of group gi, group gi+1, and group gi+2's first dt pairs
        constexpr int newer = gi <= 1 ? 14 : (gi == 2 ? 14 - 2 * dt : 6 - 2 * dt);
        lgkm_wait1<newer>(vlo[gi][dt], vhi[gi][dt]);
      }
      const u32x4 a = {vlo[gi][dt].x, vlo[gi][dt].y, vhi[gi][dt].x, vhi[gi][dt].y};
      if constexpr (m == 7) mfma_pv_fence<Tag, 64 * b + 16 * dt>(a, pf[b][gi >> 1][gi & 1], sacc[1][0], sacc[1][1]);   // B's scores are final
      else mfma_pv<Tag, 64 * b + 16 * dt>(a, pf[b][gi >> 1][gi & 1]);
      if constexpr (b == 0 && gi + 2 < 4) v_read(MV_IC(gi + 2), MV_IC(dt));
      if constexpr (m < 4) { MV_PIN(); sm_exp(A, MV_IC(12 + m)); MV_PIN(); }
      if constexpr (m >= 8 && m < 24) { MV_PIN(); sm_exp(B, MV_IC(m - 8)); MV_PIN(); }
      if constexpr (m == 7) {
        if (sm_max(B)) sm_rescale(B);
        MV_PIN();
      }
    };
    pv_slot(MV_IC(0));  pv_slot(MV_IC(1));  pv_slot(MV_IC(2));  pv_slot(MV_IC(3));
    pv_slot(MV_IC(4));  pv_slot(MV_IC(5));  pv_slot(MV_IC(6));  pv_slot(MV_IC(7));
    pv_slot(MV_IC(8));  pv_slot(MV_IC(9));  pv_slot(MV_IC(10)); pv_slot(MV_IC(11));
    pv_slot(MV_IC(12)); pv_slot(MV_IC(13)); pv_slot(MV_IC(14)); pv_slot(MV_IC(15));
    pv_slot(MV_IC(16)); pv_slot(MV_IC(17)); pv_slot(MV_IC(18)); pv_slot(MV_IC(19));
    pv_slot(MV_IC(20)); pv_slot(MV_IC(21)); pv_slot(MV_IC(22)); pv_slot(MV_IC(23));
    pv_slot(MV_IC(24)); pv_slot(MV_IC(25)); pv_slot(MV_IC(26)); pv_slot(MV_IC(27));
    pv_slot(MV_IC(28)); pv_slot(MV_IC(29)); pv_slot(MV_IC(30)); pv_slot(MV_IC(31));
#undef MV_PIN
#undef MV_IC
  };
  auto step = [&](const int t, auto BUF) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // nothing of the prologue may still be counted
    const bool has_next = t + 1 < n_tiles && !((ablate & 1) && t > 0);   // (ablate: bench switches, timing only)
    if (has_next) dma_prep(t + 1, (t + 1) & 1);
    const int kv0 = t * BN;
    if (kv0 <= wave_qpos_max) {
      if ((kv0 + BN - 1 > wave_qpos_min) || (kv0 + BN > sk) || !tile_ok) tile(t, BUF, std::true_type{}, has_next);
      else tile(t, BUF, std::false_type{}, has_next);
    } else if (has_next) {
      dma_all();   // masked out for every row of this wave: its share of the DMA and the barrier only
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tile_ok = tile_ok_next;
    __syncthreads();
  };
  for (int t = 0; t < n_tiles; t += 2) {
    step(t, std::integral_constant<int, 0>{});
    if (t + 1 < n_tiles) step(t + 1, std::integral_constant<int, 1>{});
  }

  // ---- epilogue: O[q][d] = O^T[d][q] / l ; lane (r,h) owns d = 32dt + 8c + 4h + {0..3}
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // the last PV MFMAs (opaque to hipcc) drain before O is read
  auto store_block = [&](auto Bc) __attribute__((always_inline)) {
    constexpr int b = decltype(Bc)::value;
    float l_lo, l_hi;
    half_swap(l_run[b], l_lo, l_hi);
    const float l_tot = l_lo + l_hi;
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    auto quad = [&](auto DT, auto C) __attribute__((always_inline)) -> u32x2 {
      constexpr int base = 64 * b + 16 * decltype(DT)::value + 4 * decltype(C)::value;
      u32x2 w;
      w[0] = pack2<Tag>(o_read<base + 0>() * inv, o_read<base + 1>() * inv);
      w[1] = pack2<Tag>(o_read<base + 2>() * inv, o_read<base + 3>() * inv);
      return w;
    };
    T *op = (T *)p.out + ((int64_t)(q0 + qidx0 + 32 * b) * p.n_heads + head) * D;
    auto row = [&](auto DT) __attribute__((always_inline)) {
      constexpr int dt = decltype(DT)::value;
      const u32x2 w0 = quad(DT, std::integral_constant<int, 0>{}), w1 = quad(DT, std::integral_constant<int, 1>{});
      const u32x2 w2 = quad(DT, std::integral_constant<int, 2>{}), w3 = quad(DT, std::integral_constant<int, 3>{});
      if (q_ok[b]) {
        *reinterpret_cast<u32x2 *>(op + 32 * dt + 0 + 4 * h) = w0;
        *reinterpret_cast<u32x2 *>(op + 32 * dt + 8 + 4 * h) = w1;
        *reinterpret_cast<u32x2 *>(op + 32 * dt + 16 + 4 * h) = w2;
        *reinterpret_cast<u32x2 *>(op + 32 * dt + 24 + 4 * h) = w3;
      }
    };
    row(std::integral_constant<int, 0>{});
    row(std::integral_constant<int, 1>{});
    row(std::integral_constant<int, 2>{});
    row(std::integral_constant<int, 3>{});
  };
  store_block(std::integral_constant<int, 0>{});
  store_block(std::integral_constant<int, 1>{});
}

}  // namespace

int mv_prefill_mfma_launch(const PrefillParams &p, int dtype, int variant, hipStream_t stream) {
  if (p.head_dim != D || (dtype != MV_BF16 && dtype != MV_F16)) return MV_E_SHAPE;
  if (p.block_tables && (p.block_size % BN) != 0) return MV_E_SHAPE;  // a 64-key tile must sit in one block
  // 8 waves (256 query rows) per workgroup share a K/V tile where the sequences are long enough that the larger
  // blocks still fill the chip evenly; short prompts keep 4 waves x 2 workgroups per CU (measured per shape,
  // profiles/r02_prefill_kernel.md).  variant (bench switch, flags >> 4): 1 = force 4 waves, 2 = force 8
  // round 3: variants 3 / 4 = the round-2 loop (compiler-placed LDS reads) with 4 / 8 waves, for comparison
#ifndef MV_BENCH_ABLATE
  // timing ablations whose results are garbage (no K/V refresh, no output stores, the 64-row kernel's ablations) exist
  // only in a bench build (make ABLATE=1); the variants that stay compute the same attention
  if (variant & (16 | (3 << 8) | (1 << 10))) return MV_E_RANGE;
#endif
  const int variant_in = variant;
  const bool no_defer = (variant & 8) != 0;   // bench switch: rescale on every tile whose maximum moved
  const bool no_dma = (variant & 16) != 0;    // bench ablation: K/V tiles are not refreshed (results are garbage)
  const int ablate64 = (no_dma ? 1 : 0) | (((variant >> 8) & 3) << 1);   // bench ablations of the 64-rows-per-wave kernel
  const float thr_pick = 6.0f + 2.0f * (float)((variant >> 6) & 3);   // bench switch: rescale threshold 6 / 8 / 10 / 12
  variant &= 7;
  if (variant == 5) {   // bench switch: 64 query rows per wave, one wave per SIMD (profiles/r03_prefill_kernel.md)
    const int qblocks = (p.max_seqlen_q + 255) / 256;
    if (qblocks == 0 || p.n_seqs == 0) return MV_OK;
    const float thr64 = no_defer ? 0.0f : thr_pick;
    dim3 grid64(p.n_heads, p.n_seqs, qblocks);
    const dim3 b256(256);
    const size_t lds64 = 4 * TILE_BYTES;
    if (dtype == MV_F16) hipLaunchKernelGGL((prefill_mfma64_kernel<f16_tag, 0>), grid64, b256, lds64, stream, p, thr64);
    else if (ablate64 == 1) hipLaunchKernelGGL((prefill_mfma64_kernel<bf16_tag, 1>), grid64, b256, lds64, stream, p, thr64);
    else if (ablate64 == 6) hipLaunchKernelGGL((prefill_mfma64_kernel<bf16_tag, 6>), grid64, b256, lds64, stream, p, thr64);
    else if (ablate64 == 7) hipLaunchKernelGGL((prefill_mfma64_kernel<bf16_tag, 7>), grid64, b256, lds64, stream, p, thr64);
    else hipLaunchKernelGGL((prefill_mfma64_kernel<bf16_tag, 0>), grid64, b256, lds64, stream, p, thr64);
    MV_LAUNCH_CHECK();
    return MV_OK;
  }
  const bool pipe = variant != 3 && variant != 4;
  // rescale threshold (log2 units): the O accumulator is rescaled only when a row maximum grows by more than a factor of
  // 2^6 (+5.5 % at 4 x 4096; P then holds values up to 64, well inside bf16 / f16 range; same relative rounding)
  // (bench ablations, timing only: bit 4 = no K/V refresh (thr -1), bit 10 = additionally no output stores (thr -2))
  const float thr = ((variant_in >> 10) & 1) ? -2.0f : (no_dma ? -1.0f : ((pipe && !no_defer) ? thr_pick : 0.0f));
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
