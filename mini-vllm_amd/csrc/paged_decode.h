// Shared parameter block of the paged-decode kernels (K3).
#pragma once
#include "common.h"

struct DecodeParams {
  const void *q;
  const void *k_cache;
  const void *v_cache;
  const int32_t *block_tables;
  const int32_t *context_lens;
  void *out;
  float *ws_acc;  // [B, H, nsplit, D]
  float *ws_ml;   // [B, H, nsplit, 2]
  int64_t q_stride;
  int64_t table_stride;
  int64_t num_blocks;
  int n_heads;
  int n_kv_heads;
  int g_total;   // q heads per kv head
  int block_size;
  int bs_shift;  // log2(block_size) or -1
  int max_blocks;
  int nsplit;
  float scale_log2;  // softmax scale * log2(e)
  // ROPE (mv_rope_paged_decode): q points at the RAW query rows of the qkv projection; the kernel applies the
  // per-head RMSNorm (optional) and the rotary embedding itself, and the workgroup that owns a sequence's last
  // token first writes that token's K (normed, rotated) and V rows into the cache — the arithmetic of
  // rope_qk_store_kernel (rope.hip), which this replaces for decode steps
  const void *k_new = nullptr, *v_new = nullptr;   // raw [B, Hkv, D] rows (views of the qkv output)
  int64_t k_stride = 0, v_stride = 0;
  const int64_t *positions = nullptr;
  const float *cos = nullptr, *sin = nullptr;
  int64_t max_pos = 0;
  const void *q_norm_w = nullptr, *k_norm_w = nullptr;
  float eps = 0.f;
  const int32_t *slot_mapping = nullptr;
};

// ---- accesses to bytes that ANOTHER workgroup of the same launch wrote or will read (persistent decode kernel) -------
// COH = false: plain accesses (one kernel per phase: the kernel boundary makes everything visible).  COH = true: the
// fence-free hand-off of MI355X_MICROARCH.md "Valid forms", table row 1 — every store write-through (`sc1`: relaxed
// agent-scope atomics of 4 / 8 bytes lower to global_store_dword[x2] ... sc1), every load an `sc1` load that bypasses this
// CU's L1; the storing waves drain (s_waitcnt vmcnt(0)) and meet at a workgroup barrier before ONE lane signals, readers
// load behind the poll and a workgroup barrier (decode_persistent.hip: grid_barrier).  Measured: tools/dbg/barrier_probe.hip.
template <bool COH>
__device__ __forceinline__ u32x4 ld16(const void *p) {
  if constexpr (!COH) {
    return *reinterpret_cast<const u32x4 *>(p);
  } else {
    uint64_t *q = (uint64_t *)p;
    const uint64_t a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return u32x4{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
  }
}
template <bool COH>
__device__ __forceinline__ void st16(void *p, const u32x4 &v) {
  if constexpr (!COH) {
    *reinterpret_cast<u32x4 *>(p) = v;
  } else {
    uint64_t *q = (uint64_t *)p;
    __hip_atomic_store(q, (uint64_t)v[0] | ((uint64_t)v[1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, (uint64_t)v[2] | ((uint64_t)v[3] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
template <bool COH>
__device__ __forceinline__ void st8(void *p, const u32x2 &v) {
  if constexpr (!COH) *reinterpret_cast<u32x2 *>(p) = v;
  else __hip_atomic_store((uint64_t *)p, (uint64_t)v[0] | ((uint64_t)v[1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool COH>
__device__ __forceinline__ float ldf(const float *p) {
  if constexpr (!COH) return *p;
  else return u2f(__hip_atomic_load((uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
template <bool COH>
__device__ __forceinline__ void stf(float *p, float v) {
  if constexpr (!COH) *p = v;
  else __hip_atomic_store((uint32_t *)p, f2u(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool COH>
__device__ __forceinline__ f32x4 ldf4(const float *p) {
  if constexpr (!COH) {
    return *reinterpret_cast<const f32x4 *>(p);
  } else {
    const u32x4 r = ld16<true>(p);
    return f32x4{u2f(r[0]), u2f(r[1]), u2f(r[2]), u2f(r[3])};
  }
}
// a 2-byte element: there is no 16-bit atomic store; the write-through form is the instruction itself
template <bool COH>
__device__ __forceinline__ void st2(uint16_t *p, uint16_t v) {
  if constexpr (!COH) *p = v;
  else asm volatile("global_store_short %0, %1, off sc1" ::"v"(p), "v"((uint32_t)v) : "memory");
}

// one head row (D elements on LPT lanes, VEC per lane): optional RMSNorm over the row (result rounded to dtype) and
// the NeoX rotation by (cs, sn) — rope_qk_store_kernel's arithmetic and rounding points, statement for statement
template <typename Tag, int D>
__device__ __forceinline__ u32x4 norm_rope_row(const u32x4 &raw, const typename elem<Tag>::T *nw, float eps, int sub,
                                               const float (&cs)[elem<Tag>::VEC], const float (&sn)[elem<Tag>::VEC]) {
  constexpr int VEC = elem<Tag>::VEC, LPT = D / VEC, HALF = LPT / 2;
  static_assert(VEC == 8, "16-bit dtypes");
  float x[VEC];
  unpack8<Tag>(raw, x);
  if (nw != nullptr) {
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) ss = fmaf(x[j], x[j], ss);
    for (int off = 1; off < LPT; off <<= 1) ss += __shfl_xor(ss, off, 64);
    const float rs = 1.0f / sqrtf(ss / (float)D + eps);
    float w[VEC];
    const u32x4 wraw = *(const MV_AS1 u32x4 *)(nw + sub * VEC);
    unpack8<Tag>(wraw, w);
#pragma unroll
    for (int j = 0; j < VEC; ++j) x[j] = elem<Tag>::ld(elem<Tag>::st(__fmul_rn(__fmul_rn(x[j], rs), w[j])));
  }
  float y[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float other;
    if constexpr (LPT == 16) other = row_ror<8>(x[j]);
    else other = __shfl_xor(x[j], HALF, 64);
    const float a = __fmul_rn(x[j], cs[j]);
    const float b = __fmul_rn(other, sn[j]);
    y[j] = sub < HALF ? __fsub_rn(a, b) : __fadd_rn(a, b);
  }
  return pack8<Tag>(y);
}

// MFMA variant (16-bit dtypes, D = 128, >= 4 query heads per kv head); g_chunk <= 16 heads per workgroup;
// klds: 1 / 0 = K through LDS or straight to registers, -1 = chosen from the grid
int mv_paged_decode_mfma_launch(const DecodeParams &p, int n_chunks, int nsplit, int batch, int dtype,
                                int klds, hipStream_t stream);
