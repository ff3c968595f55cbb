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
    unpack8<Tag>(*reinterpret_cast<const u32x4 *>(nw + sub * VEC), w);
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
