// K7 (+ per-head K6, + K1) — q/k head RMSNorm -> NeoX rotary embedding ->
// K/V cache store in ONE launch
// (reference: minivllm/models/qwen_base.py:173-190 flat path,
//  layers/layernorm.py:81-93, layers/rotary_embedding.py:77-81,166-193,
//  layers/attention_backend.py:196-266).
// One workgroup per token.  A head row (D elements) is spread over
// LPT = D/VEC adjacent lanes, 16 bytes each; the rotation partner
// (dims i and i + D/2) lives LPT/2 lanes away and is fetched with one DPP
// row rotate (bf16/f16, D=128) or a wave shuffle.
// Rounding points are the reference's: norm result -> dtype, rotation in
// fp32 with separately rounded products -> dtype.
#include "common.h"

struct RopeParams {
  const void *q, *k, *v;
  int64_t q_stride, k_stride, v_stride;
  const int64_t *positions;
  const float *cos, *sin;
  int64_t max_pos;
  const void *q_norm_w, *k_norm_w;
  float eps;
  void *q_out, *k_out, *k_cache, *v_cache;
  const int32_t *slot_mapping;
  int64_t num_slots;
  int n_heads, n_kv_heads;
};

template <int LPT>
__device__ __forceinline__ float lpt_sum(float x) {
  // butterfly over the LPT lanes of one head row; every lane gets the total
  for (int off = 1; off < LPT; off <<= 1) x += __shfl_xor(x, off, 64);
  return x;
}

template <typename Tag, int D>
__global__ void __launch_bounds__(256) rope_qk_store_kernel(RopeParams p) {
  typedef typename elem<Tag>::T T;
  constexpr int VEC = elem<Tag>::VEC;
  constexpr int LPT = D / VEC;
  constexpr int HALF = LPT / 2;
  const int tok = blockIdx.x;
  const int sub = threadIdx.x % LPT;
  const int rows_per_iter = blockDim.x / LPT;
  const int H = p.n_heads, HKV = p.n_kv_heads;
  const bool has_cache = p.k_cache != nullptr && p.slot_mapping != nullptr;
  int64_t slot = has_cache ? (int64_t)p.slot_mapping[tok] : -1;
  if (slot >= p.num_slots) slot = -1;
  const int total_rows = H + HKV + (slot >= 0 ? HKV : 0);

  const int64_t pos = p.positions[tok];
  const bool pos_ok = pos >= 0 && pos < p.max_pos;
  float cs[VEC], sn[VEC];
  {
    const int i0 = (sub % HALF) * VEC;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      cs[j] = pos_ok ? p.cos[pos * (D / 2) + i0 + j] : 1.0f;
      sn[j] = pos_ok ? p.sin[pos * (D / 2) + i0 + j] : 0.0f;
    }
  }

  for (int r = threadIdx.x / LPT; r < total_rows; r += rows_per_iter) {
    if (r >= H + HKV) {  // raw V row -> cache
      const int hv = r - H - HKV;
      const T *src = (const T *)p.v + (int64_t)tok * p.v_stride + hv * D + sub * VEC;
      T *dst = (T *)p.v_cache + (slot * HKV + hv) * D + sub * VEC;
      *reinterpret_cast<u32x4 *>(dst) = *reinterpret_cast<const u32x4 *>(src);
      continue;
    }
    const bool is_q = r < H;
    const int hh = is_q ? r : r - H;
    const T *src = is_q ? (const T *)p.q + (int64_t)tok * p.q_stride + hh * D
                        : (const T *)p.k + (int64_t)tok * p.k_stride + hh * D;
    const T *nw = is_q ? (const T *)p.q_norm_w : (const T *)p.k_norm_w;
    float x[VEC];
    {
      const u32x4 raw = *reinterpret_cast<const u32x4 *>(src + sub * VEC);
      if constexpr (VEC == 8) unpack8<Tag>(raw, x);
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = u2f(raw[j]);
      }
    }
    if (nw != nullptr) {  // per-head RMSNorm, result rounded to dtype like the reference
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) ss = fmaf(x[j], x[j], ss);
      ss = lpt_sum<LPT>(ss);
      const float rs = 1.0f / sqrtf(ss / (float)D + p.eps);
      float w[VEC];
      const u32x4 wraw = *reinterpret_cast<const u32x4 *>(nw + sub * VEC);
      if constexpr (VEC == 8) unpack8<Tag>(wraw, w);
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = u2f(wraw[j]);
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        x[j] = elem<Tag>::ld(elem<Tag>::st(__fmul_rn(__fmul_rn(x[j], rs), w[j])));
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
    u32x4 packed;
    if constexpr (VEC == 8) packed = pack8<Tag>(y);
    else {
#pragma unroll
      for (int j = 0; j < 4; ++j) packed[j] = f2u(y[j]);
    }
    if (is_q) {
      T *dst = (T *)p.q_out + ((int64_t)tok * H + hh) * D + sub * VEC;
      *reinterpret_cast<u32x4 *>(dst) = packed;
    } else {
      if (p.k_out != nullptr) {
        T *dst = (T *)p.k_out + ((int64_t)tok * HKV + hh) * D + sub * VEC;
        *reinterpret_cast<u32x4 *>(dst) = packed;
      }
      if (slot >= 0) {
        T *dst = (T *)p.k_cache + (slot * HKV + hh) * D + sub * VEC;
        *reinterpret_cast<u32x4 *>(dst) = packed;
      }
    }
  }
}

extern "C" int mv_rope_qk_store(const void *q, const void *k, const void *v, int64_t q_stride_tok,
                                int64_t k_stride_tok, int64_t v_stride_tok,
                                const int64_t *positions, const float *cos, const float *sin,
                                int64_t max_pos, const void *q_norm_w, const void *k_norm_w,
                                float eps, void *q_out, void *k_out, void *k_cache, void *v_cache,
                                const int32_t *slot_mapping, int64_t num_slots, int n_tok,
                                int n_heads, int n_kv_heads, int head_dim, int dtype,
                                void *stream) {
  if (n_tok == 0) return MV_OK;
  if (!q || !k || !positions || !cos || !sin || !q_out) return MV_E_NULLPTR;
  if ((q_norm_w == nullptr) != (k_norm_w == nullptr)) return MV_E_NULLPTR;
  const bool has_cache = k_cache != nullptr || v_cache != nullptr || slot_mapping != nullptr;
  if (has_cache && (!k_cache || !v_cache || !slot_mapping || !v)) return MV_E_NULLPTR;
  if (n_tok < 0 || n_heads <= 0 || n_kv_heads <= 0 || max_pos <= 0) return MV_E_SHAPE;
  if (head_dim != 64 && head_dim != 128) return MV_E_SHAPE;
  if (dtype != MV_BF16 && dtype != MV_F16 && dtype != MV_F32) return MV_E_DTYPE;
  const int esz = dtype == MV_F32 ? 4 : 2;
  if (!mv_aligned16(q) || !mv_aligned16(k) || !mv_aligned16(q_out) ||
      (v && !mv_aligned16(v)) || (k_out && !mv_aligned16(k_out)) ||
      (k_cache && !mv_aligned16(k_cache)) || (v_cache && !mv_aligned16(v_cache)) ||
      (q_norm_w && (!mv_aligned16(q_norm_w) || !mv_aligned16(k_norm_w))) ||
      (q_stride_tok * esz) % 16 || (k_stride_tok * esz) % 16 || (v_stride_tok * esz) % 16)
    return MV_E_ALIGN;
  RopeParams p;
  p.q = q; p.k = k; p.v = v;
  p.q_stride = q_stride_tok; p.k_stride = k_stride_tok; p.v_stride = v_stride_tok;
  p.positions = positions; p.cos = cos; p.sin = sin; p.max_pos = max_pos;
  p.q_norm_w = q_norm_w; p.k_norm_w = k_norm_w; p.eps = eps;
  p.q_out = q_out; p.k_out = k_out;
  p.k_cache = has_cache ? k_cache : nullptr;
  p.v_cache = has_cache ? v_cache : nullptr;
  p.slot_mapping = has_cache ? slot_mapping : nullptr;
  p.num_slots = num_slots;
  p.n_heads = n_heads; p.n_kv_heads = n_kv_heads;
  hipStream_t st = (hipStream_t)stream;
  if (head_dim == 128) {
    MV_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((rope_qk_store_kernel<Tag, 128>), dim3(n_tok), dim3(256), 0, st, p));
  } else {
    MV_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((rope_qk_store_kernel<Tag, 64>), dim3(n_tok), dim3(256), 0, st, p));
  }
  MV_LAUNCH_CHECK();
  return MV_OK;
}
