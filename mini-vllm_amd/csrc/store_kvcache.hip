// K1 — KV-cache scatter store (reference: attention_backend.py:50-78,196-266).
// One workgroup per token; every lane moves one 16-byte piece of the
// token's K row or V row.  Pure byte copy => bit-exact.
#include "common.h"

__global__ void __launch_bounds__(256)
store_kvcache_kernel(const char *__restrict__ k, const char *__restrict__ v,
                     int64_t k_stride_bytes, int64_t v_stride_bytes,
                     char *__restrict__ k_cache, char *__restrict__ v_cache,
                     const int32_t *__restrict__ slot_mapping,
                     int row_vecs /* 16-B pieces per row */, int64_t num_slots) {
  const int tok = blockIdx.x;
  const int64_t slot = slot_mapping[tok];
  if (slot < 0 || slot >= num_slots) return;  // -1 padding (graph replay) is skipped
  const int64_t row_bytes = (int64_t)row_vecs * 16;
  const char *ksrc = k + (int64_t)tok * k_stride_bytes;
  const char *vsrc = v + (int64_t)tok * v_stride_bytes;
  char *kdst = k_cache + slot * row_bytes;
  char *vdst = v_cache + slot * row_bytes;
  for (int c = threadIdx.x; c < 2 * row_vecs; c += blockDim.x) {
    const bool is_v = c >= row_vecs;
    const int piece = is_v ? c - row_vecs : c;
    const u32x4 val = *reinterpret_cast<const u32x4 *>((is_v ? vsrc : ksrc) + piece * 16);
    *reinterpret_cast<u32x4 *>((is_v ? vdst : kdst) + piece * 16) = val;
  }
}

extern "C" int mv_store_kvcache(const void *k, const void *v, int64_t k_stride_tok,
                                int64_t v_stride_tok, void *k_cache, void *v_cache,
                                const int32_t *slot_mapping, int n_tok, int n_kv_heads,
                                int head_dim, int64_t num_slots, int dtype, void *stream) {
  if (n_tok == 0) return MV_OK;
  if (!k || !v || !k_cache || !v_cache || !slot_mapping) return MV_E_NULLPTR;
  if (n_tok < 0 || n_kv_heads <= 0 || head_dim <= 0 || num_slots < 0) return MV_E_SHAPE;
  int esz;
  switch (dtype) {
    case MV_BF16: case MV_F16: esz = 2; break;
    case MV_F32: esz = 4; break;
    default: return MV_E_DTYPE;
  }
  const int64_t row_bytes = (int64_t)n_kv_heads * head_dim * esz;
  if (row_bytes % 16 || (k_stride_tok * esz) % 16 || (v_stride_tok * esz) % 16 ||
      !mv_aligned16(k) || !mv_aligned16(v) || !mv_aligned16(k_cache) || !mv_aligned16(v_cache))
    return MV_E_ALIGN;
  const int row_vecs = (int)(row_bytes / 16);
  int threads = 2 * row_vecs;
  threads = threads < 64 ? 64 : (threads > 256 ? 256 : (threads + 63) / 64 * 64);
  hipLaunchKernelGGL(store_kvcache_kernel, dim3(n_tok), dim3(threads), 0, (hipStream_t)stream,
                     (const char *)k, (const char *)v, k_stride_tok * esz, v_stride_tok * esz,
                     (char *)k_cache, (char *)v_cache, slot_mapping, row_vecs, num_slots);
  MV_LAUNCH_CHECK();
  return MV_OK;
}
