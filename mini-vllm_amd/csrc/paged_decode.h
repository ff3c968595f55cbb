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

// MFMA variant (16-bit dtypes, D = 128, >= 4 query heads per kv head); g_chunk <= 16 heads per workgroup;
// klds: 1 / 0 = K through LDS or straight to registers, -1 = chosen from the grid
int mv_paged_decode_mfma_launch(const DecodeParams &p, int n_chunks, int nsplit, int batch, int dtype,
                                int klds, hipStream_t stream);
