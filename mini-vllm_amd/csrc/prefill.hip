// K2 — C-ABI entry of the packed varlen causal prefill attention
// (reference call sites: minivllm/models/layers/attention.py:422-433 flash
//  varlen, :467-504 fallback).  Dispatches to the MFMA flash kernel for
// 16-bit tensors and to the vector-ALU kernel otherwise.
#include "prefill.h"

extern "C" int mv_prefill_varlen(const void *q, int64_t q_stride_tok, const void *k, const void *v,
                                 int64_t k_stride_tok, int64_t v_stride_tok, const void *k_cache,
                                 const void *v_cache, const int32_t *block_tables,
                                 int64_t table_stride, int64_t num_blocks, void *out,
                                 const int32_t *cu_seqlens_q, const int32_t *cu_seqlens_k,
                                 int n_seqs, int max_seqlen_q, int max_seqlen_k, int n_heads,
                                 int n_kv_heads, int head_dim, int block_size, float scale,
                                 int dtype, int flags, void *stream) {
  if (n_seqs == 0 || max_seqlen_q == 0) return MV_OK;
  if (!q || !out || !cu_seqlens_q || !cu_seqlens_k) return MV_E_NULLPTR;
  const bool paged = block_tables != nullptr;
  if (paged && (!k_cache || !v_cache)) return MV_E_NULLPTR;
  if (!paged && (!k || !v)) return MV_E_NULLPTR;
  if (n_seqs < 0 || max_seqlen_q < 0 || max_seqlen_k < 0 || n_heads <= 0 || n_kv_heads <= 0 ||
      n_heads % n_kv_heads || (paged && (block_size <= 0 || num_blocks <= 0)))
    return MV_E_SHAPE;
  if (head_dim != 64 && head_dim != 128) return MV_E_SHAPE;
  if (dtype != MV_BF16 && dtype != MV_F16 && dtype != MV_F32) return MV_E_DTYPE;
  const int esz = dtype == MV_F32 ? 4 : 2;
  if (!mv_aligned16(q) || !mv_aligned16(out) || (q_stride_tok * esz) % 16 ||
      (!paged && (!mv_aligned16(k) || !mv_aligned16(v) || (k_stride_tok * esz) % 16 ||
                  (v_stride_tok * esz) % 16)) ||
      (paged && (!mv_aligned16(k_cache) || !mv_aligned16(v_cache))))
    return MV_E_ALIGN;
  PrefillParams p;
  p.q = q; p.k = k; p.v = v; p.k_cache = k_cache; p.v_cache = v_cache; p.out = out;
  p.block_tables = block_tables; p.cu_q = cu_seqlens_q; p.cu_k = cu_seqlens_k;
  p.q_stride = q_stride_tok; p.k_stride = k_stride_tok; p.v_stride = v_stride_tok;
  p.table_stride = table_stride; p.num_blocks = num_blocks;
  p.n_seqs = n_seqs; p.max_seqlen_q = max_seqlen_q; p.max_seqlen_k = max_seqlen_k;
  p.n_heads = n_heads; p.n_kv_heads = n_kv_heads; p.head_dim = head_dim;
  p.block_size = block_size;
  p.table_width = paged ? (int)((max_seqlen_k + block_size - 1) / block_size) : 0;
  p.scale = scale;
  hipStream_t st = (hipStream_t)stream;
  // MV_PREFILL_SIMPLE routes 16-bit tensors through the vector-ALU kernel (debug aid)
  if (dtype != MV_F32 && !(flags & MV_PREFILL_SIMPLE)) {
    const int rc = mv_prefill_mfma_launch(p, dtype, flags >> 4, st);
    if (rc != MV_E_SHAPE) return rc;
  }
  return mv_prefill_simple_launch(p, (int64_t)n_seqs * max_seqlen_q, dtype, st);
}
