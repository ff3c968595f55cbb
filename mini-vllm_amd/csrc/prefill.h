// Shared parameter block of the prefill kernels (K2).
#pragma once
#include "common.h"

struct PrefillParams {
  const void *q, *k, *v, *k_cache, *v_cache;
  void *out;
  const int32_t *block_tables;  // NULL => contiguous k/v
  const int32_t *cu_q, *cu_k;
  int64_t q_stride, k_stride, v_stride, table_stride, num_blocks;
  int n_seqs, max_seqlen_q, max_seqlen_k;
  int n_heads, n_kv_heads, head_dim, block_size, table_width;
  float scale;
};

// generic vector-ALU path (every dtype, D in {64,128})
int mv_prefill_simple_launch(const PrefillParams &p, int64_t max_total_q, int dtype, hipStream_t stream);
// MFMA flash path (bf16/f16, D in {64,128}); returns MV_E_SHAPE when it does not apply
// variant: bench switches (flags >> 4 of mv_prefill_varlen), 0 = the shipped kernel
int mv_prefill_mfma_launch(const PrefillParams &p, int dtype, int variant, hipStream_t stream);
