/*
 * mvhip.h — C-ABI of libmvhip.so, the MI355X (gfx950 / CDNA4) kernels behind
 * mini-vllm's paged-attention inference hot path.
 *
 * The reference (jianzhnie/mini-vllm) is pure Python and has no FFI; its
 * pluggable seam is the AttentionBackend ABC and the layer classes.  Every
 * entry point below names the reference interface it replaces (paths are
 * relative to the reference checkout).
 *
 * Conventions (all entry points):
 *   - return 0 on success, <0 for an argument error (MV_E_*), >0 = hipError_t
 *   - no allocation, no host<->device sync, no host reads of device memory,
 *     no global state and no environment variables: every launch goes to the
 *     caller's `stream` (a hipStream_t) and is hipGraph-capturable; tuning
 *     overrides are explicit `flags` / `shape` / `cfg` arguments (0 = default)
 *   - pointers are device pointers (tensor.data_ptr()); strides are in ELEMENTS
 *   - `dtype` selects the element type of activations, KV cache and weights
 *   - index tensors: slot_mapping / block_tables / context_lens / cu_seqlens
 *     are int32; positions, top_ks and sampled tokens are int64 (the dtypes
 *     the reference engine produces, engine/inference_executor.py:520-549,
 *     :149-180, :787-796)
 */
#ifndef MVHIP_H
#define MVHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MV_ABI_VERSION 1

typedef enum { MV_BF16 = 0, MV_F16 = 1, MV_F32 = 2 } mv_dtype;

enum {
  MV_OK = 0,
  MV_E_NULLPTR = -1,   /* a required pointer is NULL                       */
  MV_E_SHAPE = -2,     /* unsupported / inconsistent shape                 */
  MV_E_DTYPE = -3,     /* unknown dtype                                    */
  MV_E_WORKSPACE = -4, /* workspace too small                              */
  MV_E_ALIGN = -5,     /* pointer or stride not aligned for 16-byte access */
  MV_E_RANGE = -6,     /* scalar argument out of range                     */
  MV_E_ALIAS = -7      /* an output overlaps an input other workgroups still read */
};

int mv_version(void);
const char *mv_strerror(int code);

/* K1 — scatter new K/V rows into the paged cache.
 * Replaces StandardAttentionBackend.store_kv_cache and the Triton
 * store_kvcache_kernel (minivllm/models/layers/attention_backend.py:50-78,
 * :196-266).  k,v: [n_tok, n_kv_heads, head_dim] with token stride
 * k_stride_tok / v_stride_tok (views of the fused qkv GEMM output);
 * caches: contiguous [num_slots, n_kv_heads*head_dim] (= [nb, bs, Hkv, D]).
 * Rows with slot < 0 are skipped (reference :70-71); rows with
 * slot >= num_slots are skipped too (the reference would fault). */
int mv_store_kvcache(const void *k, const void *v, int64_t k_stride_tok,
                     int64_t v_stride_tok, void *k_cache, void *v_cache,
                     const int32_t *slot_mapping, int n_tok, int n_kv_heads,
                     int head_dim, int64_t num_slots, int dtype, void *stream);

/* K6 — RMSNorm.  Replaces RMSNorm.rms_forward
 * (minivllm/models/layers/layernorm.py:81-93): fp32 math,
 * y = ((x * rsqrt(mean(x^2)+eps)) * w) -> dtype. */
int mv_rmsnorm(const void *x, int64_t x_stride, const void *w, void *out,
               int64_t out_stride, int rows, int cols, float eps, int dtype,
               void *stream);

/* K6 — fused residual-add + RMSNorm.  Replaces RMSNorm.add_rms_forward
 * (layernorm.py:95-105): s = f32(x)+f32(residual); residual <- s -> dtype
 * (in place); out = norm(s) from the UNROUNDED sum. */
int mv_add_rmsnorm(const void *x, int64_t x_stride, void *residual,
                   int64_t res_stride, const void *w, void *out,
                   int64_t out_stride, int rows, int cols, float eps,
                   int dtype, void *stream);

/* K7 (+K6 per-head, +K1) — q/k head RMSNorm (optional) -> NeoX RoPE -> K/V
 * cache store, one launch.  Replaces q_norm/k_norm + RotaryEmbedding.forward
 * + store_kv_cache on the flat path (minivllm/models/qwen_base.py:173-190,
 * layers/rotary_embedding.py:77-81,166-193, attention_backend.py:196-266).
 * q: [n_tok,H,D] stride q_stride_tok; k,v: [n_tok,Hkv,D]; positions int64;
 * cos,sin: fp32 [max_pos, D/2].  q_norm_w/k_norm_w (dtype, [D]) may both be
 * NULL (Qwen2).  q_out [n_tok,H,D] contiguous is required; k_out
 * [n_tok,Hkv,D] contiguous may be NULL (decode); k_cache/v_cache/slot_mapping
 * may all be NULL (warm-up prefill has no slots, inference_executor.py:373-410).
 * Rounding points follow the reference: norm -> dtype, then rope in fp32 ->
 * dtype.  positions outside [0,max_pos) give MV-defined zeros rotation
 * (cos=1,sin=0) instead of the reference's index error. */
int mv_rope_qk_store(const void *q, const void *k, const void *v,
                     int64_t q_stride_tok, int64_t k_stride_tok,
                     int64_t v_stride_tok, const int64_t *positions,
                     const float *cos, const float *sin, int64_t max_pos,
                     const void *q_norm_w, const void *k_norm_w, float eps,
                     void *q_out, void *k_out, void *k_cache, void *v_cache,
                     const int32_t *slot_mapping, int64_t num_slots, int n_tok,
                     int n_heads, int n_kv_heads, int head_dim, int dtype,
                     void *stream);

/* K3 — paged-KV decode attention (one query token per sequence).
 * Replaces PageAttention.__call__ / flash_attn_with_kvcache
 * (minivllm/models/layers/page_attention.py:35-133, attention.py:436-447).
 * q: [B,H,D] token stride q_stride_tok; caches [num_blocks,block_size,Hkv,D];
 * block_tables int32 [B, >=max_blocks] row stride table_stride, -1 padded;
 * context_lens int32 [B]; out [B,H,D] contiguous.  Rows with
 * context_len <= 0 or max_blocks == 0 produce zeros
 * (page_attention.py:62-65).  Block ids outside [0,num_blocks) inside the
 * context contribute nothing (reference: zero K/V rows).
 * workspace: mv_paged_decode_workspace_bytes() bytes, 16-byte aligned. */
/* `flags` (same value for the size query and the call), 0 = kernel variant chosen from the shape: */
enum {
  MV_DECODE_VECTOR = 1,   /* never the matrix-core kernel                              */
  MV_DECODE_MFMA = 2,     /* the matrix-core kernel wherever it takes the shape        */
  MV_DECODE_KLDS_OFF = 4, /* matrix-core kernel: K straight to registers               */
  MV_DECODE_KLDS_ON = 8,  /* matrix-core kernel: K through LDS                         */
  MV_DECODE_PER_HEAD = 16 /* vector kernel: one kv head per workgroup (never whole-row loads)  */
  /* bits 8-15, when non-zero: workgroups to aim for when splitting the context, in units of 16 (tuning aid) */
};
size_t mv_paged_decode_workspace_bytes(int batch, int n_heads, int n_kv_heads,
                                       int head_dim, int block_size,
                                       int max_blocks, int flags);
int mv_paged_decode(const void *q, int64_t q_stride_tok, const void *k_cache,
                    const void *v_cache, const int32_t *block_tables,
                    int64_t table_stride, const int32_t *context_lens,
                    void *out, int batch, int n_heads, int n_kv_heads,
                    int head_dim, int block_size, int max_blocks,
                    int64_t num_blocks, float scale, void *workspace,
                    size_t ws_bytes, int dtype, int flags, void *stream);

/* K7 + K6 (per head) + K1 + K3 in one launch, for decode steps: q / k / v are the RAW rows of the qkv projection
 * (views of its output: row strides in elements); the attention kernel applies the per-head RMSNorm (q_norm_w /
 * k_norm_w, NULL = none) and the rotary embedding to q itself, and the workgroup that reads a sequence's last token
 * first writes that token's K (normed, rotated) and V rows to cache slot slot_mapping[b] (-1 = no store).  Replaces
 * mv_rope_qk_store + mv_paged_decode for the flat decode path of QwenAttention.forward
 * (minivllm/models/qwen_base.py:173-190 with layers/attention.py:436-447) with bit-identical results.  Arguments as in
 * those two calls.  MV_E_SHAPE for layouts the fused form does not take (fp32; more than 16 query heads per kv
 * head): the caller then issues the two launches. */
int mv_rope_paged_decode(const void *q, int64_t q_stride_tok, const void *k, int64_t k_stride_tok,
                         const void *v, int64_t v_stride_tok, const int64_t *positions, const float *cos,
                         const float *sin, int64_t max_pos, const void *q_norm_w, const void *k_norm_w,
                         float eps, const int32_t *slot_mapping, void *k_cache, void *v_cache,
                         const int32_t *block_tables, int64_t table_stride, const int32_t *context_lens,
                         void *out, int batch, int n_heads, int n_kv_heads, int head_dim, int block_size,
                         int max_blocks, int64_t num_blocks, float scale, void *workspace, size_t ws_bytes,
                         int dtype, int flags, void *stream);


/* K2 — packed varlen causal prefill attention (flash-style, MFMA).
 * Replaces flash_attn_varlen_func / Attention._fallback_attention
 * (minivllm/models/layers/attention.py:422-433, :467-504,554-597).
 * q: [T,H,D] stride q_stride_tok, out [T,H,D] contiguous.
 * K/V source, exactly one of:
 *   contiguous: k,v [Tk,Hkv,D] with strides (block_tables == NULL)
 *   paged:      k_cache/v_cache + block_tables (prefix-cache hit; keys of
 *               sequence i are its first seqlen_k tokens in the cache)
 * Causal mask is bottom-right aligned: query j of sequence i (0-based in its
 * seqlen_q) attends keys 0 .. seqlen_k-seqlen_q+j (attention.py:584-588).
 * flags: MV_PREFILL_SIMPLE routes 16-bit tensors through the vector-ALU kernel (debug aid); 0 = default. */
enum { MV_PREFILL_SIMPLE = 1 };
int mv_prefill_varlen(const void *q, int64_t q_stride_tok, const void *k,
                      const void *v, int64_t k_stride_tok, int64_t v_stride_tok,
                      const void *k_cache, const void *v_cache,
                      const int32_t *block_tables, int64_t table_stride,
                      int64_t num_blocks, void *out,
                      const int32_t *cu_seqlens_q, const int32_t *cu_seqlens_k,
                      int n_seqs, int max_seqlen_q, int max_seqlen_k,
                      int n_heads, int n_kv_heads, int head_dim,
                      int block_size, float scale, int dtype, int flags, void *stream);

/* K8 — fused temperature / top-k / top-p / min-p / multinomial sampler.
 * Replaces Sampler.forward + sampling/functional.py:34-233,410-437 as driven
 * by InferenceExecutor._sample_tokens (inference_executor.py:776-806).
 * logits [B,V] row stride ld; temperatures/top_ps/min_ps fp32 [B];
 * top_ks int64 [B] (<=0 or >=V disables); out_tokens int64 [B].
 * temperature == 0 -> argmax (lowest index on ties).
 * RNG: Philox4x32-10 keyed by (seed, offset + *offset_dev, row); offset_dev
 * (device u64, may be NULL) lets a captured graph draw fresh numbers.
 * keep_mask (uint8 [B,V], may be NULL) receives the post-filter support set
 * (1 = kept), for parity tests against the reference's filtered logits.
 * workspace (may be NULL): when ws_bytes >= 8 * batch it receives, per row, the fp32 pair {row maximum, kept
 * mass = sum over the kept tokens of exp((logit - maximum) / temperature)} (greedy rows: {maximum, 1}) — what
 * mv_mass_pack sends instead of the logits. */
size_t mv_sample_workspace_bytes(int batch, int vocab);
int mv_sample(const void *logits, int64_t ld, const float *temperatures,
              const int64_t *top_ks, const float *top_ps, const float *min_ps,
              uint64_t seed, uint64_t offset, const uint64_t *offset_dev,
              int64_t *out_tokens, uint8_t *keep_mask, int batch, int vocab,
              void *workspace, size_t ws_bytes, int dtype, void *stream);

/* a2 (device side) — advance the decode inputs of a batch by one step WITHOUT the host:
 * ids <- tokens just sampled, positions += 1, context_lens += 1 (rows with context 0 are padding and
 * stay 0), slot_mapping += 1 (rows with slot -1 stay -1), *rng_offset += 1.  Appended to the captured
 * decode graph so that the NEXT step can be replayed speculatively before the host has read this step's
 * tokens; the host rewrites all of these whenever the batch changes or a sequence enters a new block.
 * Replaces, for consecutive steps of an unchanged batch, the per-step refresh of
 * InferenceExecutor._prepare_decode_input (minivllm/engine/inference_executor.py:551-617). */
int mv_decode_advance(int64_t *input_ids, const int64_t *tokens, int64_t *positions,
                      int32_t *slot_mapping, int32_t *context_lens, int64_t *rng_offset,
                      int batch, void *stream);

/* K12 — every decoder layer of a SMALL decode step (<= 16 rows) in ONE persistent launch: per layer, between grid
 * barriers, residual add + RMSNorm + qkv projection | head norms + RoPE + K/V store + paged attention | split-KV combine |
 * o_proj | residual add + RMSNorm + gate_up + SiluAndMul | down_proj — the op order of QwenDecoderLayer.forward /
 * QwenAttention.forward / QwenMLP.forward (minivllm/models/qwen_base.py:281-298, :173-190, :232-236), replacing the six
 * launches per layer of the multi-launch path bit for bit (same K slices, same summation order, same rounding points).
 * The next projection's weights stream into LDS while a workgroup waits at a barrier.  One workgroup per CU; all must be
 * resident (nothing else may hold CUs for long on the device while it runs).
 *   layers_dev: DEVICE array of n_layers MvPersistLayer (pointers to the packed weights / norm weights / KV planes);
 *   a:          HOST struct (copied into the launch); x_in [m, hidden] = the embedding output, res_in = NULL;
 *   workspace:  mv_decode_layers_persistent_workspace_bytes(...) bytes, 256-byte aligned; on return (stream order) the last
 *               layer's down_proj output is at workspace + out_h_offset and the residual at workspace + out_res_offset
 *               ([m, hidden] contiguous): the caller runs the final norm + LM head on them;
 *   status_dev: one uint32: != 0 after the launch = a bounded barrier wait gave up, the results are invalid;
 *   flags:      bits 0-7 weight chunks (8 KiB) per streaming wave kept in flight across a barrier (0 = default 2, 0xff = none).
 * MV_E_SHAPE for what this form does not take (more than 16 rows, head_dim != 128, more than 4 query heads per kv head,
 * wide output tiles, TP shards ...): the caller then runs the multi-launch path. */
typedef struct MvPersistLayer {
  const void *ln1_w, *wqkv, *bqkv, *q_norm_w, *k_norm_w, *wo, *ln2_w, *wgu, *wdown;
  void *k_cache, *v_cache;
} MvPersistLayer;
typedef struct MvPersistArgs {
  int m, hidden, n_heads, n_kv_heads, head_dim, inter;
  float eps, scale;
  const void *x_in, *res_in;
  int64_t x_in_stride, res_in_stride;
  const int64_t *positions;
  const int32_t *slot_mapping, *block_tables, *context_lens;
  int64_t table_stride, num_blocks, max_pos;
  int block_size, max_blocks;
  const float *cos, *sin;
  int64_t out_h_offset, out_res_offset;   /* filled in by the call */
} MvPersistArgs;
size_t mv_decode_layers_persistent_workspace_bytes(int m, int hidden, int n_heads, int n_kv_heads, int head_dim, int inter,
                                                   int block_size, int max_blocks);
int mv_decode_layers_persistent(const MvPersistLayer *layers_dev, int n_layers, MvPersistArgs *a, void *workspace,
                                size_t ws_bytes, unsigned *status_dev, int dtype, int flags, void *stream);
/* sizeof(MvPersistLayer) (which = 0) / sizeof(MvPersistArgs) (which = 1) as this library was built: a binding checks its own
 * struct declarations against them (host only) */
size_t mv_persist_struct_bytes(int which);

/* K9 — SiluAndMul: out[r, c] = silu(x[r, c]) * x[r, cols + c].
 * Replaces SiluAndMul.forward (minivllm/models/layers/activation.py:70-96). */
int mv_silu_and_mul(const void *x, int64_t x_stride, void *out,
                    int64_t out_stride, int rows, int cols, int dtype,
                    void *stream);

/* K10 — decode-sized linear layer: out[m, :] = x[m, :] . W^T (+ bias), 1 <= M <= 64 rows, 16-bit
 * dtypes, fp32 accumulation, one rounding at the end.  Replaces the F.linear calls of the decode step
 * (minivllm/models/layers/linear.py:130,163,316; layers/embed_head.py:217) while the batch is small
 * enough that the projection is a pure HBM stream of W; larger M stays on hipBLASLt.
 * x [M,K] row stride x_stride; w [N,K] contiguous (torch Linear layout); bias [N] or NULL;
 * out [M,N] row stride out_stride.  K % 32 == 0, N % 4 == 0; MV_E_SHAPE otherwise (and for fp32 /
 * M > 64) so that the caller can fall back to the library GEMM. */
/* shape: launch-shape override for tools/bench_linear.py --sweep (bits 0-3 row tiles per wave, 4-8 waves,
 * 9-10 x through LDS 1 on / 2 off, 11-12 W through LDS 1 on / 2 off); 0 = chosen from the shape. */
int mv_linear_small_m(const void *x, int64_t x_stride, const void *w,
                      const void *bias, void *out, int64_t out_stride, int m,
                      int n, int k, int dtype, int shape, void *stream);

/* K10 + K9 — decode-sized gate_up projection with SiluAndMul in its epilogue:
 * out[m, j] = silu(x[m,:] . Wg[j,:]) * (x[m,:] . Wu[j,:]),  w_gate_up = [Wg ; Wu] of shape [2*n_inter, K]
 * (the merged layout of MergedColumnParallelLinear, minivllm/models/layers/linear.py:166-231), with the
 * rounding points of the unfused pair gate_up_proj -> act_fn (minivllm/models/qwen_base.py:232-234):
 * both GEMM outputs, silu and the product are each rounded to dtype.  Same limits as mv_linear_small_m. */
int mv_gate_up_silu_small_m(const void *x, int64_t x_stride, const void *w_gate_up,
                            void *out, int64_t out_stride, int m, int n_inter,
                            int k, int dtype, int shape, void *stream);

/* K10 on the fragment-ordered weight copy of mv_pack_weight (below): the same two operations, the same summation
 * order and rounding points (results bit-identical to the row-major entry points); only the weight address stream
 * differs — 1 KiB contiguous per wave-wide load instead of 16 pieces of 64 B.  The gated form needs n_inter % 16 == 0. */
int mv_linear_small_m_packed(const void *x, int64_t x_stride, const void *w_packed,
                             const void *bias, void *out, int64_t out_stride, int m,
                             int n, int k, int dtype, int shape, void *stream);
int mv_gate_up_silu_small_m_packed(const void *x, int64_t x_stride, const void *w_gate_up_packed,
                                   void *out, int64_t out_stride, int m, int n_inter,
                                   int k, int dtype, int shape, void *stream);

/* K6 + K10 — the RMSNorm in front of a projection folded into the projection's launch, for decode steps of <= 4
 * rows (K <= 2048; 2 rows to 4096, 1 row to 8192): every workgroup re-enacts the norm of those few rows in its
 * prologue (the arithmetic of mv_rmsnorm / mv_add_rmsnorm thread for thread), so the step loses two launches per
 * layer.  Replaces RMSNorm.forward followed by the F.linear of qkv_proj / gate_up_proj
 * (minivllm/models/qwen_base.py:266-279 with layers/layernorm.py:81-105 and layers/linear.py:130,163).
 *   residual == NULL: out = Linear(RMSNorm(x))
 *   else:             residual_out = x + residual (rounded to dtype), out = Linear(RMSNorm(x + residual));
 *                     residual_out must not alias residual (other workgroups still read the old rows).
 * Results are bit-identical to the two-launch sequence on the packed weight.  MV_E_SHAPE when the fused form does
 * not take the shape — the caller then issues the two launches. */
int mv_norm_linear_small_m_packed(const void *x, int64_t x_stride, const void *residual,
                                  int64_t res_stride, void *residual_out, int64_t res_out_stride,
                                  const void *norm_w, float eps, const void *w_packed, const void *bias,
                                  void *out, int64_t out_stride, int m, int n, int k, int dtype,
                                  int shape, void *stream);
int mv_norm_gate_up_silu_small_m_packed(const void *x, int64_t x_stride, const void *residual,
                                        int64_t res_stride, void *residual_out, int64_t res_out_stride,
                                        const void *norm_w, float eps, const void *w_gate_up_packed,
                                        void *out, int64_t out_stride, int m, int n_inter, int k,
                                        int dtype, int shape, void *stream);

/* K11 — decode-batch linear layer for 17..512 rows (the headline batch, max_num_seqs = 256): same contract as
 * mv_linear_small_m (fp32 accumulation, bias in fp32, one rounding) with K % 64 == 0, N % 4 == 0.  Replaces the
 * F.linear calls of minivllm/models/layers/linear.py:130,163,316 where the library GEMM is bound by how fast a CU
 * takes bytes in, not by HBM or the matrix cores (mini-vllm_amd/csrc/linear_mid_m.hip).
 * The weight comes RE-PACKED (once, at load time) by mv_pack_weight: [ceil(N/16)][K/32][64][8] elements, the
 * A-operand fragment order of v_mfma_f32_16x16x32 (lane l = row l & 15, k group l >> 4), rows beyond N zero;
 * mv_packed_weight_elems(n, k) elements.  A workgroup's weight share is then a set of linear HBM streams
 * (6.8 TB/s measured against 4.4 TB/s for 128-byte pieces of row-major rows).
 * `cfg` selects the workgroup tile (0 = chosen from the shape; 1.. = the table in linear_mid_m.hip, for
 * tools/bench_mid.py; +100 = the DMA ring alone, nothing computed or written; +200 = K loop rotated per workgroup). */
size_t mv_packed_weight_elems(int n, int k);
int mv_pack_weight(const void *w, void *w_packed, int n, int k, int dtype, void *stream);
int mv_linear_mid_m(const void *x, int64_t x_stride, const void *w_packed, const void *bias,
                    void *out, int64_t out_stride, int m, int n, int k, int dtype,
                    int cfg, void *stream);
/* K11 + K9 — merged gate_up projection ([gate rows ; up rows], packed as ONE [2*n_inter, K] matrix,
 * n_inter % 16 == 0) with SiluAndMul in the epilogue (see mv_gate_up_silu_small_m for the rounding points). */
int mv_gate_up_silu_mid_m(const void *x, int64_t x_stride, const void *w_gate_up_packed,
                          void *out, int64_t out_stride, int m, int n_inter, int k,
                          int dtype, int cfg, void *stream);
/* K11, K split over `splits` workgroup groups: out_partial [splits][m][n] fp32, unrounded, no bias; the
 * consumer kernel behind it sums the slices in split order (deterministic). */
int mv_linear_mid_m_partial(const void *x, int64_t x_stride, const void *w_packed,
                            float *out_partial, int m, int n, int k, int splits,
                            int dtype, int cfg, void *stream);

/* K6 behind a K-split projection: mv_add_rmsnorm whose x is the sum of `splits` fp32 partial results
 * [splits][rows][cols] of mv_linear_mid_m_partial, summed in split order and rounded to dtype first (the rounding
 * point of the GEMM output that RMSNorm.add_rms_forward reads, minivllm/models/layers/layernorm.py:95-105). */
int mv_add_rmsnorm_partial(const float *partial, int splits, void *residual, int64_t res_stride,
                           const void *w, void *out, int64_t out_stride, int rows, int cols,
                           float eps, int dtype, void *stream);

/* C1 — small-message all-reduce over xGMI peer-to-peer loads, for the [tokens, hidden] sums of a
 * tensor-parallel decode step.  Replaces dist.all_reduce in RowParallelLinear.forward and
 * VocabParallelEmbedding.forward (minivllm/models/layers/linear.py:316-318, layers/embed_head.py:170-173)
 * for 16-bit messages of at most max_bytes; larger ones (prefill) stay on RCCL.
 * Protocol (two-shot, one launch, bounded waits): mini-vllm_amd/csrc/allreduce.hip.
 * Set-up, on every rank (the host exchanges the handles, e.g. through the process group):
 *   mv_ar_alloc(max_bytes, &base)            one uncached region of mv_ar_region_bytes(max_bytes)
 *   mv_ar_ipc_export(base, handle)           mv_ar_ipc_handle_bytes() bytes, sent to every peer
 *   mv_ar_ipc_open(peer_handle, &peer_base)  for every peer
 * Per call: bases[world] = the region of every rank as mapped HERE (bases[rank] = own base).
 * mv_ar_all_reduce: out = sum over ranks of x (x == out allowed), n elements, n % (8*world) == 0, fp32
 * accumulation in rank order, identical bits on every rank.  These four are the only entry points of the
 * library that allocate or synchronise (set-up / tear-down / status only, never on the data path). */
size_t mv_ar_region_bytes(size_t max_bytes);
int mv_ar_max_world(void);
int mv_ar_alloc(size_t max_bytes, void **base_out);
int mv_ar_free(void *base);
int mv_ar_ipc_handle_bytes(void);
int mv_ar_ipc_export(void *base, void *handle_out);
int mv_ar_ipc_open(const void *handle, void **ptr_out);
int mv_ar_ipc_close(void *ptr);
/* Status block of the own region, 8 words: [0] number of peer waits that gave up so far (0 = every result is
 * valid)  [1] != 0 once the record below is filled  [2] phase | peer << 8 | block << 16 | kernel << 24 of the
 * FIRST expired wait (kernel 0 = mv_ar_all_reduce, 1 = the fused row kernel)  [3] the epoch (call number) it
 * waited for  [4] the flag value it last saw (an older epoch: the peer never reached this call)  [5] the wait
 * bound in ms (mv_ar_set_wait_limit_ms; default 20 000).  mv_ar_status synchronises the device;
 * mv_ar_status_async copies the block on `stream` into PINNED host memory (read it after that stream work is
 * known complete, e.g. together with a decode step's tokens). */
int mv_ar_status(void *base, size_t max_bytes, uint32_t *status_out);
int mv_ar_status_async(void *base, size_t max_bytes, uint32_t *pinned_host_out, void *stream);
int mv_ar_set_wait_limit_ms(void *base, size_t max_bytes, uint32_t ms);
int mv_ar_all_reduce(const void *x, void *out, int64_t n, void *const *bases,
                     int rank, int world, size_t max_bytes, int dtype, void *stream);

/* f1 — all-reduce + residual add + RMSNorm in one launch (same regions and protocol as mv_ar_all_reduce,
 * rows as the unit): x [m][k] = this rank's partial sums; afterwards
 *   residual <- (dtype(sum over ranks of x) + residual) -> dtype          (in place)
 *   out      <- ((s * rsqrt(mean(s^2)+eps)) * norm_w) -> dtype,  s the unrounded fp32 sum above
 * = dist.all_reduce (minivllm/models/layers/linear.py:316-318) + RMSNorm.add_rms_forward
 * (layers/layernorm.py:95-105), rounding points of the unfused pair.  norm_w == NULL: out <- the plain
 * row-wise all-reduce, residual untouched.  m % world == 0, k % 8 == 0, k <= 8192, m*k*2 <= max_bytes. */
int mv_ar_all_reduce_add_rmsnorm(const void *x, void *residual, const void *norm_w,
                                 float eps, void *out, int m, int k,
                                 void *const *bases, int rank, int world,
                                 size_t max_bytes, int dtype, void *stream);

/* f2 — greedy decoding under tensor parallelism without the logit all-gather (replaces, for all-greedy
 * batches, the gather of ParallelLMHead.forward, minivllm/models/layers/embed_head.py:219-229, + argmax).
 * mv_greedy_pack: buf [world][rows_padded][4] (16-bit dtype) <- zeros, except slot [rank][r] =
 *   (logits_local[r][local_tokens[r]], three 8-bit pieces of the GLOBAL index rank*vocab_per_rank + token),
 *   all exactly representable; local_tokens = shard-local argmax (mv_sample at temperature 0).
 * After an exact sum all-reduce of buf (mv_ar_all_reduce: x + 0 = x), mv_greedy_merge writes per row the
 * global index of the largest value, lowest index on ties.  world * vocab_per_rank <= 2^24. */
int mv_greedy_pack(const void *logits_local, int64_t ld, const int64_t *local_tokens,
                   int rank, int world, int64_t vocab_per_rank, int rows,
                   int rows_padded, void *buf, int dtype, void *stream);
int mv_greedy_merge(const void *buf, int world, int rows, int rows_padded,
                    int64_t *tokens, int dtype, void *stream);

/* f2 — SAMPLED decoding under tensor parallelism without the logit all-gather, for rows that are greedy or whose
 * top_k is in 1..mv_topc_width() (= 64; replaces, for such batches, the gather of ParallelLMHead.forward,
 * minivllm/models/layers/embed_head.py:219-229, in front of Sampler.forward, sampling/sampler.py:84-110).
 * mv_topc_pack: buf [world][rows_padded][64][4] (16-bit dtype) <- zeros, except slot [rank][r][c] = the c-th of the
 *   64 largest logits of this rank's vocab shard of row r (shard-visiting order, deterministic) as
 *   (logit, three 8-bit pieces of its GLOBAL index), all exact; shards shorter than 64 pad with (-inf, 0).
 * After an exact sum all-reduce of buf (mv_ar_all_reduce), mv_topc_unpack writes cand_logits [rows][world*64] and
 * cand_idx int32 [rows][world*64]; mv_sample on cand_logits draws a candidate position, cand_idx maps it back.
 * The union of the per-rank candidates holds a complete top-k set, and top-p / min-p / the draw only look at that. */
int mv_topc_width(void);
/* f2 for rows sampled from the WHOLE vocabulary (the default SamplingParams of minivllm/sampling_params.py:41-67:
 * top_k = -1, top_p = 1, min_p = 0) and greedy rows — an exact two-stage draw instead of the logit gather of
 * embed_head.py:219-229.  Every rank runs mv_sample on its own shard WITH the statistics workspace (local token drawn
 * with the row's uniform, shard maximum, shard mass); mv_mass_pack writes {maximum, mass as a binary exponent + three
 * 8-bit mantissa pieces, three 8-bit pieces of the token's GLOBAL index} — eight exact 16-bit values — into slot [rank]
 * of a zeroed buf [world][rows_padded][8]; after an exact sum all-reduce mv_mass_merge picks, per row, the rank r with
 * probability mass_r * exp((max_r - max) / T) / Z from a second uniform of the same Philox counter (seed, offset +
 * *offset_dev, row: pass the values given to mv_sample) and returns that rank's token: the softmax of the full row.
 * Temperature 0: the rank with the largest maximum, lowest rank on ties (= argmax, lowest index). */
int mv_mass_pack(const float *stats, const int64_t *local_tokens, int rank, int world,
                 int64_t vocab_per_rank, int rows, int rows_padded, void *buf, int dtype, void *stream);
int mv_mass_merge(const void *buf, const float *temperatures, uint64_t seed, uint64_t offset,
                  const uint64_t *offset_dev, int world, int rows, int rows_padded, int64_t *tokens,
                  int dtype, void *stream);
/* f2 for NUCLEUS (top_p < 1) and min-p rows with top_k off — the edge of the kept set found across the ranks instead of
 * the logit gather of embed_head.py:219-229 in front of Sampler.forward (sampling/sampler.py:84-110; top-p keeps a
 * token while the mass strictly above it is <= p * Z, min-p while prob >= min_p * max prob).  16-bit logits only.
 *   row_max [rows] fp32: the row's maximum over ALL ranks (MAX all-reduce of the shard maxima).
 *   mv_shard_key_hist: hist [rows][256] int64 <- this shard's mass per bin, mass = exp((x - max) / T) in 2^-40 fixed
 *     point; bins count the 16-bit order keys DOWN from key(max): prefix == NULL -> bin = high byte of the offset,
 *     else only elements whose high byte equals prefix[row], bin = low byte.  Integer sums: after a SUM all-reduce
 *     every rank holds the same histogram whatever the order of the additions.  Greedy / degenerate rows: zeros.
 *   mv_nucleus_pick on the SUMMED histogram: level 0 writes state [rows][2] = {budget = p * Z, mass above the chosen
 *     bin} and prefix [rows] (the last high-byte bin whose strictly-larger mass fits the budget); level 1 continues
 *     inside that bin and writes floor_out [rows] fp32 = the smallest kept VALUE (raised to max + T ln(min_p) when
 *     min-p asks for more; -inf for rows with neither filter and for greedy rows).  Ties at the edge are kept whole,
 *     like mv_sample's top-p.
 *   mv_mask_below: out[r][i] = logits[r][i] >= floor[r] ? logits[r][i] : -inf (NaN passes through).
 * Then mv_sample (filters off, statistics on) on the masked shard and mv_mass_pack / mv_mass_merge pick the rank: two
 * 2 KB-per-row integer all-reduces and 16 bytes per row and rank instead of 2 * V / tp bytes per row and rank. */
int mv_shard_key_hist(const void *logits_local, int64_t ld, int vocab_local, int rows, const float *row_max,
                      const float *temperatures, const int32_t *prefix, int64_t *hist, int dtype, void *stream);
int mv_nucleus_pick(const int64_t *hist, int level, const float *top_ps, const float *min_ps,
                    const float *temperatures, const float *row_max, int64_t *state, int32_t *prefix,
                    float *floor_out, int rows, int dtype, void *stream);
int mv_mask_below(const void *logits, int64_t ld, int vocab_local, int rows, const float *floor_row,
                  void *out, int64_t ld_out, int dtype, void *stream);
int mv_topc_pack(const void *logits_local, int64_t ld, int vocab_local, int rank, int world,
                 int64_t vocab_per_rank, int rows, int rows_padded, void *buf, int dtype, void *stream);
int mv_topc_unpack(const void *buf, int world, int rows, int rows_padded, void *cand_logits,
                   int32_t *cand_idx, int dtype, void *stream);
/* mv_topc_pick: tokens[r] = cand_idx[r][pos[r]] for sampled rows; for rows with temperatures[r] == 0 (greedy; NULL =
 * no greedy rows) the LOWEST global index among the candidates equal to the row maximum — the token TP = 1's argmax and
 * mv_greedy_merge name on ties (candidate order is shard-visiting order, not index order). */
int mv_topc_pick(const void *cand_logits, const int32_t *cand_idx, const int64_t *pos,
                 const float *temperatures, int64_t *tokens, int rows, int width, int dtype, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MVHIP_H */
