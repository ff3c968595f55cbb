// K3 — paged-KV decode attention for gfx950
// (reference: minivllm/models/layers/page_attention.py:35-133 and the
//  flash_attn_with_kvcache call site, layers/attention.py:436-447).
//
// HBM-bound: every K/V byte of a (sequence, kv-head) pair is read exactly
// once, 16 bytes per lane, LPT = D/VEC adjacent lanes covering one token
// row (256 contiguous bytes for bf16 D=128), straight into VGPRs.  One
// workgroup (4 waves) owns one (sequence, kv-head[, head-chunk], split); all
// G query heads of the kv-head are served from the same K/V registers.
// Each LPT-lane group runs an independent online softmax over its tokens
// (no cross-lane traffic in the loop besides the in-row DPP dot-product
// reduction); groups, waves and splits are merged at the end through
// shuffles, LDS and (only when split) an fp32 workspace.
//
// WIDE (n_kv_heads a multiple of TPL = 64 / LPT): the TPL lane groups of a wave take TPL adjacent kv heads of
// ONE token instead of TPL tokens of one kv head, so a wave-wide load is 1 KiB of contiguous cache (a token
// row is n_kv_heads x D contiguous elements) instead of TPL pieces of 256 B at a row stride, a workgroup owns
// (sequence, TPL kv heads, split), and the token -> cache row arithmetic is scalar.  Measured as pure reads
// (tools/dbg/ingest_probe3.hip): 6.85 TB/s against 6.5 TB/s for the per-head mapping.
// (The same idea for the matrix-core kernel — a wave per kv head — was measured SLOWER, 94 vs 88 us on Qwen2-7B
//  heads at B = 256: that kernel is bound by the bytes its 8 resident waves per CU keep in flight, and a wave
//  that walks twice as many tiles serially keeps no more of them in flight.)
#include "paged_decode_body.h"
#include <cstdlib>


// ---- host side -------------------------------------------------------------
namespace {

struct DecodePlan {
  int g_chunk;  // q heads served per workgroup (template G)
  int n_chunks;
  int nload;
  int tile;
  int unit;
  int nsplit;
  bool uniform;
  bool mfma;  // tensor-core variant (paged_decode_mfma.hip)
  bool wide;  // vector kernel, TPL adjacent kv heads per workgroup (whole-row loads)
};

constexpr int kTargetWorkgroups = 2048;  // ~8 workgroups per CU
constexpr int kMaxSplits = 64;
constexpr int kMfmaMinGroup = 4;

int pick_g(int g_total) {
  if (g_total <= 8) return g_total;
  // widest chunk <= 8 that divides evenly, else 8 with a ragged tail
  for (int g = 8; g >= 4; --g)
    if (g_total % g == 0) return g;
  return 8;
}

bool make_plan(int batch, int n_heads, int n_kv_heads, int head_dim, int block_size,
               int max_blocks, int dtype, int flags, DecodePlan *pl) {
  if (n_kv_heads <= 0 || n_heads % n_kv_heads) return false;
  if (head_dim != 64 && head_dim != 128) return false;
  const int g_total = n_heads / n_kv_heads;
  int kv_groups = n_kv_heads;  // workgroups along the kv-head axis
  pl->wide = false;
  // >= 4 query heads per kv head: QK^T and PV go to the matrix cores (the vector path spends
  // ~G x 20 VALU instructions per key and drops to 2-3.7 TB/s at G = 7..8)
  const int min_group = (flags & MV_DECODE_VECTOR) ? 1 << 30 : ((flags & MV_DECODE_MFMA) ? 1 : kMfmaMinGroup);
  pl->mfma = dtype != MV_F32 && head_dim == 128 && g_total >= min_group && block_size % 32 == 0;
  if (pl->mfma) {
    pl->g_chunk = g_total < 16 ? g_total : 16;
    pl->n_chunks = (g_total + 15) / 16;
    pl->nload = 8;
    pl->tile = 32;
    pl->unit = 128;
    pl->uniform = true;
  } else {
    pl->g_chunk = pick_g(g_total);
    pl->n_chunks = (g_total + pl->g_chunk - 1) / pl->g_chunk;
    const int vec = dtype == MV_F32 ? 4 : 8;
    const int tpl = 64 / (head_dim / vec);
    pl->nload = pl->g_chunk <= 4 ? 8 : 4;
    pl->wide = pl->n_chunks == 1 && pl->g_chunk <= 4 && n_kv_heads % tpl == 0 && !(flags & MV_DECODE_PER_HEAD);
    pl->tile = pl->wide ? pl->nload : pl->nload * tpl;
    pl->unit = 4 * pl->tile;
    pl->uniform = block_size % pl->tile == 0;
    if (pl->wide) kv_groups = n_kv_heads / tpl;
  }
  const int64_t max_ctx = (int64_t)max_blocks * block_size;
  const int max_units = (int)((max_ctx + pl->unit - 1) / pl->unit);
  const int64_t wg = (int64_t)batch * kv_groups * pl->n_chunks;
  // the matrix-core kernel keeps 4 workgroups per CU resident (LDS): one wave of 1024 is enough
  // whole-row loads: HBM likes FEWER concurrent streams (B = 256 x 1024 tokens: 162 us at 512 workgroups,
  // 170 us at 1024, 182 us at 2560; the per-head mapping: 171 us at its best)
  int target = pl->wide ? kTargetWorkgroups / 4 : (pl->mfma ? kTargetWorkgroups / 2 : kTargetWorkgroups);
  // per-head vector mapping with >= 512 workgroups already (e.g. the 4 q / 2 kv heads of a TP = 4 shard at B = 256): no
  // split — 45 us against 50 us with the 4-way split and its combine launch
  if (!pl->mfma && !pl->wide && wg >= 512) target = (int)(wg < (1 << 20) ? wg : (1 << 20));
  if ((flags >> 8) & 0xff) target = ((flags >> 8) & 0xff) * 16;  // tuning aid: bits 8-15 = target workgroups / 16
  int ns = wg > 0 ? (int)((target + wg - 1) / wg) : 1;
  if (ns > max_units) ns = max_units;
  if (ns > kMaxSplits) ns = kMaxSplits;
  if (ns < 1) ns = 1;
  pl->nsplit = ns;
  return true;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

template <typename Tag, int D, int G, int NLOAD, bool ROPE>
void launch_gr(const DecodeParams &p, const DecodePlan &pl, int batch, hipStream_t stream) {
  dim3 block(256);
  if constexpr (G <= 4) {
    if (pl.wide) {
      dim3 grid(p.n_kv_heads / (64 / (D / elem<Tag>::VEC)), pl.nsplit, batch);
      if (pl.uniform)
        hipLaunchKernelGGL((paged_decode_kernel<Tag, D, G, NLOAD, true, true, ROPE>), grid, block, 0, stream, p);
      else
        hipLaunchKernelGGL((paged_decode_kernel<Tag, D, G, NLOAD, false, true, ROPE>), grid, block, 0, stream, p);
      return;
    }
  }
  dim3 grid(p.n_kv_heads * pl.n_chunks, pl.nsplit, batch);
  if (pl.uniform)
    hipLaunchKernelGGL((paged_decode_kernel<Tag, D, G, NLOAD, true, false, ROPE>), grid, block, 0, stream, p);
  else
    hipLaunchKernelGGL((paged_decode_kernel<Tag, D, G, NLOAD, false, false, ROPE>), grid, block, 0, stream, p);
}

template <typename Tag, int D, int G, int NLOAD>
void launch_g(const DecodeParams &p, const DecodePlan &pl, int batch, hipStream_t stream) {
  if constexpr (elem<Tag>::VEC == 8 && G <= 4) {   // (rope fusion: 16-bit dtypes, up to 4 query heads per kv head)
    if (p.positions != nullptr) return launch_gr<Tag, D, G, NLOAD, true>(p, pl, batch, stream);
  }
  launch_gr<Tag, D, G, NLOAD, false>(p, pl, batch, stream);
}

template <typename Tag, int D>
int launch_d(const DecodeParams &p, const DecodePlan &pl, int batch, int dtype, int flags, hipStream_t stream) {
  if (pl.mfma) {
    const int klds = (flags & MV_DECODE_KLDS_ON) ? 1 : ((flags & MV_DECODE_KLDS_OFF) ? 0 : -1);
    const int rc = mv_paged_decode_mfma_launch(p, pl.n_chunks, pl.nsplit, batch, dtype, klds, stream);
    if (rc != MV_OK) return rc;
    if (pl.nsplit > 1) {
      hipLaunchKernelGGL((paged_decode_reduce_kernel<Tag, D>), dim3(batch * p.n_heads), dim3(128), 0,
                         stream, p.ws_acc, p.ws_ml, (typename elem<Tag>::T *)p.out, pl.nsplit);
      MV_LAUNCH_CHECK();
    }
    return MV_OK;
  }
  switch (pl.g_chunk) {
    case 1: launch_g<Tag, D, 1, 8>(p, pl, batch, stream); break;
    case 2: launch_g<Tag, D, 2, 8>(p, pl, batch, stream); break;
    case 3: launch_g<Tag, D, 3, 8>(p, pl, batch, stream); break;
    case 4: launch_g<Tag, D, 4, 8>(p, pl, batch, stream); break;
    case 5: launch_g<Tag, D, 5, 4>(p, pl, batch, stream); break;
    case 6: launch_g<Tag, D, 6, 4>(p, pl, batch, stream); break;
    case 7: launch_g<Tag, D, 7, 4>(p, pl, batch, stream); break;
    case 8: launch_g<Tag, D, 8, 4>(p, pl, batch, stream); break;
    default: return MV_E_SHAPE;
  }
  MV_LAUNCH_CHECK();
  if (pl.nsplit > 1) {
    hipLaunchKernelGGL((paged_decode_reduce_kernel<Tag, D>), dim3(batch * p.n_heads), dim3(128), 0,
                       stream, p.ws_acc, p.ws_ml, (typename elem<Tag>::T *)p.out, pl.nsplit);
    MV_LAUNCH_CHECK();
  }
  return MV_OK;
}

}  // namespace

// the launch plan, for decode_persistent.hip (its attention phase runs the same virtual grid): 0 = ok
int mv_paged_decode_plan(int batch, int n_heads, int n_kv_heads, int head_dim, int block_size, int max_blocks, int dtype,
                         int flags, int *g_chunk, int *n_chunks, int *nsplit, int *wide, int *mfma, int *uniform) {
  DecodePlan pl;
  if (!make_plan(batch, n_heads, n_kv_heads, head_dim, block_size, max_blocks, dtype, flags, &pl)) return MV_E_SHAPE;
  *g_chunk = pl.g_chunk, *n_chunks = pl.n_chunks, *nsplit = pl.nsplit, *wide = pl.wide, *mfma = pl.mfma, *uniform = pl.uniform;
  return MV_OK;
}

extern "C" size_t mv_paged_decode_workspace_bytes(int batch, int n_heads, int n_kv_heads,
                                                  int head_dim, int block_size, int max_blocks, int flags) {
  // sized for the widest element type so one buffer serves every dtype
  DecodePlan a, b;
  if (batch <= 0 || !make_plan(batch, n_heads, n_kv_heads, head_dim, block_size, max_blocks, MV_BF16, flags, &a) ||
      !make_plan(batch, n_heads, n_kv_heads, head_dim, block_size, max_blocks, MV_F32, flags, &b))
    return 256;
  const int ns = a.nsplit > b.nsplit ? a.nsplit : b.nsplit;
  if (ns <= 1) return 256;
  const size_t slots = (size_t)batch * n_heads * ns;
  return align256(slots * head_dim * sizeof(float)) + align256(slots * 2 * sizeof(float));
}

namespace {
struct RopeArgs {
  const void *k_new, *v_new;
  int64_t k_stride, v_stride;
  const int64_t *positions;
  const float *cos, *sin;
  int64_t max_pos;
  const void *q_norm_w, *k_norm_w;
  float eps;
  const int32_t *slot_mapping;
};
int paged_decode_impl(const void *q, int64_t q_stride_tok, const void *k_cache,
                      const void *v_cache, const int32_t *block_tables,
                      int64_t table_stride, const int32_t *context_lens, void *out,
                      int batch, int n_heads, int n_kv_heads, int head_dim,
                      int block_size, int max_blocks, int64_t num_blocks, float scale,
                      void *workspace, size_t ws_bytes, int dtype, int flags, void *stream, const RopeArgs *rope);
}  // namespace

extern "C" int mv_paged_decode(const void *q, int64_t q_stride_tok, const void *k_cache,
                               const void *v_cache, const int32_t *block_tables,
                               int64_t table_stride, const int32_t *context_lens, void *out,
                               int batch, int n_heads, int n_kv_heads, int head_dim,
                               int block_size, int max_blocks, int64_t num_blocks, float scale,
                               void *workspace, size_t ws_bytes, int dtype, int flags, void *stream) {
  return paged_decode_impl(q, q_stride_tok, k_cache, v_cache, block_tables, table_stride, context_lens, out, batch,
                           n_heads, n_kv_heads, head_dim, block_size, max_blocks, num_blocks, scale, workspace, ws_bytes,
                           dtype, flags, stream, nullptr);
}

// K7 + K6 (per head) + K1 + K3 in ONE launch for decode steps: q / k / v are the RAW rows of the qkv projection; the
// attention kernel norms and rotates q itself, and the workgroup that reads a sequence's last token first writes that
// token's K (normed, rotated) and V into the cache (see paged_decode_kernel<ROPE>).  Bit-identical to
// mv_rope_qk_store followed by mv_paged_decode.  MV_E_SHAPE for layouts the fused form does not take (fp32; more than
// 16 query heads per kv head): the caller then issues the two launches.
extern "C" int mv_rope_paged_decode(const void *q, int64_t q_stride_tok, const void *k, int64_t k_stride_tok,
                                    const void *v, int64_t v_stride_tok, const int64_t *positions, const float *cos,
                                    const float *sin, int64_t max_pos, const void *q_norm_w, const void *k_norm_w,
                                    float eps, const int32_t *slot_mapping, void *k_cache, void *v_cache,
                                    const int32_t *block_tables, int64_t table_stride, const int32_t *context_lens,
                                    void *out, int batch, int n_heads, int n_kv_heads, int head_dim, int block_size,
                                    int max_blocks, int64_t num_blocks, float scale, void *workspace, size_t ws_bytes,
                                    int dtype, int flags, void *stream) {
  if (batch == 0) return MV_OK;
  if (!k || !v || !positions || !cos || !sin || !slot_mapping) return MV_E_NULLPTR;
  if (dtype != MV_BF16 && dtype != MV_F16) return dtype == MV_F32 ? MV_E_SHAPE : MV_E_DTYPE;
  if (max_blocks == 0 || num_blocks == 0 || n_kv_heads <= 0 || n_heads % n_kv_heads) return MV_E_SHAPE;
  if (!mv_aligned16(k) || !mv_aligned16(v) || (k_stride_tok & 7) || (v_stride_tok & 7) ||
      (q_norm_w && !mv_aligned16(q_norm_w)) || (k_norm_w && !mv_aligned16(k_norm_w)))
    return MV_E_ALIGN;
  DecodePlan pl;
  if (!make_plan(batch, n_heads, n_kv_heads, head_dim, block_size, max_blocks, dtype, flags, &pl)) return MV_E_SHAPE;
  // vector kernel: up to 4 query heads per kv head; matrix-core kernel: one chunk of <= 16 heads
  if (pl.n_chunks != 1) return MV_E_SHAPE;
  if (pl.mfma ? head_dim != 128 : pl.g_chunk > 4) return MV_E_SHAPE;
  const RopeArgs rope{k, v, k_stride_tok, v_stride_tok, positions, cos, sin, max_pos, q_norm_w, k_norm_w, eps, slot_mapping};
  return paged_decode_impl(q, q_stride_tok, k_cache, v_cache, block_tables, table_stride, context_lens, out, batch,
                           n_heads, n_kv_heads, head_dim, block_size, max_blocks, num_blocks, scale, workspace, ws_bytes,
                           dtype, flags, stream, &rope);
}

namespace {
int paged_decode_impl(const void *q, int64_t q_stride_tok, const void *k_cache,
                      const void *v_cache, const int32_t *block_tables,
                      int64_t table_stride, const int32_t *context_lens, void *out,
                      int batch, int n_heads, int n_kv_heads, int head_dim,
                      int block_size, int max_blocks, int64_t num_blocks, float scale,
                      void *workspace, size_t ws_bytes, int dtype, int flags, void *stream, const RopeArgs *rope) {
  if (batch == 0) return MV_OK;
  if (!q || !out || !context_lens) return MV_E_NULLPTR;
  if (batch < 0 || n_heads <= 0 || block_size <= 0 || max_blocks < 0 || num_blocks < 0)
    return MV_E_SHAPE;
  if (dtype != MV_BF16 && dtype != MV_F16 && dtype != MV_F32) return MV_E_DTYPE;
  const int esz = dtype == MV_F32 ? 4 : 2;
  if (max_blocks == 0 || num_blocks == 0) {
    // reference: zero-column block tables (warm-up) give zeros, page_attention.py:62-65
    return mv_check_hip(hipMemsetAsync(out, 0, (size_t)batch * n_heads * head_dim * esz,
                                       (hipStream_t)stream));
  }
  if (!k_cache || !v_cache || !block_tables) return MV_E_NULLPTR;
  DecodePlan pl;
  if (!make_plan(batch, n_heads, n_kv_heads, head_dim, block_size, max_blocks, dtype, flags, &pl))
    return MV_E_SHAPE;
  if (!mv_aligned16(q) || !mv_aligned16(k_cache) || !mv_aligned16(v_cache) || !mv_aligned16(out) ||
      (q_stride_tok * esz) % 16)
    return MV_E_ALIGN;

  DecodeParams p;
  p.q = q;
  p.k_cache = k_cache;
  p.v_cache = v_cache;
  p.block_tables = block_tables;
  p.context_lens = context_lens;
  p.out = out;
  p.q_stride = q_stride_tok;
  p.table_stride = table_stride;
  p.num_blocks = num_blocks;
  p.n_heads = n_heads;
  p.n_kv_heads = n_kv_heads;
  p.g_total = n_heads / n_kv_heads;
  p.block_size = block_size;
  p.bs_shift = (block_size & (block_size - 1)) == 0 ? __builtin_ctz((unsigned)block_size) : -1;
  p.max_blocks = max_blocks;
  p.nsplit = pl.nsplit;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.ws_acc = nullptr;
  p.ws_ml = nullptr;
  if (pl.nsplit > 1) {
    const size_t slots = (size_t)batch * n_heads * pl.nsplit;
    const size_t acc_bytes = align256(slots * head_dim * sizeof(float));
    const size_t need = acc_bytes + align256(slots * 2 * sizeof(float));
    if (!workspace) return MV_E_NULLPTR;
    if (ws_bytes < need || !mv_aligned16(workspace)) return MV_E_WORKSPACE;
    p.ws_acc = (float *)workspace;
    p.ws_ml = (float *)((char *)workspace + acc_bytes);
  }
  if (rope) {
    p.k_new = rope->k_new, p.v_new = rope->v_new, p.k_stride = rope->k_stride, p.v_stride = rope->v_stride;
    p.positions = rope->positions, p.cos = rope->cos, p.sin = rope->sin, p.max_pos = rope->max_pos;
    p.q_norm_w = rope->q_norm_w, p.k_norm_w = rope->k_norm_w, p.eps = rope->eps, p.slot_mapping = rope->slot_mapping;
  }
  hipStream_t st = (hipStream_t)stream;
  if (head_dim == 128) {
    MV_DISPATCH_DTYPE(dtype, return (launch_d<Tag, 128>(p, pl, batch, dtype, flags, st)));
  } else {
    MV_DISPATCH_DTYPE(dtype, return (launch_d<Tag, 64>(p, pl, batch, dtype, flags, st)));
  }
  return MV_OK;
}
}  // namespace
