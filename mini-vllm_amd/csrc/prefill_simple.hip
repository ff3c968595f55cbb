// K2 (generic path) — packed varlen causal prefill attention on the vector
// ALU.  One wave per (query token, query head); keys are streamed exactly
// like the decode kernel does (LPT lanes per key row, online softmax per lane
// group).  This is the correctness path for fp32 tensors and for shapes the
// MFMA kernel (prefill_mfma.hip) does not cover; bf16/f16 with D in {64,128}
// go to the MFMA kernel.
// Reference: minivllm/models/layers/attention.py:467-504,554-597.
#include "common.h"
#include "prefill.h"


namespace {

template <typename Tag>
__device__ __forceinline__ float dot16(const u32x4 &a, const u32x4 &b) {
  float acc = 0.f;
  if constexpr (__is_same(Tag, bf16_tag)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc = __builtin_amdgcn_fdot2_f32_bf16(u2bf2(a[i]),
                                            u2bf2(b[i]), acc, false);
  } else if constexpr (__is_same(Tag, f16_tag)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc = __builtin_amdgcn_fdot2(u2h2(a[i]),
                                   u2h2(b[i]), acc, false);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc = fmaf(u2f(a[i]), u2f(b[i]), acc);
  }
  return acc;
}

template <typename Tag, int D>
__global__ void __launch_bounds__(256) prefill_simple_kernel(PrefillParams p) {
  typedef typename elem<Tag>::T T;
  constexpr int VEC = elem<Tag>::VEC;
  constexpr int LPT = D / VEC;
  constexpr int TPL = 64 / LPT;
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPT, grp = lane / LPT;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (token, head)
  const int64_t total_q = p.cu_q[p.n_seqs];
  if (item >= total_q * p.n_heads) return;
  const int tok = (int)(item / p.n_heads);
  const int h = (int)(item - (int64_t)tok * p.n_heads);
  const int hkv = h / (p.n_heads / p.n_kv_heads);
  // which sequence owns this token (n_seqs is small: linear scan)
  int seq = 0;
  while (seq + 1 < p.n_seqs && p.cu_q[seq + 1] <= tok) ++seq;
  const int q0 = p.cu_q[seq], sq = p.cu_q[seq + 1] - q0;
  const int k0 = p.cu_k[seq], sk = p.cu_k[seq + 1] - k0;
  const int nkeys = min(sk, (tok - q0) + (sk - sq) + 1);  // bottom-right aligned causal

  const T *qp = (const T *)p.q + (int64_t)tok * p.q_stride + (int64_t)h * D + sub * VEC;
  const u32x4 qraw = *reinterpret_cast<const u32x4 *>(qp);
  float m = -1e30f, l = 0.f, acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  const float scale_log2 = p.scale * 1.4426950408889634f;

  for (int base = 0; base < nkeys; base += TPL) {
    const int key = base + grp;
    bool ok = key < nkeys;
    const T *kp = nullptr, *vp = nullptr;
    if (p.block_tables) {
      int blk = -1;
      const int bi = key / p.block_size;
      if (ok && bi < p.table_width) blk = p.block_tables[(int64_t)seq * p.table_stride + bi];
      ok = ok && blk >= 0 && blk < p.num_blocks;
      const int64_t row = (int64_t)blk * p.block_size + (key - bi * p.block_size);
      kp = (const T *)p.k_cache + (row * p.n_kv_heads + hkv) * D + sub * VEC;
      vp = (const T *)p.v_cache + (row * p.n_kv_heads + hkv) * D + sub * VEC;
    } else {
      kp = (const T *)p.k + (int64_t)(k0 + key) * p.k_stride + (int64_t)hkv * D + sub * VEC;
      vp = (const T *)p.v + (int64_t)(k0 + key) * p.v_stride + (int64_t)hkv * D + sub * VEC;
    }
    const u32x4 zero = {0u, 0u, 0u, 0u};
    const u32x4 kraw = ok ? *reinterpret_cast<const u32x4 *>(kp) : zero;
    const u32x4 vraw = ok ? *reinterpret_cast<const u32x4 *>(vp) : zero;
    float s = dot16<Tag>(qraw, kraw);
    for (int off = 1; off < LPT; off <<= 1) s += __shfl_xor(s, off, 64);
    s = ok ? s * scale_log2 : -INFINITY;
    const float mx = fmaxf(m, s);
    const float alpha = __builtin_amdgcn_exp2f(m - mx);
    const float e = __builtin_amdgcn_exp2f(s - mx);
    m = mx;
    l = l * alpha + e;
    float vf[VEC];
    if constexpr (VEC == 8) unpack8<Tag>(vraw, vf);
    else {
#pragma unroll
      for (int j = 0; j < 4; ++j) vf[j] = u2f(vraw[j]);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = fmaf(e, vf[j], acc[j] * alpha);
  }
  // merge lane groups
  float mm = m;
  for (int off = LPT; off < 64; off <<= 1) mm = fmaxf(mm, __shfl_xor(mm, off, 64));
  const float f = __builtin_amdgcn_exp2f(m - mm);
  float ll = l * f;
  for (int off = LPT; off < 64; off <<= 1) ll += __shfl_xor(ll, off, 64);
  float o[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float a = acc[j] * f;
    for (int off = LPT; off < 64; off <<= 1) a += __shfl_xor(a, off, 64);
    o[j] = ll > 0.f ? a / ll : 0.f;
  }
  if (grp == 0) {
    T *op = (T *)p.out + ((int64_t)tok * p.n_heads + h) * D + sub * VEC;
    u32x4 packed;
    if constexpr (VEC == 8) packed = pack8<Tag>(o);
    else {
#pragma unroll
      for (int j = 0; j < 4; ++j) packed[j] = f2u(o[j]);
    }
    *reinterpret_cast<u32x4 *>(op) = packed;
  }
}

}  // namespace

int mv_prefill_simple_launch(const PrefillParams &p, int64_t max_total_q, int dtype, hipStream_t stream) {
  const int64_t items = max_total_q * p.n_heads;
  if (items == 0) return MV_OK;
  const unsigned blocks = (unsigned)((items + 3) / 4);
  if (p.head_dim == 128) {
    MV_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((prefill_simple_kernel<Tag, 128>), dim3(blocks), dim3(256), 0, stream, p));
  } else if (p.head_dim == 64) {
    MV_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((prefill_simple_kernel<Tag, 64>), dim3(blocks), dim3(256), 0, stream, p));
  } else {
    return MV_E_SHAPE;
  }
  MV_LAUNCH_CHECK();
  return MV_OK;
}
