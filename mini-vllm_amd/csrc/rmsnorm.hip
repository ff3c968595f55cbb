// K6 — RMSNorm and fused residual-add + RMSNorm
// (reference: minivllm/models/layers/layernorm.py:81-105).
// One workgroup per row; the row lives in registers between the
// sum-of-squares pass and the scale pass, so HBM sees each element once.
#include "common.h"

template <typename Tag>
__device__ __forceinline__ void load_vec(const typename elem<Tag>::T *p, float (&f)[elem<Tag>::VEC]) {
  const u32x4 raw = *reinterpret_cast<const u32x4 *>(p);
  if constexpr (elem<Tag>::VEC == 8) {
    unpack8<Tag>(raw, f);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = u2f(raw[i]);
  }
}
template <typename Tag>
__device__ __forceinline__ void store_vec(typename elem<Tag>::T *p, const float (&f)[elem<Tag>::VEC]) {
  u32x4 raw;
  if constexpr (elem<Tag>::VEC == 8) {
    raw = pack8<Tag>(f);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) raw[i] = f2u(f[i]);
  }
  *reinterpret_cast<u32x4 *>(p) = raw;
}

// Block-wide sum; every thread gets the total.  `red` holds one float per wave.
__device__ __forceinline__ float block_sum(float x, float *red) {
  x = wave_sum(x);
  const int nw = blockDim.x >> 6;
  if (nw == 1) return x;
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = x;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// MAXV = 16-byte vectors per thread held in registers.
// PARTIAL: x is not a [rows, cols] tensor of dtype but `splits` fp32 partial sums [splits][rows][cols] of a
// K-split projection (mv_linear_mid_m_partial): they are summed in split order and rounded to dtype first —
// the rounding point of the GEMM whose output the reference's add_rms_forward reads.
template <typename Tag, int MAXV, bool ADD, bool PARTIAL = false>
__global__ void rmsnorm_kernel(const typename elem<Tag>::T *__restrict__ x, int64_t x_stride,
                               typename elem<Tag>::T *residual, int64_t res_stride,
                               const typename elem<Tag>::T *__restrict__ w,
                               typename elem<Tag>::T *__restrict__ out, int64_t out_stride,
                               int cols, float eps, const float *__restrict__ partial = nullptr, int splits = 0) {
  constexpr int VEC = elem<Tag>::VEC;
  __shared__ float red[16];
  const int row = blockIdx.x;
  const int nvec = cols / VEC;
  const typename elem<Tag>::T *xr = x + (int64_t)row * x_stride;
  float vals[MAXV][VEC];
  u32x4 wraw[MAXV];  // norm weights, fetched with the row (packed: 4 registers per 16-byte vector)
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = threadIdx.x + i * blockDim.x;
    if (c < nvec) {
      wraw[i] = *reinterpret_cast<const u32x4 *>(w + c * VEC);  // not a second round trip after the reduction
      if constexpr (PARTIAL) {
        static_assert(!PARTIAL || VEC == 8, "partial sums feed 16-bit outputs");
        const float *pp = partial + (int64_t)row * cols + c * VEC;
        const int64_t sstride = (int64_t)gridDim.x * cols;
        f32x4 lo = *reinterpret_cast<const f32x4 *>(pp), hi = *reinterpret_cast<const f32x4 *>(pp + 4);
        for (int sp = 1; sp < splits; ++sp) {
          const f32x4 l2 = *reinterpret_cast<const f32x4 *>(pp + sp * sstride);
          const f32x4 h2 = *reinterpret_cast<const f32x4 *>(pp + sp * sstride + 4);
          lo += l2;
          hi += h2;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vals[i][j] = elem<Tag>::ld(elem<Tag>::st(lo[j]));
          vals[i][4 + j] = elem<Tag>::ld(elem<Tag>::st(hi[j]));
        }
      } else {
        load_vec<Tag>(xr + c * VEC, vals[i]);
      }
      if constexpr (ADD) {
        typename elem<Tag>::T *rr = residual + (int64_t)row * res_stride + c * VEC;
        float r[VEC];
        load_vec<Tag>(rr, r);
#pragma unroll
        for (int j = 0; j < VEC; ++j) vals[i][j] = __fadd_rn(vals[i][j], r[j]);
        store_vec<Tag>(rr, vals[i]);  // residual <- (x + r) -> dtype
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) ss = fmaf(vals[i][j], vals[i][j], ss);
    }
  }
  ss = block_sum(ss, red);
  const float var = ss / (float)cols;
  const float rs = 1.0f / sqrtf(var + eps);
  typename elem<Tag>::T *orow = out + (int64_t)row * out_stride;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = threadIdx.x + i * blockDim.x;
    if (c < nvec) {
      float wv[VEC], y[VEC];
      if constexpr (VEC == 8) {
        unpack8<Tag>(wraw[i], wv);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[j] = u2f(wraw[i][j]);
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) y[j] = __fmul_rn(__fmul_rn(vals[i][j], rs), wv[j]);
      store_vec<Tag>(orow + c * VEC, y);
    }
  }
}

template <typename Tag, bool ADD>
static int launch_rmsnorm(const void *x, int64_t x_stride, void *residual, int64_t res_stride,
                          const void *w, void *out, int64_t out_stride, int rows, int cols,
                          float eps, hipStream_t stream) {
  typedef typename elem<Tag>::T T;
  constexpr int VEC = elem<Tag>::VEC;
  if (cols % VEC) return MV_E_SHAPE;
  if ((x_stride % VEC) || (out_stride % VEC) || (ADD && (res_stride % VEC))) return MV_E_ALIGN;
  const int nvec = cols / VEC;
  int threads = (nvec + 63) / 64 * 64;
  if (threads > 256) threads = nvec > 1024 ? 1024 : 256;
  const int per_thread = (nvec + threads - 1) / threads;
#define MV_RMS_LAUNCH(MV)                                                                        \
  hipLaunchKernelGGL((rmsnorm_kernel<Tag, MV, ADD>), dim3(rows), dim3(threads), 0, stream,       \
                     (const T *)x, x_stride, (T *)residual, res_stride, (const T *)w, (T *)out, \
                     out_stride, cols, eps)
  if (per_thread <= 1) MV_RMS_LAUNCH(1);
  else if (per_thread <= 2) MV_RMS_LAUNCH(2);
  else if (per_thread <= 4) MV_RMS_LAUNCH(4);
  else if (per_thread <= 8) MV_RMS_LAUNCH(8);
  else return MV_E_SHAPE;  // > 65536 (16-bit) / 32768 (fp32) columns
#undef MV_RMS_LAUNCH
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_rmsnorm(const void *x, int64_t x_stride, const void *w, void *out,
                          int64_t out_stride, int rows, int cols, float eps, int dtype,
                          void *stream) {
  if (rows == 0) return MV_OK;
  if (!x || !w || !out) return MV_E_NULLPTR;
  if (rows < 0 || cols <= 0) return MV_E_SHAPE;
  if (!mv_aligned16(x) || !mv_aligned16(w) || !mv_aligned16(out)) return MV_E_ALIGN;
  MV_DISPATCH_DTYPE(dtype, return (launch_rmsnorm<Tag, false>(x, x_stride, nullptr, 0, w, out, out_stride,
                                                              rows, cols, eps, (hipStream_t)stream)));
  return MV_OK;
}

extern "C" int mv_add_rmsnorm(const void *x, int64_t x_stride, void *residual, int64_t res_stride,
                              const void *w, void *out, int64_t out_stride, int rows, int cols,
                              float eps, int dtype, void *stream) {
  if (rows == 0) return MV_OK;
  if (!x || !w || !out || !residual) return MV_E_NULLPTR;
  if (rows < 0 || cols <= 0) return MV_E_SHAPE;
  if (!mv_aligned16(x) || !mv_aligned16(w) || !mv_aligned16(out) || !mv_aligned16(residual))
    return MV_E_ALIGN;
  MV_DISPATCH_DTYPE(dtype, return (launch_rmsnorm<Tag, true>(x, x_stride, residual, res_stride, w, out,
                                                             out_stride, rows, cols, eps,
                                                             (hipStream_t)stream)));
  return MV_OK;
}

// K6 behind a K-split projection: see rmsnorm_kernel<PARTIAL>.
extern "C" int mv_add_rmsnorm_partial(const float *partial, int splits, void *residual, int64_t res_stride,
                                      const void *w, void *out, int64_t out_stride, int rows, int cols,
                                      float eps, int dtype, void *stream) {
  if (rows == 0) return MV_OK;
  if (!partial || !w || !out || !residual) return MV_E_NULLPTR;
  if (rows < 0 || cols <= 0 || (cols & 7) || splits < 1 || splits > 16) return MV_E_SHAPE;
  if (dtype != MV_BF16 && dtype != MV_F16) return dtype == MV_F32 ? MV_E_SHAPE : MV_E_DTYPE;
  if (!mv_aligned16(partial) || !mv_aligned16(w) || !mv_aligned16(out) || !mv_aligned16(residual) ||
      (out_stride & 7) || (res_stride & 7))
    return MV_E_ALIGN;
  const int nvec = cols / 8;
  int threads = (nvec + 63) / 64 * 64;
  if (threads > 256) threads = nvec > 1024 ? 1024 : 256;
  const int per_thread = (nvec + threads - 1) / threads;
#define MV_RMSP_LAUNCH(TAG, MV)                                                                              \
  hipLaunchKernelGGL((rmsnorm_kernel<TAG, MV, true, true>), dim3(rows), dim3(threads), 0, (hipStream_t)stream, \
                     (const uint16_t *)nullptr, (int64_t)0, (uint16_t *)residual, res_stride, (const uint16_t *)w, \
                     (uint16_t *)out, out_stride, cols, eps, partial, splits)
#define MV_RMSP_DISPATCH(TAG)                 \
  do {                                        \
    if (per_thread <= 1) MV_RMSP_LAUNCH(TAG, 1);      \
    else if (per_thread <= 2) MV_RMSP_LAUNCH(TAG, 2); \
    else if (per_thread <= 4) MV_RMSP_LAUNCH(TAG, 4); \
    else if (per_thread <= 8) MV_RMSP_LAUNCH(TAG, 8); \
    else return MV_E_SHAPE;                   \
  } while (0)
  if (dtype == MV_BF16) MV_RMSP_DISPATCH(bf16_tag);
  else MV_RMSP_DISPATCH(f16_tag);
#undef MV_RMSP_DISPATCH
#undef MV_RMSP_LAUNCH
  MV_LAUNCH_CHECK();
  return MV_OK;
}

// K9 — SiluAndMul (reference: minivllm/models/layers/activation.py:70-96).
// The reference rounds silu(x1) to the activation dtype before the multiply;
// the same two rounding points are kept here.
template <typename Tag>
__global__ void __launch_bounds__(256)
silu_and_mul_kernel(const typename elem<Tag>::T *__restrict__ x, int64_t x_stride,
                    typename elem<Tag>::T *__restrict__ out, int64_t out_stride, int rows, int cols) {
  constexpr int VEC = elem<Tag>::VEC;
  constexpr int U = 4;   // vectors per thread and trip: 2 U loads in flight before the first exp (prefill rows: 4.3 -> TB/s)
  const int nvec = cols / VEC;
  // blockIdx.y walks the rows, blockIdx.x the vectors of a row (no 64-bit division per element)
  for (int r = blockIdx.y; r < rows; r += gridDim.y) {
    const typename elem<Tag>::T *xr = x + (int64_t)r * x_stride;
    typename elem<Tag>::T *orow = out + (int64_t)r * out_stride;
    for (int c0 = (blockIdx.x * blockDim.x + threadIdx.x); c0 < nvec; c0 += gridDim.x * blockDim.x * U) {
      float a[U][VEC], b[U][VEC];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * gridDim.x * blockDim.x;
        if (c < nvec) {   // read once: non-temporal
          if constexpr (VEC == 8) {
            unpack8<Tag>(__builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(xr + c * VEC)), a[u]);
            unpack8<Tag>(__builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(xr + cols + c * VEC)), b[u]);
          } else {
            load_vec<Tag>(xr + c * VEC, a[u]);
            load_vec<Tag>(xr + cols + c * VEC, b[u]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * gridDim.x * blockDim.x;
        if (c < nvec) {
          float y[VEC];
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            float s = a[u][j] / (1.0f + expf(-a[u][j]));
            s = elem<Tag>::ld(elem<Tag>::st(s));
            y[j] = __fmul_rn(s, b[u][j]);
          }
          store_vec<Tag>(orow + c * VEC, y);
        }
      }
    }
  }
}

extern "C" int mv_silu_and_mul(const void *x, int64_t x_stride, void *out, int64_t out_stride,
                               int rows, int cols, int dtype, void *stream) {
  if (rows == 0) return MV_OK;
  if (!x || !out) return MV_E_NULLPTR;
  if (rows < 0 || cols <= 0) return MV_E_SHAPE;
  if (!mv_aligned16(x) || !mv_aligned16(out)) return MV_E_ALIGN;
  MV_DISPATCH_DTYPE(dtype, {
    typedef typename elem<Tag>::T T;
    constexpr int VEC = elem<Tag>::VEC;
    if (cols % VEC) return MV_E_SHAPE;
    if ((x_stride % VEC) || (out_stride % VEC)) return MV_E_ALIGN;
    // x: workgroups along a row (4 vectors per thread and trip), y: rows; ~2048+ workgroups in flight, 65535 at most
    const int nvec = cols / VEC;
    int bx = (nvec + 256 * 4 - 1) / (256 * 4);
    if (bx < 1) bx = 1;
    int by = rows < 65535 ? rows : 65535;
    while ((int64_t)bx * by > 16384 && by > 1) by = (by + 1) / 2;
    hipLaunchKernelGGL((silu_and_mul_kernel<Tag>), dim3((unsigned)bx, (unsigned)by), dim3(256), 0,
                       (hipStream_t)stream, (const T *)x, x_stride, (T *)out, out_stride, rows, cols);
  });
  MV_LAUNCH_CHECK();
  return MV_OK;
}
