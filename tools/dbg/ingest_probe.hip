// How many bytes per second can ONE CU take in (global -> LDS by LDS-DMA, or global -> registers)?  The number
// that bounds the M = 256 projections (csrc/linear_mid_m.hip).  One workgroup per CU streams `rows_x` rows of a
// shared 1 MB x (L2-resident, read by every workgroup) and `ROWS - rows_x` rows of its own weight slice (HBM)
// over K in 64-element blocks, through the same ring the GEMM uses, computing nothing.
//   hipcc -O3 --offload-arch=gfx950 tools/dbg/ingest_probe.hip -o tools/dbg/ingest_probe.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PAT: 0 linear source, 1 xor swizzle, 2 rotate swizzle.  NT: aux bits of the weight loads (2 = nt).
template <int ROWS, int NW, int STAGES, int PAT, int AUXW>
__global__ void __launch_bounds__(NW * 64) ring_kernel(const uint16_t *x, const uint16_t *w, int K, int rows_x, int64_t w_rows_total,
                                                       int iter, float *sink) {
  constexpr int P = ROWS / 8 / NW;
  constexpr int STAGE_BYTES = ROWS * 128;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rows_w = ROWS - rows_x;
  const uint16_t *src[P];
  bool is_w[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const int row = (wave + j * NW) * 8 + (lane >> 3);
    int chunk = lane & 7;
    if (PAT == 1) chunk ^= (row & 7);
    if (PAT == 2) chunk = (chunk + row) & 7;
    if (row < rows_x) {
      src[j] = x + (int64_t)((blockIdx.x & 1) * 128 % 256 + row % 256) * K + chunk * 8;
      is_w[j] = false;
    } else {
      int64_t wr = ((int64_t)(iter * gridDim.x + blockIdx.x) * rows_w + (row - rows_x)) % w_rows_total;
      src[j] = w + wr * K + chunk * 8;
      is_w[j] = true;
    }
  }
  auto issue = [&](int kb, int slot) {
    char *dst = lds + slot * STAGE_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < P; ++j) {
      if (AUXW != 0 && is_w[j])
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[j] + kb * 64),
                                         (__attribute__((address_space(3))) void *)(dst + j * NW * 1024), 16, 0, AUXW);
      else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[j] + kb * 64),
                                         (__attribute__((address_space(3))) void *)(dst + j * NW * 1024), 16, 0, 0);
    }
  };
  const int nk = K / 64;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue(s, s);
  for (int i = 0; i < nk; ++i) {
    const int ahead = min(nk - 1 - i, STAGES - 2);
    if (ahead == STAGES - 2) wait_vmcnt<(STAGES - 2) * P>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (i + STAGES - 1 < nk) issue(i + STAGES - 1, (i + STAGES - 1) % STAGES);
  }
  if (sink && K < 0) sink[threadIdx.x] = ((float *)lds)[threadIdx.x];
}

// global -> registers, the same rows; UNR row-pieces in flight per wave
template <int ROWS, int NW, int UNR>
__global__ void __launch_bounds__(NW * 64) reg_kernel(const uint16_t *x, const uint16_t *w, int K, int rows_x, int64_t w_rows_total,
                                                      int iter, float *sink) {
  constexpr int P = ROWS / 8 / NW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rows_w = ROWS - rows_x;
  const uint16_t *src[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const int row = (wave + j * NW) * 8 + (lane >> 3);
    const int chunk = lane & 7;
    if (row < rows_x) src[j] = x + (int64_t)(row % 256) * K + chunk * 8;
    else {
      int64_t wr = ((int64_t)(iter * gridDim.x + blockIdx.x) * rows_w + (row - rows_x)) % w_rows_total;
      src[j] = w + wr * K + chunk * 8;
    }
  }
  u32x4 acc = {0, 0, 0, 0};
  const int nk = K / 64;
  for (int i = 0; i < nk; i += UNR) {
    u32x4 v[UNR][P];
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int j = 0; j < P; ++j) v[u][j] = *(const u32x4 *)(src[j] + (i + u) * 64);
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int j = 0; j < P; ++j) acc ^= v[u][j];
  }
  if (sink && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[threadIdx.x] = 1.f;
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

template <typename F>
float time_launches(F launch, int n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch(i);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) launch(i + 3);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / n;
}

int main() {
  const int K = 2048;
  const int64_t w_rows_total = (int64_t)(1 << 30) / (K * 2);   // 1 GiB of weights
  uint16_t *x, *w; float *sink;
  CK(hipMalloc(&x, 256 * K * 2 * 2)); CK(hipMalloc(&w, (size_t)w_rows_total * K * 2 + (1 << 20))); CK(hipMalloc(&sink, 1 << 16));
  CK(hipMemset(x, 0, 256 * K * 2 * 2)); CK(hipMemset(w, 0, (size_t)w_rows_total * K * 2));
  const int grids[] = {256};
#define RING(ROWS, NW, ST, PAT, AUX, RX, G)                                                                        \
  do {                                                                                                             \
    auto kern = ring_kernel<ROWS, NW, ST, PAT, AUX>;                                                               \
    size_t ldsb = (size_t)ST * ROWS * 128;                                                                         \
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));            \
    float us = time_launches([&](int it) { hipLaunchKernelGGL(kern, dim3(G), dim3(NW * 64), ldsb, 0, x, w, K, RX, w_rows_total, it, sink); }, 20); \
    CK(hipGetLastError());                                                                                         \
    printf("ring rows %3d (x %3d) waves %2d stages %d pat %d aux %d grid %4d: %7.2f us  %6.1f GB/s per WG  %5.2f TB/s chip\n", ROWS, RX, NW,  \
           ST, PAT, AUX, G, us, (double)ROWS * K * 2 / us / 1e3, (double)ROWS * K * 2 * G / us / 1e6);             \
  } while (0)
#define REGK(ROWS, NW, UNR, RX, G)                                                                                 \
  do {                                                                                                             \
    auto kern = reg_kernel<ROWS, NW, UNR>;                                                                         \
    float us = time_launches([&](int it) { hipLaunchKernelGGL(kern, dim3(G), dim3(NW * 64), 0, 0, x, w, K, RX, w_rows_total, it, sink); }, 20); \
    CK(hipGetLastError());                                                                                         \
    printf("regs rows %3d (x %3d) waves %2d unroll %d grid %4d: %7.2f us  %6.1f GB/s per WG  %5.2f TB/s chip\n", ROWS, RX, NW, UNR, G, us,   \
           (double)ROWS * K * 2 / us / 1e3, (double)ROWS * K * 2 * G / us / 1e6);                                  \
  } while (0)
  // empty-kernel floor
  {
    auto kern = ring_kernel<256, 8, 4, 1, 0>;
    size_t ldsb = 4 * 256 * 128;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    float us = time_launches([&](int it) { hipLaunchKernelGGL(kern, dim3(256), dim3(512), ldsb, 0, x, w, 0, 128, w_rows_total, it, sink); }, 20);
    printf("empty launch (K=0): %.2f us\n", us);
  }
  for (int rx : {256, 128, 0}) {
    RING(256, 8, 4, 0, 0, rx, 256);
    RING(256, 8, 4, 1, 0, rx, 256);
    RING(256, 8, 4, 2, 0, rx, 256);
    RING(256, 8, 4, 1, 2, rx, 256);
    RING(256, 8, 3, 1, 0, rx, 256);
    RING(256, 8, 5, 1, 0, rx, 256);
    RING(256, 16, 4, 1, 0, rx, 256);
    RING(256, 4, 4, 1, 0, rx, 256);
    RING(128, 8, 4, 1, 0, rx / 2, 512);   // two workgroups per CU, half the ring each
    RING(128, 4, 4, 1, 0, rx / 2, 512);
    RING(128, 4, 6, 1, 0, rx / 2, 512);
    RING(64, 4, 4, 1, 0, rx / 4, 1024);   // four per CU
    RING(64, 4, 8, 1, 0, rx / 4, 1024);
    REGK(256, 8, 4, rx, 256);
    REGK(256, 8, 8, rx, 256);
    REGK(256, 16, 4, rx, 256);
    REGK(128, 8, 8, rx / 2, 512);
    REGK(128, 8, 8, rx / 2, 1024);
    REGK(64, 4, 8, rx / 4, 2048);
  }
  // fewer workgroups than CUs: does a CU alone go faster?
  RING(256, 8, 4, 1, 0, 128, 64);
  RING(256, 8, 4, 1, 0, 128, 128);
  RING(256, 8, 4, 1, 0, 256, 64);
  RING(256, 8, 4, 1, 0, 0, 64);
  return 0;
}
