// Does one wave per SIMD overlap its own VALU work with its own in-flight MFMAs?  Cycles per v_mfma_f32_32x32x16_bf16
// for: bare MFMAs; + N independent v_fma fillers per MFMA; + fillers with v_exp; the same with the accumulators in AGPRs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int FILL, int EXPS, bool ACC_AGPR>
__global__ void __launch_bounds__(256, 1) probe(float *out, long long *cyc, int iters) {
  u32x4 a = {threadIdx.x, 2u, 3u, 4u}, b = {5u, 6u, 7u, threadIdx.x};
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float f[8];
  for (int j = 0; j < 8; ++j) f[j] = (float)threadIdx.x * 0.001f + j;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if constexpr (ACC_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m & 3]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < FILL; ++k) {
        if (k < EXPS) asm volatile("v_exp_f32 %0, %0" : "+v"(f[k & 7]));
        else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[k & 7]) : "v"(f[(k + 1) & 7]));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  for (int j = 0; j < 8; ++j) s += f[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int FILL, int EXPS, bool AG>
void run(const char *name, float *out, long long *cyc) {
  const int iters = 200, blocks = 256;
  hipLaunchKernelGGL((probe<FILL, EXPS, AG>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<FILL, EXPS, AG>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
  const double n = iters * 16.0;
  printf("%-34s fill %d (exp %d): %7.1f us  %6.1f ns/MFMA  counter %8.1f ticks/MFMA  -> %6.1f TFLOP/s chip\n", name, FILL, EXPS, ms * 1e3,
         ms * 1e6 / n, avg / n, 256.0 * 4 * n * 32768 / (ms * 1e-3) / 1e12);
}

int main() {
  float *out; long long *cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  run<0, 0, false>("acc in VGPRs", out, cyc);
  run<2, 0, false>("acc in VGPRs", out, cyc);
  run<4, 0, false>("acc in VGPRs", out, cyc);
  run<6, 0, false>("acc in VGPRs", out, cyc);
  run<8, 0, false>("acc in VGPRs", out, cyc);
  run<4, 1, false>("acc in VGPRs", out, cyc);
  run<4, 2, false>("acc in VGPRs", out, cyc);
  run<7, 2, false>("acc in VGPRs", out, cyc);
  run<0, 0, true>("acc in AGPRs", out, cyc);
  run<4, 0, true>("acc in AGPRs", out, cyc);
  run<6, 0, true>("acc in AGPRs", out, cyc);
  run<7, 2, true>("acc in AGPRs", out, cyc);
  return 0;
}
