// Round 3: does a projection's DMA ring go faster when the WEIGHT stream does not use the LDS ring?
// One workgroup per CU streams what a 128 x 96 tile of the Qwen3-1.7B gate_up projection (M = 256, N = 12288, K = 2048) takes in:
// 128 rows of x (L2-resident, re-read by every workgroup of a row half) and 96 rows of packed W (HBM; the two row-half workgroups
// b and b + 8 stream the same tile), nothing is computed.  Modes:
//   0  both through LDS-DMA rings (x 4 stages of 16 KB, W 7 stages of 12 KB): what linear_mid_m_kernel does (ring-only: 14.3 us K loop)
//   1  x through a 9-stage LDS-DMA ring (144 KB), W straight to registers (4 waves, DEPTH k-blocks of 3 fragments in flight each)
//   2  x alone (mode 1 without the W waves)      3  W-to-registers alone
//   hipcc -O3 --offload-arch=gfx950 tools/dbg/wreg_probe.hip -o tools/dbg/wreg_probe.bin && tools/dbg/wreg_probe.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
constexpr int K = 2048, M = 256, N = 12288, BM = 128, BN = 96, NKB = K / 64;

template <int NN>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NN) : "memory"); }

template <int MODE, int SX, int SW, int DEPTH>
__global__ void __launch_bounds__(512) probe(const uint16_t *x, const uint16_t *wp, float *sink) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x / 16, r = blockIdx.x % 16;
  const int n_tile = q * 8 + (r & 7), m_tile = r >> 3;
  if (wave < 4) {
    if (MODE == 3) return;
    // x loaders: 4 waves x 4 pieces (8 rows x 128 B) per k-block
    const uint16_t *src[4];
    for (int j = 0; j < 4; ++j) {
      const int row = (wave + j * 4) * 8 + (lane >> 3);
      src[j] = x + (int64_t)(m_tile * BM + row) * K + (((lane & 7) ^ (row & 7)) << 3);
    }
    auto issue = [&](int i) {
      char *dst = lds + (i % SX) * (BM * 128) + wave * 1024;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[j] + i * 64),
                                         (__attribute__((address_space(3))) void *)(dst + j * 4096), 16, 0, 0);
    };
    for (int s = 0; s < SX - 1; ++s) issue(s);
    for (int i = 0; i < NKB; ++i) {
      // block i landed <=> at most (SX - 2) later blocks outstanding
      if (NKB - 1 - i >= SX - 2) wait_vm<(SX - 2) * 4>(); else wait_vm<0>();
      if (i + SX - 1 < NKB) issue(i + SX - 1);
    }
    return;
  }
  const int w = wave - 4;
  if (MODE == 2) return;
  if (MODE == 0) {
    // W through an LDS ring: 3 loader waves x 4 pieces of one fragment (1 KiB) per k-block (wave 3 idle)
    if (w == 3) return;
    char *wl = lds + SX * BM * 128;
    const uint16_t *src[4];
    for (int j = 0; j < 4; ++j) {
      const int piece = w + j * 3, t = piece >> 1, ks = piece & 1;
      src[j] = wp + ((int64_t)(n_tile * 6 + t) * (K / 32) + ks) * 512 + lane * 8;
    }
    auto issue = [&](int i) {
      char *dst = wl + (i % SW) * (BN * 128) + w * 1024;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[j] + i * 1024),
                                         (__attribute__((address_space(3))) void *)(dst + j * 3072), 16, 0, 0);
    };
    for (int s = 0; s < SW - 1; ++s) issue(s);
    for (int i = 0; i < NKB; ++i) {
      if (NKB - 1 - i >= SW - 2) wait_vm<(SW - 2) * 4>(); else wait_vm<0>();
      if (i + SW - 1 < NKB) issue(i + SW - 1);
    }
    return;
  }
  // W straight to registers: wave w owns 3 of the tile's 12 fragments per k-block (1.5 n16-tiles x 2 k-steps), DEPTH k-blocks in flight
  const uint16_t *base[3];
  for (int j = 0; j < 3; ++j) {
    const int piece = w * 3 + j, t = piece >> 1, ks = piece & 1;
    base[j] = wp + ((int64_t)(n_tile * 6 + t) * (K / 32) + ks) * 512 + lane * 8;
  }
  u32x4 acc = {0, 0, 0, 0};
  u32x4 buf[DEPTH][3];
#pragma unroll
  for (int s = 0; s < DEPTH; ++s)
#pragma unroll
    for (int j = 0; j < 3; ++j) buf[s][j] = __builtin_nontemporal_load((const u32x4 *)(base[j] + s * 1024));
  for (int i0 = 0; i0 < NKB; i0 += DEPTH) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      const int i = i0 + s;
#pragma unroll
      for (int j = 0; j < 3; ++j) acc ^= buf[s][j];
      if (i + DEPTH < NKB) {
#pragma unroll
        for (int j = 0; j < 3; ++j) buf[s][j] = __builtin_nontemporal_load((const u32x4 *)(base[j] + (i + DEPTH) * 1024));
      }
    }
  }
  if (acc[0] == 0x12345678u && sink) sink[threadIdx.x] = (float)acc[1];
}

template <int MODE, int SX, int SW, int DEPTH>
float run(const uint16_t *x, const std::vector<uint16_t *> &wps, float *sink, const char *label) {
  auto kern = probe<MODE, SX, SW, DEPTH>;
  const int ldsb = MODE == 0 ? SX * BM * 128 + SW * BN * 128 : SX * BM * 128;
  hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 40;
  for (int i = 0; i < 6; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), ldsb, 0, x, wps[i % wps.size()], sink);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), ldsb, 0, x, wps[i % wps.size()], sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const float us = ms * 1e3f / iters;
  printf("%-58s %7.2f us per launch\n", label, us);
  return us;
}

int main() {
  uint16_t *x;
  hipMalloc(&x, (size_t)M * K * 2);
  hipMemset(x, 1, (size_t)M * K * 2);
  std::vector<uint16_t *> wps(12);
  for (auto &p : wps) { hipMalloc(&p, (size_t)N * K * 2); hipMemset(p, 1, (size_t)N * K * 2); }
  float *sink;
  hipMalloc(&sink, 4096);
  for (int rep = 0; rep < 3; ++rep) {
    run<0, 4, 7, 1>(x, wps, sink, "0: x + W through LDS rings (4 / 7 stages)");
    run<2, 4, 7, 1>(x, wps, sink, "2: x alone, 4-stage ring");
    run<2, 9, 7, 1>(x, wps, sink, "2: x alone, 9-stage ring");
    run<3, 4, 7, 8>(x, wps, sink, "3: W to registers alone, 8 k-blocks in flight");
    run<3, 4, 7, 16>(x, wps, sink, "3: W to registers alone, 16 k-blocks in flight");
    run<1, 4, 7, 8>(x, wps, sink, "1: x 4-stage ring + W to registers (8 in flight)");
    run<1, 9, 7, 8>(x, wps, sink, "1: x 9-stage ring + W to registers (8 in flight)");
    run<1, 9, 7, 16>(x, wps, sink, "1: x 9-stage ring + W to registers (16 in flight)");
    run<1, 6, 7, 16>(x, wps, sink, "1: x 6-stage ring + W to registers (16 in flight)");
    printf("\n");
  }
  return 0;
}
