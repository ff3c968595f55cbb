// Does the paged-decode kernel's K/V access pattern cost HBM bandwidth?  Read 2 x 537 MB (K and V planes of 256
// sequences x 1024 tokens x 8 kv heads x 128 x bf16) into registers, 256-thread workgroups, 8 loads of 16 B in
// flight per lane, nt, in two mappings:
//   0  one workgroup per (sequence, kv head, split): 256-byte rows at a 2-KB stride (what paged_decode_kernel does)
//   1  one workgroup per (sequence, split) reading whole 2-KB token rows (each wave instruction = 1 KiB contiguous)
//   2  one workgroup per (sequence, split), wave w reading the 256-B pieces of kv heads {w, w + 4}: every
//      instruction is strided like mode 0, but the four waves of a workgroup cover whole rows between them
//   hipcc -O3 --offload-arch=gfx950 tools/dbg/ingest_probe3.hip -o tools/dbg/ingest_probe3.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int MODE>
__global__ void __launch_bounds__(256) rd(const char *k, const char *v, int tokens, int nsplit, float *sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 acc = {0, 0, 0, 0};
  if (MODE == 0) {
    const int hkv = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
    const int per = tokens / nsplit, t0 = split * per;
    const char *kb = k + ((int64_t)b * tokens) * 2048 + hkv * 256 + (lane & 15) * 16;
    const char *vb = v + ((int64_t)b * tokens) * 2048 + hkv * 256 + (lane & 15) * 16;
    // wave-wide load = 4 tokens x 256 B; 8 loads (32 tokens) per wave and iteration, 4 waves -> 128 tokens / iteration
    for (int t = t0 + wave * 32; t < t0 + per; t += 128) {
      u32x4 x[8], y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int tok = t + j * 4 + (lane >> 4);
        x[j] = __builtin_nontemporal_load((const u32x4 *)(kb + (int64_t)tok * 2048));
        y[j] = __builtin_nontemporal_load((const u32x4 *)(vb + (int64_t)tok * 2048));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= x[j] ^ y[j];
    }
  } else if (MODE == 2) {
    const int split = blockIdx.y, b = blockIdx.z;
    const int per = tokens / nsplit, t0 = split * per;
    const char *kb = k + ((int64_t)b * tokens) * 2048 + wave * 256 + (lane & 15) * 16;
    const char *vb = v + ((int64_t)b * tokens) * 2048 + wave * 256 + (lane & 15) * 16;
    // 8 loads = 16 tokens x 2 heads of this wave
    for (int t = t0; t < t0 + per; t += 16) {
      u32x4 x[8], y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int tok = t + (j >> 1) * 4 + (lane >> 4);
        x[j] = __builtin_nontemporal_load((const u32x4 *)(kb + (int64_t)tok * 2048 + (j & 1) * 1024));
        y[j] = __builtin_nontemporal_load((const u32x4 *)(vb + (int64_t)tok * 2048 + (j & 1) * 1024));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= x[j] ^ y[j];
    }
  } else {
    const int split = blockIdx.y, b = blockIdx.z;
    const int per = tokens / nsplit, t0 = split * per;
    const char *kb = k + ((int64_t)b * tokens) * 2048 + lane * 16;
    const char *vb = v + ((int64_t)b * tokens) * 2048 + lane * 16;
    // wave-wide load = 1 KiB contiguous = half a token row; 8 loads = 4 tokens per wave and iteration
    for (int t = t0 + wave * 4; t < t0 + per; t += 16) {
      u32x4 x[8], y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        x[j] = __builtin_nontemporal_load((const u32x4 *)(kb + (int64_t)(t + (j >> 1)) * 2048 + (j & 1) * 1024));
        y[j] = __builtin_nontemporal_load((const u32x4 *)(vb + (int64_t)(t + (j >> 1)) * 2048 + (j & 1) * 1024));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= x[j] ^ y[j];
    }
  }
  if (sink && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x1234567u) sink[threadIdx.x] = 1.f;
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

int main() {
  const int B = 256, T = 1024, LAYERS = 3;
  const size_t plane = (size_t)B * T * 2048;
  char *k[LAYERS], *v[LAYERS]; float *sink;
  for (int l = 0; l < LAYERS; ++l) { CK(hipMalloc(&k[l], plane)); CK(hipMalloc(&v[l], plane)); CK(hipMemset(k[l], 1, plane)); CK(hipMemset(v[l], 2, plane)); }
  CK(hipMalloc(&sink, 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    for (int mode = 0; mode < 3; ++mode) {
      for (int nsplit : {1, 2, 4, 8}) {
        if (mode == 0 && nsplit > 2) continue;
        dim3 grid(mode == 0 ? 8 : 1, nsplit, B);
        for (int i = 0; i < 2; ++i) {
          if (mode == 0) hipLaunchKernelGGL(rd<0>, grid, dim3(256), 0, 0, k[i % LAYERS], v[i % LAYERS], T, nsplit, sink);
          else if (mode == 2) hipLaunchKernelGGL(rd<2>, grid, dim3(256), 0, 0, k[i % LAYERS], v[i % LAYERS], T, nsplit, sink);
          else hipLaunchKernelGGL(rd<1>, grid, dim3(256), 0, 0, k[i % LAYERS], v[i % LAYERS], T, nsplit, sink);
        }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int n = 12;
        for (int i = 0; i < n; ++i) {
          if (mode == 0) hipLaunchKernelGGL(rd<0>, grid, dim3(256), 0, 0, k[i % LAYERS], v[i % LAYERS], T, nsplit, sink);
          else if (mode == 2) hipLaunchKernelGGL(rd<2>, grid, dim3(256), 0, 0, k[i % LAYERS], v[i % LAYERS], T, nsplit, sink);
          else hipLaunchKernelGGL(rd<1>, grid, dim3(256), 0, 0, k[i % LAYERS], v[i % LAYERS], T, nsplit, sink);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / n;
        printf("mode %d (%s) splits %d workgroups %5d: %7.1f us  %5.2f TB/s\n", mode, mode == 1 ? "whole 2-KB token rows per instruction pair" : (mode == 2 ? "whole rows per workgroup, 256-B pieces per wave" : "256-B rows of one kv head"),
               nsplit, grid.x * grid.y * grid.z, us, 2.0 * plane / us / 1e6);
      }
    }
  }
  return 0;
}
