// Can two consecutive GEMV launches of a decode step overlap?  A chain of y = W_i x (N = K = 2048, bf16, batch 1),
// every launch streaming a different 8-MB matrix, run three ways:
//   0  one stream, the kernel boundary is the dependency (what the engine does today)
//   1  one stream, plus the device-side flag protocol with agent-scope fences (cost of the protocol alone)
//   3  one stream, the "light" protocol: no fences, exchanged data through agent-scope relaxed atomics
//   2  two streams, launch i on stream i & 1 and NO event between them: launch i + 1 prefetches its weight rows
//      into registers while launch i still runs, then waits on a device counter that launch i's workgroups bump
//      after their stores ("software dependent launch"); at most two launches are ever in flight
// each eagerly and as a captured hipGraph.  Spins are bounded by the wall clock (s_memrealtime, 20 ms).
//   hipcc -O3 --offload-arch=gfx950 tools/dbg/chain_probe.hip -o tools/dbg/chain_probe.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
constexpr int N = 2048, K = 2048, WGS = 256, ROWS_PER_WG = N / WGS;  // 8 rows, 2 per wave

__device__ __forceinline__ float dot8(const u32x4 &a, const u32x4 &b, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a[i]), __builtin_bit_cast(bf16x2, b[i]), acc, false);
  return acc;
}

// PROTO: 0 = none, 1 = fences (agent-scope release / acquire), 2 = "light": no fence — x and y travel through
// agent-scope relaxed atomic loads / stores (sc1 accesses), the counter bump is ordered behind the y stores by
// s_waitcnt vmcnt(0) and the barrier
template <int PROTO>
__global__ void __launch_bounds__(256) gemv(const uint16_t *__restrict__ w, const uint16_t *x, uint16_t *y,
                                            unsigned *flag_in, unsigned *flag_out, unsigned want, unsigned *err) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * ROWS_PER_WG + wave * 2;
  // the weights do not depend on the predecessor: fetch them first
  u32x4 wr[2][4];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      wr[r][j] = __builtin_nontemporal_load((const u32x4 *)(w + (size_t)(row0 + r) * K + j * 512 + lane * 8));
  constexpr bool FLAGS = PROTO != 0;
  if (FLAGS) {
    if (threadIdx.x == 0) {
      const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
      while (__hip_atomic_load(flag_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) {  // 100 MHz: 20 ms
          atomicAdd(err, 1u);
          break;
        }
      }
    }
    __syncthreads();
    if (PROTO == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  u32x4 xr[4];
  if (PROTO == 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        xr[j][e] = __hip_atomic_load((const uint32_t *)(x + j * 512 + lane * 8) + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) xr[j] = *(const u32x4 *)(x + j * 512 + lane * 8);
  }
  float acc[2] = {0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[r] = dot8(wr[r][j], xr[j], acc[r]);
#pragma unroll
  for (int r = 0; r < 2; ++r)
    for (int off = 32; off; off >>= 1) acc[r] += __shfl_xor(acc[r], off, 64);
  if (lane == 0) {
    const float v0 = acc[0] * 0.02f + 1.0f, v1 = acc[1] * 0.02f + 1.0f;  // keep the chain's values in range
    const uint32_t pk = (__builtin_bit_cast(uint32_t, v0) >> 16) | (__builtin_bit_cast(uint32_t, v1) & 0xffff0000u);
    if (PROTO == 2) __hip_atomic_store((uint32_t *)(y + row0), pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *(uint32_t *)(y + row0) = pk;
  }
  if (FLAGS) {
    if (PROTO == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (PROTO == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(flag_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
  // launches per "step"; distinct weight matrices (48 x 8 MB = 384 MB > the 256 MB Infinity Cache; pass 12 to keep
  // them all resident in it, 2 for L2-resident: what would a prefetcher buy?)
  const int CHAIN = 224, NW = argc > 1 ? atoi(argv[1]) : 48;
  const int MODES = argc > 2 ? atoi(argv[2]) : 3;
  std::vector<uint16_t *> w(NW);
  for (auto &p : w) { CK(hipMalloc(&p, (size_t)N * K * 2)); CK(hipMemset(p, 0x3c, (size_t)N * K * 2)); }
  uint16_t *xb[2]; unsigned *flags, *err;
  CK(hipMalloc(&xb[0], K * 2)); CK(hipMalloc(&xb[1], K * 2)); CK(hipMemset(xb[0], 0x3f, K * 2)); CK(hipMemset(xb[1], 0x3f, K * 2));
  CK(hipMalloc(&flags, (CHAIN + 1) * sizeof(unsigned))); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  hipStream_t s[2]; CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
  hipEvent_t e0, e1, fork, join; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));

  unsigned epoch = 0;  // flags[i] counts the workgroups of launch i - 1 that have finished, over all steps so far
  auto step = [&](int mode) {
    // mode 1/2: launch i waits for flags[i] >= epoch * WGS and bumps flags[i + 1]; launch 0 waits for the previous
    // step's last launch (flags[0] >= (epoch - 1) * WGS)
    for (int i = 0; i < CHAIN; ++i) {
      hipStream_t st = mode == 2 ? s[i & 1] : s[0];
      if (mode == 0)
        hipLaunchKernelGGL(gemv<0>, dim3(WGS), dim3(256), 0, st, w[i % NW], xb[i & 1], xb[(i + 1) & 1], nullptr, nullptr, 0u, err);
      else if (mode == 3)
        hipLaunchKernelGGL(gemv<2>, dim3(WGS), dim3(256), 0, st, w[i % NW], xb[i & 1], xb[(i + 1) & 1], flags + i,
                           flags + (i + 1) % CHAIN, (i == 0 ? epoch - 1 : epoch) * WGS, err);
      else
        hipLaunchKernelGGL(gemv<1>, dim3(WGS), dim3(256), 0, st, w[i % NW], xb[i & 1], xb[(i + 1) & 1], flags + i,
                           flags + (i + 1) % CHAIN, (i == 0 ? epoch - 1 : epoch) * WGS, err);
    }
  };
  for (int mode : {0, 1, 3}) {
    if (mode >= MODES && mode != 3) continue;
    for (int graph = 0; graph < 2; ++graph) {
      CK(hipMemset(flags, 0, (CHAIN + 1) * sizeof(unsigned)));
      CK(hipDeviceSynchronize());
      epoch = 0;
      hipGraphExec_t exec = nullptr;
      const int reps = 20;
      if (graph) {
        // a graph freezes `want`: capture one step per epoch value would defeat the point, so the captured step
        // resets the counters itself (a memset node at its head)
        hipGraph_t g;
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
        CK(hipMemsetAsync(flags, 0, (CHAIN + 1) * sizeof(unsigned), s[0]));
        if (mode == 2) { CK(hipEventRecord(fork, s[0])); CK(hipStreamWaitEvent(s[1], fork, 0)); }
        epoch = 1;
        step(mode);
        if (mode == 2) { CK(hipEventRecord(join, s[1])); CK(hipStreamWaitEvent(s[0], join, 0)); }
        CK(hipStreamEndCapture(s[0], &g));
        CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, s[0]));
        CK(hipStreamSynchronize(s[0]));
        CK(hipEventRecord(e0, s[0]));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(exec, s[0]));
        CK(hipEventRecord(e1, s[0]));
        CK(hipEventSynchronize(e1));
      } else {
        for (int i = 0; i < 3; ++i) { ++epoch; step(mode); }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s[0]));
        for (int i = 0; i < reps; ++i) { ++epoch; step(mode); }
        if (mode == 2) { CK(hipEventRecord(join, s[1])); CK(hipStreamWaitEvent(s[0], join, 0)); }
        CK(hipEventRecord(e1, s[0]));
        CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      const double us = ms * 1e3 / (reps * CHAIN);
      printf("mode %d %-6s: %6.2f us per launch  (%5.2f TB/s of weights)  expired waits %u\n", mode, graph ? "graph" : "eager", us,
             (double)N * K * 2 / us / 1e6, herr);
      fflush(stdout);
      if (herr) { printf("a bounded wait expired: stopping\n"); return 2; }
    }
  }
  return 0;
}
