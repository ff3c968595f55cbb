// Follow-up to ingest_probe.hip: how does the HBM side of the ring depend on how many CONTIGUOUS bytes of a
// weight row one DMA piece covers?  Each workgroup streams its own 1 MiB weight slice ("256 rows x 4 KiB")
// in 1-KiB pieces of (1024/RUN) rows x RUN bytes, PW pieces per wave and stage, plus PX pieces per wave and
// stage of the shared, L2-resident x (8 rows x 128 B each).  RUN = 0: the slice is read linearly (a weight
// matrix re-packed tile by tile).
//   hipcc -O3 --offload-arch=gfx950 tools/dbg/ingest_probe2.hip -o tools/dbg/ingest_probe2.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int NW, int STAGES, int PX, int PW, int RUN, int AUXW>
__global__ void __launch_bounds__(NW * 64) ring2(const char *x, const char *w, int64_t w_bytes, int w_pieces, int x_pieces_total,
                                                 int iter, float *sink) {
  constexpr int P = PX + PW;
  constexpr int STAGE_BYTES = P * NW * 1024;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char *wbase = w + ((int64_t)(iter * gridDim.x + blockIdx.x) << 20) % w_bytes;
  // piece id -> byte offset of this lane inside the 1 MiB slice (rows of 4096 B)
  auto w_off = [&](int piece) -> int {
    if (RUN == 0) return piece * 1024 + lane * 16;
    constexpr int R = RUN == 0 ? 128 : RUN;
    constexpr int RPP = 1024 / R;                 // rows per piece
    constexpr int RB = 256 / RPP;                 // row blocks
    const int cb = piece / RB, rb = piece % RB;   // column block outer, row block inner
    const int row = rb * RPP + lane / (R / 16), c = lane % (R / 16);
    return row * 4096 + cb * R + c * 16;
  };
  const int stages_total = w_pieces > 0 ? w_pieces / (PW > 0 ? PW * NW : 1) : x_pieces_total / (PX * NW);
  auto issue = [&](int s, int slot) {
    char *dst = lds + slot * STAGE_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < PX; ++j) {
      const int piece = (s * PX + j) * NW + wave;             // x: 256 rows x 4096 B, 8 rows x 128 B pieces, wraps
      const int cb = (piece / 32) % 32, rb = piece % 32;
      const char *src = x + (rb * 8 + (lane >> 3)) * 4096 + cb * 128 + (lane & 7) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(dst + j * NW * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int piece = (s * PW + j) * NW + wave;
      const char *src = wbase + w_off(piece);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(dst + (PX + j) * NW * 1024), 16, 0, AUXW);
    }
  };
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < stages_total) issue(s, s);
  for (int i = 0; i < stages_total; ++i) {
    const int ahead = min(stages_total - 1 - i, STAGES - 2);
    if (ahead == STAGES - 2) wait_vmcnt<(STAGES - 2) * P>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (i + STAGES - 1 < stages_total) issue(i + STAGES - 1, (i + STAGES - 1) % STAGES);
  }
  if (sink && w_pieces < 0) sink[threadIdx.x] = ((float *)lds)[threadIdx.x];
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

template <typename F>
float time_launches(F launch, int n) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch(i);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < n; ++i) launch(i + 3);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / n;
}

int main() {
  const int64_t w_bytes = (int64_t)2 << 30;
  char *x, *w; float *sink;
  CK(hipMalloc(&x, 1 << 20)); CK(hipMalloc(&w, (size_t)w_bytes + (2 << 20))); CK(hipMalloc(&sink, 1 << 16));
  CK(hipMemset(x, 1, 1 << 20)); CK(hipMemset(w, 1, (size_t)w_bytes));
  const float floor_us = 2.5f;
#define RUNK(NW, ST, PX, PW, RUN, AUX, G, WP, XP)                                                                    \
  do {                                                                                                               \
    auto kern = ring2<NW, ST, PX, PW, RUN, AUX>;                                                                     \
    size_t ldsb = (size_t)ST * (PX + PW) * NW * 1024;                                                                \
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));              \
    float us = time_launches([&](int it) { hipLaunchKernelGGL(kern, dim3(G), dim3(NW * 64), ldsb, 0, x, w, w_bytes, WP, XP, it, sink); }, 20); \
    CK(hipGetLastError());                                                                                           \
    const int st_total = (WP) > 0 ? (WP) / ((PW) * NW) : (XP) / ((PX) * NW);                                          \
    const double wb = (double)(WP) * 1024 * (G), xb = (double)st_total * (PX) * NW * 1024 * (G);                      \
    printf("waves %2d stages %d px %d pw %d run %4d aux %d grid %4d lds %3zu KB: %7.2f us | HBM %5.2f TB/s  x/CU %6.1f GB/s  (minus %.1f us floor: HBM %5.2f TB/s)\n", \
           NW, ST, PX, PW, RUN, AUX, G, ldsb >> 10, us, wb / us / 1e6, xb / (G) / us / 1e3, floor_us, wb / (us - floor_us) / 1e6); \
  } while (0)
  // W only: 1 MiB per workgroup, 4 pieces per wave and stage (32 KB stages), 4 stages
  RUNK(8, 4, 0, 4, 128, 0, 256, 1024, 0);
  RUNK(8, 4, 0, 4, 256, 0, 256, 1024, 0);
  RUNK(8, 4, 0, 4, 512, 0, 256, 1024, 0);
  RUNK(8, 4, 0, 4, 1024, 0, 256, 1024, 0);
  RUNK(8, 4, 0, 4, 0, 0, 256, 1024, 0);
  RUNK(8, 4, 0, 4, 128, 2, 256, 1024, 0);
  RUNK(8, 4, 0, 4, 512, 2, 256, 1024, 0);
  RUNK(8, 4, 0, 4, 1024, 2, 256, 1024, 0);
  RUNK(8, 4, 0, 4, 0, 2, 256, 1024, 0);
  // smaller stages, deeper ring
  RUNK(8, 8, 0, 2, 1024, 2, 256, 1024, 0);
  RUNK(8, 8, 0, 2, 0, 2, 256, 1024, 0);
  RUNK(4, 8, 0, 4, 0, 2, 256, 1024, 0);
  RUNK(8, 2, 0, 8, 0, 2, 256, 1024, 0);
  RUNK(8, 3, 0, 4, 0, 2, 256, 1024, 0);
  // two workgroups per CU (64 KB rings)
  RUNK(8, 4, 0, 2, 0, 2, 512, 1024, 0);
  RUNK(8, 4, 0, 2, 1024, 2, 512, 1024, 0);
  RUNK(8, 4, 0, 2, 128, 2, 512, 1024, 0);
  // the gate_up mix: per stage 256 rows of x (32 pieces) + 48 rows of W (6 pieces) -> px 4, pw 1 (ratio 4:1 ~ 256:64)
  RUNK(8, 3, 4, 1, 128, 2, 256, 256, 0);
  RUNK(8, 3, 4, 1, 1024, 2, 256, 256, 0);
  RUNK(8, 3, 4, 1, 0, 2, 256, 256, 0);
  RUNK(8, 4, 2, 2, 128, 2, 256, 512, 0);   // 128 x rows : 128 W rows
  RUNK(8, 4, 2, 2, 1024, 2, 256, 512, 0);
  RUNK(8, 4, 2, 2, 0, 2, 256, 512, 0);
  RUNK(8, 4, 4, 0, 128, 0, 256, 0, 1024);  // x only
  return 0;
}
