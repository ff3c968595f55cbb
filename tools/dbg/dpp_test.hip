#include "../../mini-vllm_amd/csrc/common.h"
#include <cstdio>
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int LPT>
__device__ __forceinline__ float token_sum(float x) {
  x += dpp_mov<0xB1>(x);
  x += dpp_mov<0x4E>(x);
  x += dpp_mov<0x141>(x);
  if constexpr (LPT >= 16) x += dpp_mov<0x140>(x);
  if constexpr (LPT >= 32) x += __shfl_xor(x, 16, 64);
  return x;
}
__global__ void k(float* out) {
  int l = threadIdx.x;
  float x = (float)(1 << (l % 16)) ;  // distinct bits per lane in row
  out[l] = token_sum<16>(x);
  out[64 + l] = token_sum<8>(x);
  out[128 + l] = row16_sum(x);
  out[192 + l] = wave_sum((float)l);
  out[256 + l] = dpp_mov<0x141>((float)l);
  out[320 + l] = dpp_mov<0x140>((float)l);
  out[384 + l] = dpp_mov<0xB1>((float)l);
  out[448 + l] = dpp_mov<0x4E>((float)l);
}
int main() {
  float* d; hipMalloc(&d, 512 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"token_sum16", "token_sum8", "row16_sum", "wave_sum", "half_mirror", "row_mirror", "qp_B1", "qp_4E"};
  for (int r = 0; r < 8; ++r) { printf("%s:", names[r]); for (int i = 0; i < 20; ++i) printf(" %g", h[r * 64 + i]); printf("\n"); }
  return 0;
}
