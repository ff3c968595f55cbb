// what v_permlane32_swap returns per lane (x = lane id)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *o) {
  unsigned x = threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  o[threadIdx.x] = r[0];
  o[64 + threadIdx.x] = r[1];
  auto r2 = __builtin_amdgcn_permlane32_swap(x, x + 100, false, false);
  o[128 + threadIdx.x] = r2[0];
  o[192 + threadIdx.x] = r2[1];
}
int main() {
  int *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int j = 0; j < 4; ++j) {
    printf("out%d:", j);
    for (int i = 0; i < 64; i += 8) printf(" [%d]=%d", i, h[64 * j + i]);
    printf("\n");
  }
  return 0;
}
