// clk.hip -- in-kernel shader clock (s_memtime vs s_memrealtime) for 1 busy CU vs all CUs busy.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned long long* out, int iters) {
  float a = threadIdx.x * 1e-3f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) { a = a * 1.0001f + 0.5f; a = a * 1.0001f + 0.25f; a = a * 1.0001f + 0.125f; a = a * 1.0001f + 0.0625f; }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
  if (a == 123.f) out[0] = 0;
}
int main() {
  unsigned long long *d, h[4096];
  hipMalloc(&d, sizeof(h));
  for (int grid : {1, 1, 256, 2048}) for (int th : {64, 1024}) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(grid), dim3(th), 0, 0, d, 20000);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
    printf("grid %4d x %4d threads: cycles %llu, realtime ticks %llu -> %.2f GHz (%.1f us, %.2f cycles per fma)\n", grid, th, h[0], h[1],
           (double)h[0] / h[1] * 0.1, h[1] / 100.0, (double)h[0] / 80000.0);
  }
  // sustained: 200 back-to-back launches of the 1-WG kernel, then read the clock again
  for (int rep = 0; rep < 200; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, d, 20000);
  hipDeviceSynchronize();
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("after 200 launches (1 WG): %.2f GHz\n", (double)h[0] / h[1] * 0.1);
  return 0;
}
