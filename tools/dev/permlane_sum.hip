#include <hip/hip_runtime.h>
__device__ inline float xsum16(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ inline float xsum32(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__global__ void k(float* o) {
  float v = o[threadIdx.x];
  o[threadIdx.x] = xsum32(xsum16(v));
  o[64 + threadIdx.x] = (v + __shfl_xor(v, 16, 64)) + __shfl_xor(v + __shfl_xor(v, 16, 64), 32, 64);
}
int main() {
  float h[128], *d; for (int i = 0; i < 64; ++i) h[i] = 1.0f + i * 0.37f + (i % 7) * 1e-3f;
  hipMalloc(&d, 512); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; ++i) bad += h[i] != h[64 + i];
  printf("mismatches %d  (%g %g | %g %g)\n", bad, h[0], h[64], h[40], h[104]);
  return bad != 0;
}
