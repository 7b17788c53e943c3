// One wave per SIMD issuing a DEPENDENT chain of v_mfma_f32_16x16x32_bf16: accumulate in place (vDst == SrcC) against a chain whose every
// link writes another register quad than it reads (vDst != SrcC, what hipcc emits when it renames accumulators between the six products of
// a bf16 x 3 group).  Cycles per MFMA.
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_chain.hip -o tools/mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  f32x4 x = {0, 0, 0, 0}, y = {0, 0, 0, 0};
  bf16x8 ah, bh;
  for (int q = 0; q < 8; ++q) { ah[q] = (__bf16)(float)threadIdx.x; bh[q] = (__bf16)(1.0f + threadIdx.x); }
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      if (MODE == 0) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(x) : "v"(ah), "v"(bh));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(x) : "v"(ah), "v"(bh));
      } else {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %1" : "=&v"(y) : "v"(x), "v"(ah), "v"(bh));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %1" : "=&v"(x) : "v"(y), "v"(ah), "v"(bh));
      }
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  float r = x[0] + x[1] + x[2] + x[3] + y[0];
  if (r == 12345.678f) out[threadIdx.x] = r;
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 64);
  const int iters = 4000;
  long long h = 0;
  k<0><<<256, 256>>>(out, cyc, iters); (void)hipDeviceSynchronize();
  (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("vDst == SrcC: %.2f cycles per MFMA\n", (double)h / (24.0 * iters));
  k<1><<<256, 256>>>(out, cyc, iters); (void)hipDeviceSynchronize();
  (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("vDst != SrcC: %.2f cycles per MFMA\n", (double)h / (24.0 * iters));
  return 0;
}
