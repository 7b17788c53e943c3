// Do MFMA and VALU work of two waves on one SIMD overlap?  f32-input MFMA (v_mfma_f32_32x32x2_f32) vs bf16 MFMA
// (v_mfma_f32_32x32x16_bf16), beside an independent v_fma_f32 stream in the partner wave.
// Build: hipcc --offload-arch=gfx950 -O2 tools/coexec.hip -o tools/coexec ; run: tools/coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// mode bit 0: waves 0..3 run MFMAs; bit 1: waves 4..7 run VALU FMAs (wave w sits on SIMD w % 4); KIND 0 f32 MFMA, 1 bf16 MFMA
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  if ((wave >> 2) == 0) {   // waves 0..3: one per SIMD; waves 4..7 are their partners
    if (!(mode & 1)) return;
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0};
    const float a = (float)threadIdx.x, b = 1.0f + threadIdx.x;
    bf16x8 ah, bh;
    for (int q = 0; q < 8; ++q) { ah[q] = (__bf16)a; bh[q] = (__bf16)b; }
    for (int i = 0; i < iters; ++i) {
      if (KIND == 0) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc2, 0, 0, 0);
      }
    }
    for (int q = 0; q < 16; ++q) r += acc0[q] + acc1[q] + acc2[q];
  } else {
    if (!(mode & 2)) return;
    float v[8];
    for (int q = 0; q < 8; ++q) v[q] = (float)(threadIdx.x + q);
    const float c = 1.0001f, d = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 6; ++u)      // 48 independent FMAs per iteration = 192 cycles of vector issue per wave
#pragma unroll
        for (int q = 0; q < 8; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(c), "v"(d));   // (plain C gets SLP-packed into v_pk_fma_f32)
    }
    for (int q = 0; q < 8; ++q) r += v[q];
  }
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int KIND>
static float run(float* out, int mode) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<256, 512>>>(out, 2000, mode);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND><<<256, 512>>>(out, 20000, mode);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  // per SIMD: one MFMA wave and one VALU wave (8 waves per workgroup, 1 workgroup per CU)
  printf("20000 iterations; MFMA wave: 3 MFMAs per iteration; VALU wave: 48 v_fma_f32 per iteration\n");
  const float f_m = run<0>(out, 1), f_v = run<0>(out, 2), f_b = run<0>(out, 3);
  printf("f32  MFMA (32x32x2):   MFMA alone %8.1f us   VALU alone %8.1f us   both %8.1f us   (sum %8.1f, max %8.1f)\n", f_m, f_v, f_b, f_m + f_v, f_m > f_v ? f_m : f_v);
  const float b_m = run<1>(out, 1), b_v = run<1>(out, 2), b_b = run<1>(out, 3);
  printf("bf16 MFMA (32x32x16):  MFMA alone %8.1f us   VALU alone %8.1f us   both %8.1f us   (sum %8.1f, max %8.1f)\n", b_m, b_v, b_b, b_m + b_v, b_m > b_v ? b_m : b_v);
  return 0;
}
