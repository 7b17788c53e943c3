// Two waves on one SIMD: one issues v_mfma_f32_16x16x32_bf16 (a dependent chain, as the fused head's six-product groups), its partner a
// vector stream (v_fma_f32 / v_exp_f32 mix).  How long does each take alone, and beside the other?  (round 5: is a de-phased fused head --
// one wave in a matrix phase, its partner in the likelihood -- worth building?)
// Build: hipcc --offload-arch=gfx950 -O2 tools/coexec16.hip -o tools/coexec16 ; run: tools/coexec16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// mode bit 0: waves 0..3 run MFMAs; bit 1: waves 4..7 run the vector stream; bit 2: roles swapped (waves 4..7 MFMA, 0..3 vector);
// chains: accumulators the MFMA stream rotates over (1 = one dependent chain); prio: s_setprio of the MFMA wave
template <int CHAINS, int TRANS>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters, int mode, int prio) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool first = (wave >> 2) == 0;
  const bool mf = (mode & 4) ? !first : first;
  float r = 0.f;
  const long long t0 = clock64();
  if (mf) {
    if (!(mode & 1)) return;
    if (prio) __builtin_amdgcn_s_setprio(1);
    f32x4 acc[4] = {{0}, {0}, {0}, {0}};
    bf16x8 ah, bh;
    for (int q = 0; q < 8; ++q) { ah[q] = (__bf16)(float)threadIdx.x; bh[q] = (__bf16)(1.0f + threadIdx.x); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 24; ++u) acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[u % CHAINS], 0, 0, 0);   // 24 x 16 = 384 cycles
    }
    for (int c = 0; c < 4; ++c) for (int q = 0; q < 4; ++q) r += acc[c][q];
  } else {
    if (!(mode & 2)) return;
    float v[8];
    for (int q = 0; q < 8; ++q) v[q] = (float)(threadIdx.x + q) * 1e-3f;
    const float c = 1.0001f, d = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 12; ++u)      // 96 vector instructions per iteration = 384 cycles of issue (more with transcendentals)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (TRANS && (u % 6) == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(v[q]));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(c), "v"(d));
        }
    }
    for (int q = 0; q < 8; ++q) r += v[q];
  }
  const long long t1 = clock64();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int CHAINS, int TRANS>
static void run(float* out, long long* cyc, int mode, int prio, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  k<CHAINS, TRANS><<<256, 512>>>(out, cyc, 200, mode, prio);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<CHAINS, TRANS><<<256, 512>>>(out, cyc, iters, mode, prio);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-44s chains %d trans %d prio %d: %8.1f us   cycles per iteration: wave0 %6.1f  wave4 %6.1f\n", what, CHAINS, TRANS, prio, ms * 1e3f,
         (double)h[0] / iters, (double)h[4] / iters);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
  hipMemset(cyc, 0, 64);
#define ALL(C, T)                                                            \
  run<C, T>(out, cyc, 1, 0, "MFMA alone (waves 0-3)");                       \
  run<C, T>(out, cyc, 2, 0, "vector alone (waves 4-7)");                     \
  run<C, T>(out, cyc, 3, 0, "MFMA (0-3) beside vector (4-7)");               \
  run<C, T>(out, cyc, 3, 1, "MFMA (0-3) beside vector (4-7), prio");         \
  run<C, T>(out, cyc, 7, 0, "MFMA (4-7) beside vector (0-3)");               \
  run<C, T>(out, cyc, 7, 1, "MFMA (4-7) beside vector (0-3), prio");
  ALL(1, 0) ALL(3, 0) ALL(1, 1)
  return 0;
}
