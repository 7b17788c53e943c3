// rowseg.hip -- what does HBM deliver when a launch reads a [R rows][C floats] matrix (row stride 240 KB, as dP / W_out at 20 000
// genes x 3 planes) in per-workgroup tiles of R rows x S bytes?  (calibration for the panel / big-K kernels, which read 128-byte
// segments: S = 128.)  Every workgroup of 512 threads reads tiles t = blockIdx.x, + gridDim.x, ...; a tile is R x S bytes, one
// 16-byte load per lane and instruction, lanes along the row first.  Prints GB/s per S and rows-in-flight shape.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d: %s\n",#x,__LINE__,hipGetErrorString(e)); return 1;} } while(0)

template <int S>   // bytes per row and tile
__global__ __launch_bounds__(512) void k_read(const float* __restrict__ a, long ld, int R, int n_tiles, float* out) {
  constexpr int LPR = S / 16;                 // lanes per row segment
  constexpr int RPI = 512 / LPR;              // rows per instruction (whole workgroup)
  const int lr = threadIdx.x / LPR, lc = threadIdx.x % LPR;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const float* base = a + (long)t * (S / 4) + 4 * lc;
#pragma unroll 4
    for (int r = lr; r < R; r += RPI) {
      const float4 v = *reinterpret_cast<const float4*>(base + (long)r * ld);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;   // (never: keeps the loads)
}

int main() {
  const int R = 128; const long C = 61440;   // 128 x 61440 floats = 31.5 MB
  float *a, *out; CK(hipMalloc(&a, (size_t)R * C * 4)); CK(hipMalloc(&out, 4)); CK(hipMemset(a, 0, (size_t)R * C * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {256, 512, 1024}) {
    for (int S : {128, 256, 512, 1024, 2048}) {
      const int n_tiles = (int)(C * 4 / S);
      auto launch = [&]() {
        switch (S) {
          case 128: hipLaunchKernelGGL(k_read<128>, dim3(grid), dim3(512), 0, 0, a, C, R, n_tiles, out); break;
          case 256: hipLaunchKernelGGL(k_read<256>, dim3(grid), dim3(512), 0, 0, a, C, R, n_tiles, out); break;
          case 512: hipLaunchKernelGGL(k_read<512>, dim3(grid), dim3(512), 0, 0, a, C, R, n_tiles, out); break;
          case 1024: hipLaunchKernelGGL(k_read<1024>, dim3(grid), dim3(512), 0, 0, a, C, R, n_tiles, out); break;
          default: hipLaunchKernelGGL(k_read<2048>, dim3(grid), dim3(512), 0, 0, a, C, R, n_tiles, out); break;
        }
      };
      for (int i = 0; i < 3; ++i) launch();
      hipEventRecord(e0, 0);
      const int reps = 20;
      for (int i = 0; i < reps; ++i) launch();
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
      const double us = 1e3 * ms / reps;
      printf("grid %4d  segment %4d B per row: %.1f us per launch, %.2f TB/s (31.5 MB; the 256 MB last-level cache holds it: reads may not reach HBM)\n", grid, S, us, (double)R * C * 4 / us * 1e-6);
      fflush(stdout);
    }
  }
  return 0;
}
