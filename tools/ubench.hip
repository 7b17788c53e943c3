// ubench.hip -- calibration microbenchmarks for the step design (not part of the product):
// per-kernel cost of dependent launches (eager / graph), dependent-load chains, stream copy rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d: %s\n",#x,__LINE__,hipGetErrorString(e)); return 1;} } while(0)

__global__ void k_empty() {}
__global__ void k_touch(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void k_chain(const int* idx, int* out, int n) {  // n dependent loads by one lane
  if (threadIdx.x == 0 && blockIdx.x == 0) { int j = 0; for (int i = 0; i < n; ++i) j = idx[j]; out[0] = j; }
}
__global__ void k_copy(const float4* a, float4* b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_copy1(const float* a, float* b, size_t n) {  // 4 B per lane, the loss kernel's access width
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_fma(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; a = a * b + 0.25f; a = a * b + 0.125f; a = a * b + 0.0625f; }
  if (a == 123.f) out[0] = a;
}

template <typename F> double time_graph(hipStream_t st, int nk, int reps, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < nk; ++i) launch();
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return us / reps / nk;
}
template <typename F> double time_eager(hipStream_t st, int nk, int reps, F launch) {
  for (int i = 0; i < nk; ++i) launch();
  hipStreamSynchronize(st);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int r = 0; r < reps; ++r) for (int i = 0; i < nk; ++i) launch();
  hipStreamSynchronize(st);
  double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
  return us / reps / nk;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  if (getenv("UBENCH_CALIB")) {  // one launch each, for PMC byte-counter calibration: 64 MiB read + 64 MiB written
    size_t sz = (size_t)64 << 20;
    float4 *a, *b; CK(hipMalloc(&a, sz)); CK(hipMalloc(&b, sz)); CK(hipMemset(a, 1, sz)); CK(hipMemset(b, 0, sz));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, st, a, b, sz / 16);
    hipLaunchKernelGGL(k_copy1, dim3(2048), dim3(256), 0, st, (const float*)a, (float*)b, sz / 4);
    CK(hipStreamSynchronize(st));
    return 0;
  }
  int *d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
  printf("per-kernel cost, dependent launches on one stream (us):\n");
  for (int wg : {1, 16, 256, 2048}) {
    double g = time_graph(st, 24, 200, [&] { hipLaunchKernelGGL(k_empty, dim3(wg), dim3(256), 0, st); });
    double e = time_eager(st, 24, 200, [&] { hipLaunchKernelGGL(k_empty, dim3(wg), dim3(256), 0, st); });
    printf("  empty  %5d WG: graph %.2f  eager %.2f\n", wg, g, e);
  }
  {
    double g = time_graph(st, 24, 200, [&] { hipLaunchKernelGGL(k_touch, dim3(1), dim3(256), 0, st, d); });
    printf("  touch      1 WG: graph %.2f\n", g);
  }
  std::vector<int> h(1 << 18);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (int)((i * 7919 + 12345) % h.size());
  CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  int* out; CK(hipMalloc(&out, 64));
  for (int n : {1, 4, 16, 64}) {
    double g = time_graph(st, 8, 200, [&] { hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, st, d, out, n); });
    printf("  chain of %2d dependent loads: graph %.2f us/kernel\n", n, g);
  }
  for (int it : {256, 1024, 4096}) {
    double g = time_graph(st, 8, 100, [&] { hipLaunchKernelGGL(k_fma, dim3(1), dim3(64), 0, st, (float*)out, it); });
    printf("  %5d x4 dependent fma (1 wave): graph %.2f us/kernel -> %.2f ns per fma\n", it, g, (g - 2.0) * 1e3 / (it * 4));
  }
  size_t bytes = (size_t)1 << 30;
  float4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
  for (size_t sz : {(size_t)4 << 20, (size_t)32 << 20, (size_t)256 << 20, (size_t)1 << 30}) {
    double g = time_graph(st, 4, 20, [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, st, a, b, sz / 16); });
    printf("  copy %5zu MiB: %.2f us -> %.1f GB/s (read+write)\n", sz >> 20, g, 2.0 * sz / g / 1e3);
  }
  return 0;
}
