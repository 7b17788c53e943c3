// What a launch waits for before its first instruction with an operand: the argument segment.  Three kernels of one wave each, launched back to back on a
// stream with a memset of other memory in between (so nothing is warm by accident), each stamping clock64() at entry and again once an argument FIELD is
// in a register:
//   by_value   a 608-byte struct by value (what the library's launches pass): s_load from the runtime's argument pool
//   by_pointer a pointer to the same struct in ordinary device memory, written once: s_load of the pointer, then of the field
//   preloaded  the same with the pointer PRELOADED into SGPRs by the dispatch (built with -mllvm -amdgpu-kernarg-preload-count=2): the field only
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=2 kernarg_probe.hip -o kernarg_probe   (by_value ignores the option's effect
// on all but its first two dwords)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
struct Args { const float* p[70]; int n[12]; };   // 608 bytes
__global__ void by_value(Args a, long long* out, int slot) {
  const long long t0 = clock64();
  const int v = a.n[11];
  asm volatile("" :: "s"(v));
  const long long t1 = clock64();
  const float x = a.p[69][0];
  asm volatile("" :: "v"(x));
  const long long t2 = clock64();
  if (threadIdx.x == 0) { out[3 * slot] = t1 - t0; out[3 * slot + 1] = t2 - t0; out[3 * slot + 2] = v; }
}
__global__ void by_pointer(const Args* __restrict__ pa, long long* out, int slot) {
  const long long t0 = clock64();
  const int v = pa->n[11];
  asm volatile("" :: "s"(v));
  const long long t1 = clock64();
  const float x = pa->p[69][0];
  asm volatile("" :: "v"(x));
  const long long t2 = clock64();
  if (threadIdx.x == 0) { out[3 * slot] = t1 - t0; out[3 * slot + 1] = t2 - t0; out[3 * slot + 2] = v; }
}
int main() {
  Args h;
  float* data; hipMalloc(&data, 1 << 20); hipMemset(data, 0, 1 << 20);
  for (int i = 0; i < 70; ++i) h.p[i] = data + 64 * i;
  for (int i = 0; i < 12; ++i) h.n[i] = i;
  Args* d; hipMalloc(&d, sizeof(Args)); hipMemcpy(d, &h, sizeof(Args), hipMemcpyHostToDevice);
  long long* out; hipMalloc(&out, 3 * 8 * 4096); hipMemset(out, 0, 3 * 8 * 4096);
  char* other; hipMalloc(&other, 64 << 20);
  const int N = 200;
  for (int mode = 0; mode < 2; ++mode) {
    for (int warm = 0; warm < 2; ++warm) {   // warm = 1: no memset between the launches (the previous launch of the same kernel read the same struct 5 us ago)
      for (int i = 0; i < N; ++i) {
        if (!warm) hipMemsetAsync(other, i, 64 << 20, 0);
        if (mode == 0) hipLaunchKernelGGL(by_value, dim3(1), dim3(64), 0, 0, h, out, i);
        else hipLaunchKernelGGL(by_pointer, dim3(1), dim3(64), 0, 0, d, out, i);
      }
      hipDeviceSynchronize();
      std::vector<long long> r(3 * N);
      hipMemcpy(r.data(), out, 3 * 8 * N, hipMemcpyDeviceToHost);
      std::vector<long long> a, b;
      for (int i = 20; i < N; ++i) { a.push_back(r[3 * i]); b.push_back(r[3 * i + 1]); }
      std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
      printf("%-10s %-28s field in a register after %5lld cycles (median; 10th %lld, 90th %lld), first dependent load back after %5lld\n", mode ? "by_pointer" : "by_value",
             warm ? "back to back" : "a 64 MB memset between", a[a.size() / 2], a[a.size() / 10], a[a.size() * 9 / 10], b[b.size() / 2]);
    }
  }
  return 0;
}
