// fuse_probe.hip -- what a launch boundary between two dependent small kernels costs on MI355X, against the same two phases in ONE launch
// whose second phase's workgroups wait for a counter the first phase's workgroups raise (blocks are dispatched in index order: the waiting
// blocks start only after every producer block has started, so the wait cannot starve the producers).
//   hipcc --offload-arch=gfx950 -O2 tools/dev/fuse_probe.hip -o /tmp/fuse_probe && /tmp/fuse_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ inline void phase_a(float* buf, int n, int bid, int nb, float s) {   // writes n floats
  for (int i = bid * 256 + threadIdx.x; i < n; i += nb * 256) buf[i] = s + (float)i;
}
__device__ inline void phase_b(const float* buf, float* out, int n, int bid, int nb) {   // reads them all, one number per block
  float t = 0.f;
  for (int i = bid * 256 + threadIdx.x; i < n; i += nb * 256) t += buf[i];
  for (int o = 32; o; o >>= 1) t += __shfl_xor(t, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out + bid, t);
}
__global__ __launch_bounds__(256) void ka(float* buf, int n, float s) { phase_a(buf, n, blockIdx.x, gridDim.x, s); }
__global__ __launch_bounds__(256) void kb(const float* buf, float* out, int n) { phase_b(buf, out, n, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(256) void kab(float* buf, float* out, int n, float s, int na, unsigned* ctr) {
  const int bid = blockIdx.x;
  if (bid < na) {
    phase_a(buf, n, bid, na, s);
    __threadfence();   // this block's writes are visible device-wide before its count
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    const int nb = gridDim.x - na;
    if (threadIdx.x == 0) {
      while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)na) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    phase_b(buf, out, n, bid - na, nb);
    __syncthreads();
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nb - 1) {   // the last consumer re-arms both
      __hip_atomic_store(ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ctr, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ... the same with the exchanged data moved by device-scope accesses (write-through stores, loads that do not trust the XCD's L2) and
// no release / acquire fence: what a fence costs on a part with eight non-coherent L2s is the write-back / invalidate of a whole L2
__global__ __launch_bounds__(256) void kab_wt(float* buf, float* out, int n, float s, int na, unsigned* ctr) {
  const int bid = blockIdx.x;
  if (bid < na) {
    for (int i = bid * 256 + threadIdx.x; i < n; i += na * 256) __hip_atomic_store(buf + i, s + (float)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);   // (vmcnt(0): the stores have been acknowledged)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    const int nb = gridDim.x - na;
    if (threadIdx.x == 0) {
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)na) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    float t = 0.f;
    for (int i = (bid - na) * 256 + threadIdx.x; i < n; i += nb * 256) t += __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int o = 32; o; o >>= 1) t += __shfl_xor(t, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + (bid - na), t);
    __syncthreads();
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nb - 1) {
      __hip_atomic_store(ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int main() {
  const int reps = 2000;
  float *buf, *out; unsigned* ctr;
  CHECK(hipMalloc(&buf, 64 << 20)); CHECK(hipMalloc(&out, 4096 * 4)); CHECK(hipMalloc(&ctr, 64)); CHECK(hipMemset(ctr, 0, 64)); CHECK(hipMemset(out, 0, 4096 * 4));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int cases[][3] = {{256, 64, 64 << 10}, {256, 64, 256 << 10}, {256, 64, 2 << 20}, {608, 73, 256 << 10}, {64, 128, 256 << 10}, {1024, 256, 8 << 20}};
  for (auto& c : cases) {
    const int na = c[0], nb = c[1], n = c[2];
    float ms2 = 0.f, ms1 = 0.f, ms0 = 0.f;
    for (int warm = 0; warm < 2; ++warm) {
      CHECK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) { hipLaunchKernelGGL(ka, dim3(na), dim3(256), 0, st, buf, n, (float)r); hipLaunchKernelGGL(kb, dim3(nb), dim3(256), 0, st, buf, out, n); }
      CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms2, e0, e1));
      CHECK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kab, dim3(na + nb), dim3(256), 0, st, buf, out, n, (float)r, na, ctr);
      CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms1, e0, e1));
      CHECK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kab_wt, dim3(na + nb), dim3(256), 0, st, buf, out, n, (float)r, na, ctr);
      CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms0, e0, e1));
    }
    // correctness of the fused form: out[b] accumulated the same sums twice per repetition in both forms -> compare one fresh pass of each
    std::vector<float> h2(nb), h1(nb);
    CHECK(hipMemset(out, 0, 4096 * 4));
    hipLaunchKernelGGL(ka, dim3(na), dim3(256), 0, st, buf, n, 3.f); hipLaunchKernelGGL(kb, dim3(nb), dim3(256), 0, st, buf, out, n);
    CHECK(hipStreamSynchronize(st)); CHECK(hipMemcpy(h2.data(), out, nb * 4, hipMemcpyDeviceToHost));
    int bad = 0, bad0 = 0;
    for (int t = 0; t < 200; ++t) {
      CHECK(hipMemsetAsync(out, 0, 4096 * 4, st));
      hipLaunchKernelGGL(kab, dim3(na + nb), dim3(256), 0, st, buf, out, n, 3.f, na, ctr);
      CHECK(hipStreamSynchronize(st)); CHECK(hipMemcpy(h1.data(), out, nb * 4, hipMemcpyDeviceToHost));
      for (int b = 0; b < nb; ++b) bad += (fabsf(h1[b] - h2[b]) > 1e-3f * fabsf(h2[b])) ? 1 : 0;
      CHECK(hipMemsetAsync(out, 0, 4096 * 4, st));
      hipLaunchKernelGGL(kab_wt, dim3(na + nb), dim3(256), 0, st, buf, out, n, 3.f, na, ctr);
      CHECK(hipStreamSynchronize(st)); CHECK(hipMemcpy(h1.data(), out, nb * 4, hipMemcpyDeviceToHost));
      for (int b = 0; b < nb; ++b) bad0 += (fabsf(h1[b] - h2[b]) > 1e-3f * fabsf(h2[b])) ? 1 : 0;
    }
    printf("producers %4d  consumers %3d  %7.2f KB:  two launches %.2f us per pair,  one launch with a counter and fences %.2f us (wrong: %d),  with device-scope accesses and no fence %.2f us (wrong: %d)\n", na, nb, n * 4 / 1024.0,
           1000.0 * ms2 / reps, 1000.0 * ms1 / reps, bad, 1000.0 * ms0 / reps, bad0);
  }
  return 0;
}
