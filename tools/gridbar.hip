// gridbar.hip -- what does a device-wide barrier inside ONE launch cost next to a kernel boundary?  (calibration, not product)
//   * k_phases: NP phases separated by a counter barrier (all-thread release fence, one agent-scope atomic per workgroup,
//     bounded spin, acquire fence); every phase writes a value per workgroup and reads another workgroup's value of the
//     previous phase, so a barrier that lets stale data through is caught;
//   * the same NP phases as NP dependent launches of a one-phase kernel.
// build: hipcc --offload-arch=gfx950 -O3 tools/gridbar.hip -o tools/gridbar ; run: tools/gridbar
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d: %s\n",#x,__LINE__,hipGetErrorString(e)); return 1;} } while(0)

// V = 0: every thread runs the agent-scope release and acquire fences (what __threadfence() on both sides amounts to);
// V = 1: every thread only waits for its own stores (workgroup-scope release), ONE thread writes the XCD's L2 back, arrives,
//        spins, and invalidates; the workgroup barrier publishes that to the other waves (they share the CU's L1, and the L2
//        maintenance acts on the whole XCD's cache, not on one wave's lines)
template <int V>
__device__ inline void grid_barrier(unsigned* ctr, unsigned target, int* err) {
  if (V == 0) __threadfence(); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (V == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > 200000000LL) { *err = 1; break; }   // 100 MHz clock: 2 s, then give up
    }
    if (V == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (V == 0) __threadfence();
}

template <int V>
__global__ void k_phases(unsigned* ctr, unsigned base, int* buf, int np, int payload, int* err, int* bad) {
  const int nb = gridDim.x, b = blockIdx.x;
  unsigned target = base;   // (the counter only ever grows: no reset launch between steps)
  for (int p = 0; p < np; ++p) {
    // the phase's "work": payload ints per workgroup, each depending on another workgroup's previous-phase value
    for (int i = threadIdx.x; i < payload; i += blockDim.x) {
      int prev = 0;
      if (p > 0) {
        prev = buf[((p - 1) & 1) * nb * payload + ((b + 37) % nb) * payload + i];
        if (prev != (p - 1) * 100000 + ((b + 37) % nb) + i) atomicAdd(bad, 1);
      }
      buf[(p & 1) * nb * payload + b * payload + i] = p * 100000 + b + i;
    }
    target += nb;
    grid_barrier<V>(ctr, target, err);
  }
}
__global__ void k_one(int* buf, int p, int payload, int* bad) {
  const int nb = gridDim.x, b = blockIdx.x;
  for (int i = threadIdx.x; i < payload; i += blockDim.x) {
    if (p > 0) {
      const int prev = buf[((p - 1) & 1) * nb * payload + ((b + 37) % nb) * payload + i];
      if (prev != (p - 1) * 100000 + ((b + 37) % nb) + i) atomicAdd(bad, 1);
    }
    buf[(p & 1) * nb * payload + b * payload + i] = p * 100000 + b + i;
  }
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  unsigned* ctr; int *buf, *err, *bad;
  const int maxb = 1024, maxpay = 4096;
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&buf, 2 * maxb * maxpay * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&bad, 4));
  CK(hipMemset(err, 0, 4)); CK(hipMemset(bad, 0, 4));
  const int np = 11, reps = 200;
  unsigned base = 0;
  CK(hipMemset(ctr, 0, 4));
  for (int threads : {256, 512}) for (int nb : {64, 256, 512}) for (int payload : {64, 4096}) {
    if (nb * threads > 256 * 2048) continue;
    // barrier form
    double us_bar[2];
    for (int v = 0; v < 2; ++v) {
      auto run_bar = [&]() {
        if (v == 0) hipLaunchKernelGGL(k_phases<0>, dim3(nb), dim3(threads), 0, st, ctr, base, buf, np, payload, err, bad);
        else hipLaunchKernelGGL(k_phases<1>, dim3(nb), dim3(threads), 0, st, ctr, base, buf, np, payload, err, bad);
        base += (unsigned)(np * nb);
      };
      for (int i = 0; i < 5; ++i) run_bar();
      CK(hipStreamSynchronize(st));
      auto t0 = std::chrono::high_resolution_clock::now();
      for (int r = 0; r < reps; ++r) run_bar();
      CK(hipStreamSynchronize(st));
      us_bar[v] = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
    }
    // launch form
    auto run_l = [&]() { for (int p = 0; p < np; ++p) hipLaunchKernelGGL(k_one, dim3(nb), dim3(threads), 0, st, buf, p, payload, bad); };
    for (int i = 0; i < 5; ++i) run_l();
    CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < reps; ++r) run_l();
    CK(hipStreamSynchronize(st));
    const double us_l = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
    int herr = 0, hbad = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    printf("threads %d workgroups %d payload %d ints: per phase, all-thread fences %.2f us, one-wave fences %.2f us | per launch %.2f us   err %d bad %d\n",
           threads, nb, payload, us_bar[0] / np, us_bar[1] / np, us_l / np, herr, hbad);
    fflush(stdout);
  }
  return 0;
}
