// What does it cost the MAIN stream to hand work to a second queue and to join it?  A chain of short dependent kernels on stream A; after
// kernel 3 a kernel is started on stream B, before kernel 8 stream A waits for it.  Forms: (0) no second queue; (1) hipEventRecord +
// hipStreamWaitEvent both ways (what the background sweep / the data-parallel chain use); (2) hipStreamWriteValue32 / hipStreamWaitValue32
// on a device word; (3) device-side flags: kernel 3 raises a word in its last workgroup, B's kernel (launched beside) spins on it, kernel
// 8's first lane spins on B's word.  Reports us per chain.
// Build: hipcc --offload-arch=gfx950 -O2 tools/xq_sync.hip -o tools/xq_sync
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void short_k(float* p, int n, unsigned* raise, unsigned epoch, const unsigned* wait_for) {
  if (wait_for && threadIdx.x == 0) {   // (bounded spin of one lane per workgroup)
    for (int i = 0; i < 2000000 && __hip_atomic_load(wait_for, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch; ++i) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
  if (raise) {
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence(); __hip_atomic_store(raise, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
  }
}
int main() {
  float *a, *b; unsigned* w;
  hipMalloc(&a, 1 << 22); hipMalloc(&b, 1 << 26); hipMalloc(&w, 256);
  hipMemset(w, 0, 256);
  hipStream_t A, B;
  hipStreamCreateWithFlags(&A, hipStreamNonBlocking); hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
  hipEvent_t e1, e2;
  hipEventCreateWithFlags(&e1, hipEventDisableTiming); hipEventCreateWithFlags(&e2, hipEventDisableTiming);
  const int n = 1 << 16, nb = 1 << 22, reps = 300;
  for (int form = 0; form < 4; ++form) {
    unsigned epoch = 0;
    for (int pass = 0; pass < 2; ++pass) {
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < (pass ? reps : 20); ++r) {
        ++epoch;
        for (int k = 0; k < 12; ++k) {
          const bool hand = k == 3, join = k == 8;
          if (join && form == 1) hipStreamWaitEvent(A, e2, 0);
          if (join && form == 2) hipStreamWaitValue32(A, w + 32, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu);
          hipLaunchKernelGGL(short_k, dim3(n / 256), dim3(256), 0, A, a, n, (hand && form == 3) ? w : nullptr, epoch, (join && form == 3) ? w + 32 : nullptr);
          if (hand) {
            if (form == 1) { hipEventRecord(e1, A); hipStreamWaitEvent(B, e1, 0); }
            if (form == 2) { hipStreamWriteValue32(A, w, epoch, 0); hipStreamWaitValue32(B, w, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu); }
            if (form) hipLaunchKernelGGL(short_k, dim3(64), dim3(256), 0, B, b, 64 * 256, form == 3 ? w + 32 : nullptr, epoch, form == 3 ? w : nullptr);
            if (form == 1) hipEventRecord(e2, B);
            if (form == 2) hipStreamWriteValue32(B, w + 32, epoch, 0);
          }
        }
      }
      hipStreamSynchronize(A); hipStreamSynchronize(B);
      auto t1 = std::chrono::steady_clock::now();
      if (pass) printf("form %d: %.2f us per chain of 12 kernels (%s)\n", form, std::chrono::duration<double, std::micro>(t1 - t0).count() / reps,
                       form == 0 ? "one queue" : form == 1 ? "events both ways" : form == 2 ? "stream write / wait value" : "device-side flags");
    }
  }
  (void)nb;
  return 0;
}
