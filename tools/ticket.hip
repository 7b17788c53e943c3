// ticket.hip -- what does a ONE-WAY "last arriver finishes" epilogue cost next to the launch boundary it would replace?
// (calibration, not product; VERDICT r03 item 5: the C2 step's launch 1 -> launch 2 seam: split-K slabs summed by the BatchNorm launch)
//
//   T output tiles x S K-slices workgroups; workgroup (t, s) writes its slab (P floats) and takes a ticket on tile t's counter; the
//   workgroup that draws S - 1 sums the S slabs in slice order (deterministic) and writes the tile's result.  Not a barrier: nobody waits.
//     form 0  two launches: slabs, then a sum kernel of T workgroups                      (what the step does today)
//     form 1  one launch, plain slab stores + agent-scope release fence (one lane) -> relaxed ticket -> acquire fence -> plain loads
//     form 2  one launch, write-through (sc1) slab stores drained by s_waitcnt -> relaxed ticket -> sc1 loads   (no fence at all)
//   Each form is timed as a stream of dependent repetitions (the consumer of repetition r reads what repetition r - 1 wrote) and the sum
//   is checked on the host.
// build: hipcc --offload-arch=gfx950 -O3 tools/ticket.hip -o tools/ticket ; run: tools/ticket
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
__device__ inline void st_sc1(f4* p, f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
typedef unsigned u4 __attribute__((ext_vector_type(4)));
// sc1 loads the compiler can count: raw buffer loads with aux = 16 (cdna_hip_programming.md: "raw_buffer_load_b128(..., 16)")
__device__ inline f4 ld_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16));
}

// slab of (t, s): value depends on (rep, t, s, i) so a stale read is caught by the host check
__device__ inline float val(int rep, int t, int s, int i) { return (float)((rep * 7 + t * 3 + s * 5 + i) % 97) * 0.25f; }

template <int FORM>
__global__ __launch_bounds__(256) void k_slabs(float* slabs, float* out, unsigned* ctr, int S, int P, int rep, const float* prev) {
  __shared__ int last_s;
  const int t = blockIdx.x / S, s = blockIdx.x % S;
  const float bias = prev ? prev[t] * 0.f : 0.f;   // (a dependence on the previous repetition's output: the launches form a chain)
  f4* mine = reinterpret_cast<f4*>(slabs + ((size_t)t * S + s) * P);
  for (int i = threadIdx.x; i < P / 4; i += 256) {
    f4 v = {val(rep, t, s, 4 * i) + bias, val(rep, t, s, 4 * i + 1), val(rep, t, s, 4 * i + 2), val(rep, t, s, 4 * i + 3)};
    if (FORM == 2) st_sc1(mine + i, v); else mine[i] = v;
  }
  if (FORM == 0) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (FORM == 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    const unsigned got = __hip_atomic_fetch_add(ctr + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_s = (got % (unsigned)S) == (unsigned)(S - 1);   // (the counter only grows: no reset between repetitions)
    if (last_s && FORM == 1) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  }
  __syncthreads();
  if (!last_s) return;
  // the last arriver: sum of the S slabs in slice order, 8 slabs' loads in flight per thread
  const f4* base = reinterpret_cast<const f4*>(slabs + (size_t)t * S * P);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4*>(base), 0, S * P * 4, 0x00020000);
  for (int i = threadIdx.x; i < P / 4; i += 256) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < S; s0 += 8) {
      f4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (s0 + u < S) v[u] = FORM == 2 ? ld_sc1(rs, (unsigned)(((size_t)(s0 + u) * (P / 4) + i) * 16)) : base[(size_t)(s0 + u) * (P / 4) + i];
        else v[u] = f4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    reinterpret_cast<f4*>(out + (size_t)t * P)[i] = acc;
  }
}

__global__ __launch_bounds__(256) void k_sum(const float* slabs, float* out, int S, int P) {
  const int t = blockIdx.x;
  const f4* base = reinterpret_cast<const f4*>(slabs + (size_t)t * S * P);
  for (int i = threadIdx.x; i < P / 4; i += 256) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < S; s0 += 8) {
      f4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = s0 + u < S ? base[(size_t)(s0 + u) * (P / 4) + i] : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    reinterpret_cast<f4*>(out + (size_t)t * P)[i] = acc;
  }
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const size_t maxf = (size_t)64 * 64 * 4096;
  float *slabs, *out; unsigned* ctr;
  CK(hipMalloc(&slabs, maxf * 4)); CK(hipMalloc(&out, (size_t)64 * 4096 * 4 * 2)); CK(hipMalloc(&ctr, 64 * 4));
  const int reps = 200;
  struct Cfg { int T, S, P; const char* what; };
  // C2 seam: [128][128] output as 16 slabs.  Tile = 128 rows x 8 / 32 / 128 columns -> (T, S, floats per slab)
  const Cfg cfgs[] = {{16, 16, 1024, "C2 launch 1 -> 2, tile 128 x 8 columns (what a BatchNorm workgroup owns): 4 KB slabs"},
                      {4, 16, 4096, "C2, tile 128 x 32 columns: 16 KB slabs"},
                      {4, 64, 4096, "C2, tile 128 x 32 columns, 64 slices (256 workgroups)"},
                      {16, 4, 1024, "4 slices of 4 KB"},
                      {64, 4, 1024, "64 tiles x 4 slices of 4 KB (256 workgroups)"},
                      {1, 64, 4096, "one tile, 64 slices of 16 KB"}};
  for (const Cfg& c : cfgs) {
    double us[3]; int bad[3] = {0, 0, 0};
    for (int form = 0; form < 3; ++form) {
      CK(hipMemset(ctr, 0, 64 * 4));
      auto run = [&](int rep) {
        float* o = out + (size_t)(rep & 1) * 64 * 4096;
        const float* prev = rep ? out + (size_t)((rep - 1) & 1) * 64 * 4096 : nullptr;
        if (form == 0) {
          hipLaunchKernelGGL(k_slabs<0>, dim3(c.T * c.S), dim3(256), 0, st, slabs, o, ctr, c.S, c.P, rep, prev);
          hipLaunchKernelGGL(k_sum, dim3(c.T), dim3(256), 0, st, slabs, o, c.S, c.P);
        } else if (form == 1) hipLaunchKernelGGL(k_slabs<1>, dim3(c.T * c.S), dim3(256), 0, st, slabs, o, ctr, c.S, c.P, rep, prev);
        else hipLaunchKernelGGL(k_slabs<2>, dim3(c.T * c.S), dim3(256), 0, st, slabs, o, ctr, c.S, c.P, rep, prev);
      };
      for (int r = 0; r < 10; ++r) run(r);
      CK(hipStreamSynchronize(st));
      auto t0 = std::chrono::high_resolution_clock::now();
      for (int r = 10; r < 10 + reps; ++r) run(r);
      CK(hipStreamSynchronize(st));
      us[form] = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
      const int last = 10 + reps - 1;
      std::vector<float> h((size_t)c.T * c.P);
      CK(hipMemcpy(h.data(), out + (size_t)(last & 1) * 64 * 4096, h.size() * 4, hipMemcpyDeviceToHost));
      for (int t = 0; t < c.T; ++t)
        for (int i = 0; i < c.P; ++i) {
          float ref = 0.f;
          for (int s = 0; s < c.S; ++s) ref += (float)((last * 7 + t * 3 + s * 5 + i) % 97) * 0.25f;
          if (h[(size_t)t * c.P + i] != ref) ++bad[form];
        }
    }
    printf("%-92s T %2d S %2d: two launches %6.2f us | ticket, plain stores + release / acquire %6.2f us | ticket, sc1 stores + sc1 loads %6.2f us   wrong %d %d %d\n",
           c.what, c.T, c.S, us[0], us[1], us[2], bad[0], bad[1], bad[2]);
    fflush(stdout);
  }
  return 0;
}
