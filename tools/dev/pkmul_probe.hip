// pkmul_probe.hip -- v_pk_mul_f32 with op_sel:[0,1] on registers v[206:207] x v[166:167] -> v[154:155] in a kernel of 234+ registers at two waves
// per SIMD (the placement head_fused_kernel<NBD, u16> had it in): does the instruction by itself compute lo = a.lo * b.hi, hi = a.hi * b.hi in every lane of every wave?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512, 2) void probe(const float* in, float* out) {
  const int i = blockIdx.x * 512 + threadIdx.x;
  const float a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
  float lo, hi;
  asm volatile("v_mov_b32 v206, %2\n\tv_mov_b32 v207, %3\n\tv_mov_b32 v166, %4\n\tv_mov_b32 v167, %5\n\tv_mov_b32 v154, 0\n\tv_mov_b32 v155, 0\n\ts_nop 4\n\t"
               "v_pk_mul_f32 v[154:155], v[206:207], v[166:167] op_sel:[0,1]\n\ts_nop 4\n\tv_mov_b32 %0, v154\n\tv_mov_b32 %1, v155"
               : "=v"(lo), "=v"(hi) : "v"(a), "v"(b), "v"(c), "v"(d) : "v154", "v155", "v166", "v167", "v206", "v207", "v233");
  out[2 * i] = lo; out[2 * i + 1] = hi;
}
int main() {
  const int grid = 1024, n = grid * 512;
  std::vector<float> h(4 * (size_t)n), o(2 * (size_t)n);
  for (size_t k = 0; k < h.size(); ++k) h[k] = 1.0f + (float)((k * 2654435761u) % 1000) * 0.001f;
  float *di, *dout; hipMalloc(&di, h.size() * 4); hipMalloc(&dout, o.size() * 4);
  hipMemcpy(di, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  long bad = 0, by_wave[8] = {0}, by_q[4] = {0};
  for (int rep = 0; rep < 20; ++rep) {
    probe<<<grid, 512>>>(di, dout);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i)
      if (o[2 * i] != h[4 * i] * h[4 * i + 3] || o[2 * i + 1] != h[4 * i + 1] * h[4 * i + 3]) { ++bad; ++by_wave[(i & 511) >> 6]; ++by_q[(i & 63) >> 4]; }
  }
  printf("v_pk_mul_f32 op_sel:[0,1] in isolation: %ld wrong of %ld (by wave %ld %ld %ld %ld %ld %ld %ld %ld; by lane quarter %ld %ld %ld %ld)\n", bad, 20L * n, by_wave[0], by_wave[1],
         by_wave[2], by_wave[3], by_wave[4], by_wave[5], by_wave[6], by_wave[7], by_q[0], by_q[1], by_q[2], by_q[3]);
  return 0;
}
