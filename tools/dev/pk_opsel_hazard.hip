// pk_opsel_hazard.hip -- v_pk_mul_f32 with op_sel:[0,1] in one wave of a SIMD while the SIMD's other wave issues <something> (gfx950).
// head_fused_kernel<NBD, u16> lost the term x / (mu + eps) of dP in lanes 48-63 of waves 4-7 when the compiler had formed
//   v_pk_mul_f32 v[154:155], v[206:207], v[166:167] op_sel:[0,1]      (lo = rcp(mu + eps) * x, hi = inv * x)
// out of two scalar products (SLP vectoriser): the LOW half came back wrong whatever wait states stood around the instruction, right with two
// v_mul_f32, and only in the intervals in which the partner waves 0-3 ran the dW phase.  This program looks for the partner instruction.
//   hipcc --offload-arch=gfx950 -O2 tools/dev/pk_opsel_hazard.hip -o /tmp/pk_opsel_hazard && /tmp/pk_opsel_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// VICTIM_HI: 1 = waves 4-7 run the packed products and waves 0-3 the partner instruction; 0 = the other way round
template <int MODE, int VICTIM_HI>
__global__ __launch_bounds__(512, 2) void probe(const float* in, unsigned* bad, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool victim = (wave >= 4) == (VICTIM_HI != 0);
  const int i = blockIdx.x * 512 + tid;
  if (victim) {
    f32x2 ab = {in[4 * i], in[4 * i + 1]}, cd = {in[4 * i + 2], in[4 * i + 3]};
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
      f32x2 r;
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(ab), "v"(cd));
      nbad += (r.x != ab.x * cd.y) | ((r.y != ab.y * cd.y) << 16);
      asm volatile("" : "+v"(ab), "+v"(cd));
    }
    bad[i] = nbad;
  } else {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sink, 0, 0x7FFFFFFF, 0x00020000);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    bf16x8 a8, b8;
    for (int k = 0; k < 8; ++k) { a8[k] = (__bf16)(float)(lane + k); b8[k] = (__bf16)(float)(k + 1); }
    float v = (float)lane;
    for (int it = 0; it < iters; ++it) {
      if (MODE == 1) { acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc, 0, 0, 0); }
      if (MODE == 2) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, i * 4, 0, 0); v += 1.f; }
      if (MODE == 3) { const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + (tid & 255) * 8)); v += (float)t[0]; }
      if (MODE == 4) { *reinterpret_cast<f32x4*>(lds + tid * 16) = acc; acc[0] += 1.f; }
      if (MODE == 5) { v = __builtin_amdgcn_rcpf(v + 1.5f); }
      if (MODE == 6) { v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)); }
      if (MODE == 7) { f32x2 p0 = {acc[0], acc[1]}, p1 = {acc[2], acc[3]}; asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(p1)); acc[0] = p0[0]; acc[1] = p0[1]; }
      asm volatile("" : "+v"(v), "+v"(acc));
    }
    sink[i] = v + acc[0] + acc[1] + acc[2] + acc[3];
  }
}

template <int MODE, int VH>
static void run(const char* what, const float* di, unsigned* dbad, float* dsink, int n, std::vector<unsigned>& h) {
  hipMemset(dbad, 0, (size_t)n * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, VH>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  probe<MODE, VH><<<n / 512, 512, 96 * 1024>>>(di, dbad, dsink, 20000);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), dbad, (size_t)n * 4, hipMemcpyDeviceToHost);
  unsigned long lo = 0, hi = 0, q[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) { lo += h[i] & 0xFFFF; hi += h[i] >> 16; q[(i & 63) >> 4] += h[i] & 0xFFFF; }
  printf("partner: %-34s victim waves %s: low half wrong %lu (lane quarters %lu %lu %lu %lu), high half wrong %lu\n", what, VH ? "4-7" : "0-3", lo, q[0], q[1], q[2], q[3], hi);
}

int main() {
  const int n = 256 * 512;
  std::vector<float> h(4 * (size_t)n);
  for (size_t k = 0; k < h.size(); ++k) h[k] = 1.0f + (float)((k * 2654435761u) % 1000) * 0.001f;
  float *di, *dsink; unsigned* dbad;
  hipMalloc(&di, h.size() * 4); hipMalloc(&dsink, (size_t)n * 4 + 4096); hipMalloc(&dbad, (size_t)n * 4);
  hipMemcpy(di, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<unsigned> hb(n);
#define BOTH(M, W) run<M, 1>(W, di, dbad, dsink, n, hb); run<M, 0>(W, di, dbad, dsink, n, hb)
  BOTH(0, "nothing");
  BOTH(1, "v_mfma_f32_16x16x32_bf16");
  BOTH(2, "buffer_store_dword");
  BOTH(3, "ds_read_b64_tr_b16");
  BOTH(4, "ds_write_b128");
  BOTH(5, "v_rcp_f32");
  BOTH(6, "v_add_f32 dpp quad_perm");
  BOTH(7, "v_pk_add_f32");
  return 0;
}
