// pk_opsel_forms.hip -- which packed-f32 forms lose a result beside the other wave's MFMAs (follow-up of pk_opsel_hazard.hip; gfx950).
// Victim: waves 0-3 of a 512-thread workgroup issue ONE packed instruction form in a loop and check both halves in every lane;
// partner: waves 4-7 (the SIMDs' second waves) issue back-to-back MFMAs of one shape.
//   hipcc --offload-arch=gfx950 -O2 tools/dev/pk_opsel_forms.hip -o /tmp/pk_opsel_forms && /tmp/pk_opsel_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// FORM(id, asm text, OP 0 mul 1 add 2 fma, lo takes a[S0L] c[S1L], hi takes a[S0H] c[S1H])
#define FORMS(X)                                                                    \
  X(0, "v_pk_mul_f32 %0, %1, %2", 0, 0, 0, 1, 1)                                    \
  X(1, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]", 0, 0, 1, 1, 1)                       \
  X(2, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]", 0, 1, 0, 1, 1)                       \
  X(3, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,1]", 0, 1, 1, 1, 1)                       \
  X(4, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]", 0, 0, 0, 1, 0)                    \
  X(5, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]", 0, 0, 0, 0, 1)                    \
  X(6, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,0]", 0, 0, 0, 0, 0)                    \
  X(7, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]", 0, 0, 1, 1, 0)       \
  X(8, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1]", 1, 0, 1, 1, 1)                       \
  X(9, "v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]", 1, 0, 0, 1, 0)                    \
  X(10, "v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0]", 2, 0, 1, 1, 1)                \
  X(11, "v_pk_fma_f32 %0, %1, %2, %1 op_sel_hi:[1,0,1]", 2, 0, 0, 1, 0)             \
  X(12, "v_pk_fma_f32 %0, %1, %2, %1", 2, 0, 0, 1, 1)                               \
  X(13, "v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,0,0]", 2, 1, 0, 1, 1)                \
  X(14, "v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,1,0]", 2, 1, 1, 1, 1)                \
  X(15, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]", 1, 0, 1, 1, 0)      \
  X(16, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0]", 1, 1, 0, 1, 1)

template <int FORM, int MF>
__global__ __launch_bounds__(512, 2) void probe(const float* in, unsigned* bad, float* sink, float* sample, int iters) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x * 512 + tid;
  if (wave < 4) {
    f32x2 ab = {in[4 * i], in[4 * i + 1]}, cd = {in[4 * i + 2], in[4 * i + 3]};
    unsigned nbad = 0;
    float wlo = 0.f;
    for (int it = 0; it < iters; ++it) {
      f32x2 r;
      float elo = 0.f, ehi = 0.f;
#define X(ID, TXT, OP, S0L, S1L, S0H, S1H)                                                                                   \
      if (FORM == ID) {                                                                                                       \
        asm volatile(TXT : "=v"(r) : "v"(ab), "v"(cd));                                                                       \
        const float al = ab[S0L], cl = cd[S1L], ah = ab[S0H], ch = cd[S1H];                                                   \
        elo = OP == 0 ? al * cl : OP == 1 ? al + cl : __builtin_fmaf(al, cl, ab[0]);                                          \
        ehi = OP == 0 ? ah * ch : OP == 1 ? ah + ch : __builtin_fmaf(ah, ch, ab[1]);                                          \
      }
      FORMS(X)
#undef X
      const bool bl = r.x != elo, bh = r.y != ehi;
      if (bl) wlo = r.x;
      nbad += (unsigned)bl | ((unsigned)bh << 16);
      asm volatile("" : "+v"(ab), "+v"(cd));
    }
    bad[i] = nbad;
    sample[i] = wlo;
  } else {
    bf16x8 a8, b8;
    for (int k = 0; k < 8; ++k) { a8[k] = (__bf16)(float)(lane + k); b8[k] = (__bf16)(float)(k + 1); }
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    f32x16 acc16 = {};
    for (int it = 0; it < iters; ++it) {
      if (MF == 0) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc4, 0, 0, 0);
      if (MF == 1) acc16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc16, 0, 0, 0);
      if (MF == 2) acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)lane, 1.5f, acc4, 0, 0, 0);
      asm volatile("" : "+v"(acc4), "+v"(acc16));
    }
    sink[i] = acc4[0] + acc4[3] + acc16[0] + acc16[15];
  }
}

template <int FORM, int MF>
static void run(const char* form, const char* mf, const float* di, unsigned* dbad, float* dsink, float* dsample, int n, const std::vector<float>& in) {
  std::vector<unsigned> h(n);
  std::vector<float> s(n);
  hipMemset(dbad, 0, (size_t)n * 4);
  probe<FORM, MF><<<n / 512, 512>>>(di, dbad, dsink, dsample, 10000);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), dbad, (size_t)n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(s.data(), dsample, (size_t)n * 4, hipMemcpyDeviceToHost);
  unsigned long lo = 0, hi = 0, q[4] = {0, 0, 0, 0}, qh[4] = {0, 0, 0, 0};
  int first = -1;
  for (int i = 0; i < n; ++i) { lo += h[i] & 0xFFFF; hi += h[i] >> 16; q[(i & 63) >> 4] += h[i] & 0xFFFF; qh[(i & 63) >> 4] += h[i] >> 16; if (first < 0 && (h[i] & 0xFFFF)) first = i; }
  printf("%-52s beside %-26s: low wrong %9lu (quarters %lu %lu %lu %lu)  high wrong %9lu (quarters %lu %lu %lu %lu)", form, mf, lo, q[0], q[1], q[2], q[3], hi, qh[0], qh[1], qh[2], qh[3]);
  if (first >= 0) printf("   e.g. lane %d: got %.9g, a = (%.9g, %.9g), c = (%.9g, %.9g)", first & 63, s[first], in[4 * first], in[4 * first + 1], in[4 * first + 2], in[4 * first + 3]);
  printf("\n");
}

int main() {
  const int n = 256 * 512;
  std::vector<float> h(4 * (size_t)n);
  for (size_t k = 0; k < h.size(); ++k) h[k] = 1.0f + (float)((k * 2654435761u) % 1000) * 0.001f;
  float *di, *dsink, *dsample; unsigned* dbad;
  hipMalloc(&di, h.size() * 4); hipMalloc(&dsink, (size_t)n * 4); hipMalloc(&dsample, (size_t)n * 4); hipMalloc(&dbad, (size_t)n * 4);
  hipMemcpy(di, h.data(), h.size() * 4, hipMemcpyHostToDevice);
#define X(ID, TXT, OP, S0L, S1L, S0H, S1H) run<ID, 0>(TXT, "v_mfma_f32_16x16x32_bf16", di, dbad, dsink, dsample, n, h);
  FORMS(X)
#undef X
  run<1, 1>("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]", "v_mfma_f32_32x32x16_bf16", di, dbad, dsink, dsample, n, h);
  run<1, 2>("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]", "v_mfma_f32_16x16x4_f32", di, dbad, dsink, dsample, n, h);
  run<4, 1>("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]", "v_mfma_f32_32x32x16_bf16", di, dbad, dsink, dsample, n, h);
  run<10, 1>("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0]", "v_mfma_f32_32x32x16_bf16", di, dbad, dsink, dsample, n, h);
  return 0;
}
