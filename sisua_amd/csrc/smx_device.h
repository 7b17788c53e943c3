// smx_device.h -- device-side helpers shared by the kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smx {

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. SC'11).  Same function as
// oracle/sisua_oracle.py:philox4x32_10; counter = (column_block, cell_id, step,
// stream | sample<<8), key = seed.  Integer-exact, so dropout masks are
// bit-identical to the oracle's.
// ---------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };

__host__ __device__ inline U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                            uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

// Parameters of one noise stream for one launch.
struct NoiseKey {
  uint32_t k0, k1;   // seed lo/hi
  uint32_t step;     // optimiser step (counter word 2) when step_ptr == nullptr
  uint32_t stream;   // stream | sample << 8 (counter word 3)
  const uint32_t* step_ptr;  // device-resident step counter (graph replay); overrides `step`
};

__device__ inline U4 philox_block(const NoiseKey& nk, uint32_t cell_id, uint32_t col_block) {
  const uint32_t step = nk.step_ptr ? *nk.step_ptr : nk.step;
  return philox4x32_10(col_block, cell_id, step, nk.stream, nk.k0, nk.k1);
}

__device__ inline float u24(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-08f; }  // 2^-24

// a zero quad as a VALUE: `ok ? *p : z4` with a named z4 is an lvalue conditional, which gives z4 an address -- a scratch
// slot per lane, and every launch of a kernel with a private segment pays for its set-up
__device__ inline float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// Inverted-dropout multipliers for the 4 columns of one Philox block.
__device__ inline float4 dropout_mult4(const U4& w, float p, float scale) {
  float4 m;
  m.x = (u24(w.x) >= p) ? scale : 0.f;
  m.y = (u24(w.y) >= p) ? scale : 0.f;
  m.z = (u24(w.z) >= p) ? scale : 0.f;
  m.w = (u24(w.w) >= p) ? scale : 0.f;
  return m;
}

__device__ inline float dropout_mult1(const U4& w, int lane4, float p, float scale) {
  const uint32_t v = lane4 == 0 ? w.x : lane4 == 1 ? w.y : lane4 == 2 ? w.z : w.w;
  return (u24(v) >= p) ? scale : 0.f;
}

// Four standard normals from one block (Box-Muller on word pairs).
__device__ inline float4 normal4(const U4& w) {
  const float u1a = ((float)(w.x >> 8) + 1.0f) * 5.9604644775390625e-08f;
  const float u2a = u24(w.y);
  const float u1b = ((float)(w.z >> 8) + 1.0f) * 5.9604644775390625e-08f;
  const float u2b = u24(w.w);
  // fast forms: v_log_f32 and the revolution-based v_sin_f32 / v_cos_f32 (sin(2 pi u) directly);
  // absolute error of a draw ~1e-6, far inside the 1e-4 parity budget
  // -2 ln u = -2 ln2 log2 u: one v_log_f32 and one v_sqrt_f32 per pair (u >= 2^-24: never denormal)
  const float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1a));
  const float rb = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1b));
  const float sa = __builtin_amdgcn_sinf(u2a), ca = __builtin_amdgcn_cosf(u2a);
  const float sb = __builtin_amdgcn_sinf(u2b), cb = __builtin_amdgcn_cosf(u2b);
  return float4{ra * ca, ra * sa, rb * cb, rb * sb};
}

// ---------------------------------------------------------------------------
// Elementary functions (fp32)
// ---------------------------------------------------------------------------
#define SMX_SOFTPLUS_INV_1 0.54132485461291810f  // log(e - 1)

// Fast transcendental forms: ONE hardware instruction each (v_exp_f32 / v_log_f32 / v_rcp_f32 / v_sqrt_f32, ~1 ulp of
// the base-2 function) -- the kernels are parity-bound at 1e-4, not at 1 ulp.  The HIP spellings do NOT give
// these: __logf expands to the denormal-safe log (v_ldexp + v_log + a 4-term correction, 14 instructions),
// __frcp_rn to the correctly rounded division (v_div_scale / v_div_fmas / v_div_fixup, 10 instructions); together
// they were half of the likelihood kernel's ~500 vector instructions per element.  Every argument here is a
// normal float (1 + e, rising factorials >= 1e-30, mu + 1e-8, ...; the Stirling shift keeps its numerator and
// denominator apart for that reason), so the denormal paths bought nothing.
__device__ inline float fexp(float x) { return __expf(x); }                                        // v_mul + v_exp
__device__ inline float flog(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }   // v_log (log2) + v_mul
__device__ inline float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// log(1 + e) for e >= 0 without losing e below 2^-24: series under 1/32, v_log above.
__device__ inline float log1p_small(float e) {
  const float ser = e * (1.0f - e * (0.5f - e * (0.33333334f - e * (0.25f - e * 0.2f))));
  return e < 0.03125f ? ser : flog(1.0f + e);
}

// softplus(x) and sigmoid(x) from ONE exponential.
struct SpSg { float sp, sg; };
__device__ inline SpSg softplus_sigmoid(float x) {
  const float e = fexp(-fabsf(x));
  const float inv = frcp(1.0f + e);
  SpSg o;
  o.sp = fmaxf(x, 0.f) + log1p_small(e);
  o.sg = x >= 0.f ? inv : e * inv;
  return o;
}
__device__ inline float softplusf(float x) { return fmaxf(x, 0.f) + log1p_small(fexp(-fabsf(x))); }
__device__ inline float sigmoidf(float x) {
  const float e = fexp(-fabsf(x));
  const float s = frcp(1.0f + e);
  return x >= 0.f ? s : e * s;
}
// log1p for count inputs (x >= 0; 1 + x is exact for integer counts < 2^24)
__device__ inline float log1p_count(float x) { return flog(1.0f + x); }

// lgamma(x + r) - lgamma(r) and digamma(x + r) - digamma(r) for x >= 0, r > 0, in
// fp32 without the cancellation of two separate lgamma calls:
//  * x a small integer: log of the rising factorial r (r+1) ... (r+x-1) and its
//    logarithmic derivative P'/P (exact recurrences);
//  * otherwise both arguments are shifted above 8 by the same n and the
//    Stirling series is differenced analytically:
//      D = x ln(x+r') + (r'-1/2) log1p(x/r') - x + c(x+r') - c(r') + ln prod (r+i)/(x+r+i).
struct LgDg { float lg, dg; };

__device__ inline float stirling_corr(float z) {  // 1/(12z) - 1/(360z^3) + 1/(1260z^5)
  const float iz = frcp(z), iz2 = iz * iz;
  return iz * (0.083333333333f + iz2 * (-0.0027777777778f + iz2 * 0.00079365079365f));
}
__device__ inline float digamma_corr(float z) {  // -1/(2z) - 1/(12z^2) + 1/(120z^4) - 1/(252z^6)
  const float iz = frcp(z), iz2 = iz * iz;
  return -0.5f * iz - iz2 * (0.083333333333f - iz2 * (0.0083333333333f - iz2 * 0.003968253968f));
}

__device__ inline LgDg lgamma_digamma_diff(float x, float r) {
  LgDg o;
  r = fminf(fmaxf(r, 1e-30f), 1e30f);
  // Common case (99 % of single-cell counts): x an integer in 0..8.  Branch-free: eight predicated steps of
  // the rising-factorial recurrence P <- P (r+i), P' <- P' (r+i) + P; x = 0 leaves P = 1 (both results 0).
  float P = 1.f, dP = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float t = r + (float)i;
    const bool on = (float)i < x;
    dP = on ? fmaf(dP, t, P) : dP;
    P = on ? P * t : P;
  }
  o.lg = flog(P);
  o.dg = dP * frcp(P);
  const bool small = (x <= 8.0f) && (x == floorf(x)) && (r < 1e4f);
  if (small) return o;
  // Rare lanes: both arguments are shifted above 4 by the same n (<= 4 predicated steps) and the Stirling
  // series is differenced analytically (no cancellation of two large lgamma values); at z >= 4 the three
  // correction terms leave < 4e-8 (lgamma) / 6e-8 (digamma).
  const float nf = fmaxf(ceilf(4.0f - r), 0.f);
  // (numerator and denominator of prod (r+i)/(x+r+i) kept apart: their ratio falls below the smallest normal float for
  // tiny r and large x -- r = 1e-30, x = 1e3 gives 6e-42 -- where v_log_f32 returns -inf; each product alone stays normal:
  // num >= 1e-30, den <= (x + 4)^4)
  float num = 1.f, den = 1.f, dg_shift = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = r + (float)i, b = x + a;
    const bool on = (float)i < nf;
    num = on ? num * a : num;
    den = on ? den * b : den;
    dg_shift = on ? dg_shift + x * frcp(b) * frcp(a) : dg_shift;
  }
  const float lg_shift = flog(num) - flog(den);
  const float rs = r + nf;
  const float zr = x + rs;
  const float l1p = log1p_small(x * frcp(rs));
  o.lg = x * flog(zr) + (rs - 0.5f) * l1p - x + (stirling_corr(zr) - stirling_corr(rs)) + lg_shift;
  o.dg = l1p + (digamma_corr(zr) - digamma_corr(rs)) + dg_shift;
  return o;
}

// ---------------------------------------------------------------------------
// f32 products from bf16 MFMAs on three-way split operands (gfx950: v_mfma_f32_32x32x16_bf16)
// ---------------------------------------------------------------------------
// x = x0 + x1 + x2, each term the bf16 rounding of what the earlier ones left (the first two remainders are exact in f32).
// Of the nine cross products of two split operands the six with index sum <= 2 are kept: x0 y0, x0 y1, x1 y0, x0 y2, x2 y0,
// x1 y1 -- what is dropped is below 2^-23 |x y|, the rounding of ONE f32 multiply, and the sum is accumulated in f32 by
// the MFMA.  Why: v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate and holds the SIMD's vector issue while it runs
// (tools/coexec.hip), so an f32-MFMA product and the vector work beside it ADD; six bf16 MFMAs of 32 cycles per 16 k take
// 0.375 of the cycles of eight f32 ones and run on the matrix pipe beside the vector work.
typedef __bf16 smx_bf16x8 __attribute__((ext_vector_type(8)));
typedef float smx_f32x16 __attribute__((ext_vector_type(16)));
struct Split8 { smx_bf16x8 t0, t1, t2; };
// x = t0 + t1 + t2 (bf16 each): t0 = bf16(x), r1 = x - t0 (exact), t1 = bf16(r1), t2 = bf16(r1 - t1).  The conversions run TWO values at a time:
// one v_cvt_pk_bf16_f32 per pair and term IS the packed operand register, and its two halves go back to f32 with a shift and a mask -- 5.5 vector
// instructions per value where the element-by-element spelling (a cast, a widening, a subtraction per term and value, then the packing) compiled
// to 8.5.  Round 6: the head's backward launch at C2 turned out to be bound by exactly this arithmetic (its two roles, 4.0 and 4.9 us alone, ADD
// to 8.6 although their workgroups are resident together: ~6.6 k of ~8.5 k cycles per SIMD are operand splits); the fused head had the
// pair-wise form since round 5 (hf_split_pair).  The same roundings in the same order: the same bits.
typedef float smx_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 smx_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int smx_u32x4 __attribute__((ext_vector_type(4)));
__device__ inline void split3_pair(float x0, float x1, unsigned& t0, unsigned& t1, unsigned& t2) {
  const unsigned a = __builtin_bit_cast(unsigned, __builtin_convertvector(smx_f32x2{x0, x1}, smx_bf16x2));
  const float r0 = x0 - __uint_as_float(a << 16), r1 = x1 - __uint_as_float(a & 0xFFFF0000u);      // exact
  const unsigned b = __builtin_bit_cast(unsigned, __builtin_convertvector(smx_f32x2{r0, r1}, smx_bf16x2));
  const float s0 = r0 - __uint_as_float(b << 16), s1 = r1 - __uint_as_float(b & 0xFFFF0000u);
  t0 = a; t1 = b; t2 = __builtin_bit_cast(unsigned, __builtin_convertvector(smx_f32x2{s0, s1}, smx_bf16x2));
}
__device__ inline Split8 split3x8(const float (&x)[8]) {
  unsigned a[4], b[4], c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) split3_pair(x[2 * k], x[2 * k + 1], a[k], b[k], c[k]);
  Split8 o;
  o.t0 = __builtin_bit_cast(smx_bf16x8, smx_u32x4{a[0], a[1], a[2], a[3]});
  o.t1 = __builtin_bit_cast(smx_bf16x8, smx_u32x4{b[0], b[1], b[2], b[3]});
  o.t2 = __builtin_bit_cast(smx_bf16x8, smx_u32x4{c[0], c[1], c[2], c[3]});
  return o;
}
// acc += A B over 16 k: lane (i, h) gives rows / columns i, k = 8 h .. 8 h + 7 of the step (smallest terms first)
__device__ inline smx_f32x16 mfma_bf16x3(const Split8& a, const Split8& b, smx_f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t2, b.t0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t0, b.t2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t1, b.t1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t1, b.t0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t0, b.t1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t0, b.t0, acc, 0, 0, 0);
  return acc;
}

// ---------------------------------------------------------------------------
// LDS-DMA (global -> LDS, 16 bytes per lane, no VGPR destination) as inline assembly
// ---------------------------------------------------------------------------
// One wave-instruction lands 64 x 16 B = 1 KiB linearly at the wave-uniform LDS byte address `lds_dst`; each lane names its own
// source.  Written as inline asm, not __builtin_amdgcn_global_load_lds, for pipelines that keep several stages in flight across a
// barrier: the compiler tracks the builtin as a pending LDS write and puts `s_waitcnt vmcnt(0)` before every ds_read it cannot
// prove disjoint (a ring of stage buffers indexed by st % N is such a case -- the pipeline then runs one stage deep); an asm load
// is invisible to that logic, so the kernel's own counted `s_waitcnt vmcnt(N)` + `s_barrier` are what orders the reads
// (cdna_hip_programming.md: "Pipelining across barriers").  M0 carries the LDS base and is compiler-reserved: saved and restored.
__device__ inline void glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ inline uint32_t lds_addr(const void* p) {   // byte address within the workgroup's LDS of a pointer into a __shared__ array
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// ---------------------------------------------------------------------------
// Kernel arguments: the argument structs of the step's launches are hundreds of bytes (4 - 20 cache lines of the kernarg segment), and the
// compiler loads their fields where it needs them.
// (A kernarg_warm<BYTES>() that requested every 64-byte line of the segment in one batch of scalar loads at kernel entry was measured -- no
// gain: the WAITS between the lazily placed field loads are the cost, not their misses -- and removed: rounded up to whole groups of lines it read
// past the end of a 600-byte segment, which is past the end of the runtime's kernarg pool when that launch's arguments are the pool's last:
// a memory access fault once per few thousand launches, in whichever workload's launch sequence happened to land there.)
// ---------------------------------------------------------------------------
// preload(field, ...): have these kernel-argument fields in scalar registers HERE.  The compiler loads a field of a by-value argument
// struct where it is first needed, which inside predicated code means s_load -> s_waitcnt -> use once per field and branch: a chain of
// DEPENDENT scalar-cache round trips (~200 cycles each even when they hit; 17 of them ahead of the first global load of the decoder's
// BatchNorm-forward launch).  An empty asm that takes the values as scalar inputs makes the loads one batch with one wait.
template <class T>
__device__ __forceinline__ auto sbits(const T& v) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "preload: 4- or 8-byte fields");
  if constexpr (sizeof(T) == 4) return __builtin_bit_cast(uint32_t, v);
  else return __builtin_bit_cast(uint64_t, v);
}
// (one asm statement per call -- all of a call's fields are one batch of loads; 8, 16 or 24 fields: an asm statement takes at most 30 operands)
template <class T0, class T1, class T2, class T3, class T4, class T5, class T6, class T7>
__device__ __forceinline__ void preload(const T0& v0, const T1& v1, const T2& v2, const T3& v3, const T4& v4, const T5& v5, const T6& v6, const T7& v7) { asm volatile("" ::"s"(sbits(v0)), "s"(sbits(v1)), "s"(sbits(v2)), "s"(sbits(v3)), "s"(sbits(v4)), "s"(sbits(v5)), "s"(sbits(v6)), "s"(sbits(v7))); }
template <class T0, class T1, class T2, class T3, class T4, class T5, class T6, class T7, class T8, class T9, class T10, class T11, class T12, class T13, class T14, class T15>
__device__ __forceinline__ void preload(const T0& v0, const T1& v1, const T2& v2, const T3& v3, const T4& v4, const T5& v5, const T6& v6, const T7& v7, const T8& v8, const T9& v9, const T10& v10, const T11& v11, const T12& v12, const T13& v13, const T14& v14, const T15& v15) { asm volatile("" ::"s"(sbits(v0)), "s"(sbits(v1)), "s"(sbits(v2)), "s"(sbits(v3)), "s"(sbits(v4)), "s"(sbits(v5)), "s"(sbits(v6)), "s"(sbits(v7)), "s"(sbits(v8)), "s"(sbits(v9)), "s"(sbits(v10)), "s"(sbits(v11)), "s"(sbits(v12)), "s"(sbits(v13)), "s"(sbits(v14)), "s"(sbits(v15))); }
template <class T0, class T1, class T2, class T3, class T4, class T5, class T6, class T7, class T8, class T9, class T10, class T11, class T12, class T13, class T14, class T15, class T16, class T17, class T18, class T19, class T20, class T21, class T22, class T23>
__device__ __forceinline__ void preload(const T0& v0, const T1& v1, const T2& v2, const T3& v3, const T4& v4, const T5& v5, const T6& v6, const T7& v7, const T8& v8, const T9& v9, const T10& v10, const T11& v11, const T12& v12, const T13& v13, const T14& v14, const T15& v15, const T16& v16, const T17& v17, const T18& v18, const T19& v19, const T20& v20, const T21& v21, const T22& v22, const T23& v23) { asm volatile("" ::"s"(sbits(v0)), "s"(sbits(v1)), "s"(sbits(v2)), "s"(sbits(v3)), "s"(sbits(v4)), "s"(sbits(v5)), "s"(sbits(v6)), "s"(sbits(v7)), "s"(sbits(v8)), "s"(sbits(v9)), "s"(sbits(v10)), "s"(sbits(v11)), "s"(sbits(v12)), "s"(sbits(v13)), "s"(sbits(v14)), "s"(sbits(v15)), "s"(sbits(v16)), "s"(sbits(v17)), "s"(sbits(v18)), "s"(sbits(v19)), "s"(sbits(v20)), "s"(sbits(v21)), "s"(sbits(v22)), "s"(sbits(v23))); }

// ---------------------------------------------------------------------------
// development: cycle stamps of a kernel's phases (builds with -DSMX_STAMPS only: tools/c2_stamps.sh).  Thread 0 of workgroup
// SMX_STAMP_WG writes clock64() into the translation unit's own table [slot][16]; smx_dbg_stamps_<unit>() copies it out.
// ---------------------------------------------------------------------------
#ifdef SMX_STAMPS
#ifndef SMX_STAMP_WG
#define SMX_STAMP_WG 0
#endif
#define SMX_STAMP_TABLE static __device__ long long smx_tu_stamps[16 * 16];
#define SMX_STAMP(slot, id) do { if ((int)blockIdx.x == SMX_STAMP_WG && threadIdx.x == 0) smx_tu_stamps[(slot) * 16 + (id)] = clock64(); } while (0)
#else
#define SMX_STAMP_TABLE
#define SMX_STAMP(slot, id) do {} while (0)
#endif

// ---------------------------------------------------------------------------
// wave / block reductions (wave = 64)
// ---------------------------------------------------------------------------
// (four DPP steps inside the rows of 16, then two ds_bpermute across the four rows -- instead of six ds_bpermute)
// a lane's value to every lane (v_readlane: the lane index is uniform); __shfl would go through the LDS crossbar (ds_bpermute)
__device__ inline float lane_bcast(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
// v + (v of lane ^ 16) and v + (v of lane ^ 32) without the LDS crossbar: gfx950's v_permlane16_swap / v_permlane32_swap exchange rows /
// halves between two registers holding the same value, after which their sum is what `v + __shfl_xor(v, 16 | 32)` gives (the same two
// addends in every lane: the same bits; tools/dev/permlane_sum.hip).  Inline asm with the hazard's wait states spelled out: given the SAME
// value for both operands the builtins (ROCm 7.2) return the first register twice.
// Callers: the rows that trade places must be active together.  Every early `return` ahead of a wave_sum / half_wave_sum in csrc/ is
// wave-uniform (one wave per cell or row, or a block-uniform test), except the factor kernels' `r >= 2 * B`, where whole half-waves
// leave and half_wave_sum's row swap stays inside the half that remains.
__device__ inline float xor16_add(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ inline float xor32_add(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ inline float wave_sum(float v) {
#define SMX_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, false))
  SMX_DPP_ADD(0xB1);    // quad_perm [1, 0, 3, 2]
  SMX_DPP_ADD(0x4E);    // quad_perm [2, 3, 0, 1]
  SMX_DPP_ADD(0x141);   // row_half_mirror
  SMX_DPP_ADD(0x140);   // row_mirror
#undef SMX_DPP_ADD
  return xor32_add(xor16_add(v));   // (were two ds_bpermute round trips)
}
// Sum over the 32 lanes of each half of a wave (lanes 0..31 and 32..63 separately), every lane gets its half's sum.
// Four DPP steps inside the rows of 16 (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: ~8 cycles each) and one row exchange.
__device__ inline float half_wave_sum(float v) {
#define SMX_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, false))
  SMX_DPP_ADD(0xB1);    // quad_perm [1, 0, 3, 2]
  SMX_DPP_ADD(0x4E);    // quad_perm [2, 3, 0, 1]
  SMX_DPP_ADD(0x141);   // row_half_mirror
  SMX_DPP_ADD(0x140);   // row_mirror
#undef SMX_DPP_ADD
  return xor16_add(v);
}
__device__ inline float wave_max(float v) {
#define SMX_DPP_MAX(ctrl) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, 0xF, 0xF, false)))
  SMX_DPP_MAX(0xB1);
  SMX_DPP_MAX(0x4E);
  SMX_DPP_MAX(0x141);
  SMX_DPP_MAX(0x140);
#undef SMX_DPP_MAX
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}

}  // namespace smx
