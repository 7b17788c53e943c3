// smx_headfused.hip -- the WHOLE output head of a training step at a wide gene panel in ONE launch (SURVEY.md 8 rows a-9,
// a-10 / a-11 and the head's part of a-16; BASELINE.json configs[4]: 20 000 genes, 128 cells per GPU and step):
//
//   P = d W_out + b  ->  NB / ZINB / NBD / ZINBD log-likelihood of x, d llk / d P (scaled) = dP
//   dW_out = d^T dP,  db_out = colsum(dP),  sum of squares of dW_out (clipnorm),  d d = dP W_out^T (per-workgroup slabs)
//
// Neither P nor dP exists in memory.  Before (round 3): out_head_loss_kernel (P in registers, dP stored: 31 MB) -> bigk_kernel +
// bigk_reduce_kernel (d d: dP and W_out read again) -> panel_dw_kernel (dW: dP read a third time): 35 + 20 + 5 + 18 us at
// 128 x 20 000 with three launch boundaries between them.  Here a workgroup OWNS gene tiles (all k planes, all <= 128 cells) from
// the raw weights to their gradients and walks them; W_out is read once and dW_out written once.
//
// Shape of the work (the flash-attention-backward decomposition, cdna_hip_programming.md "Attention backward": the gene plays
// the key, the cell the query):
//  * 8 waves (2 per SIMD, 256 registers each); wave w owns the cells 16 w .. 16 w + 15 in the forward product and in d d, and
//    the rows 16 w .. 16 w + 15 of H in dW.  All products are v_mfma_f32_16x16x32_bf16 on three-way split operands (six of
//    the nine cross products: f32 accuracy, smx_device.h).
//  * forward with the CELL ON THE LANE: P^T[rho][cell] = sum_h W[h][rho] d[cell][h] (rho = plane, gene of the unit).  A lane
//    then holds, for ONE cell, 4 genes x k planes: the likelihood runs on the accumulators where they are, and its result dP^T
//    is (after one v_permlane32_swap: below) the B operand of d d^T[h][cell] += sum_rho W[h][rho] dP^T[rho][cell] (a product that
//    sums over the accumulator's ROW index), accumulated in 32 registers for the whole launch.
//  * dW = d^T dP sums over the cell = the LANE index: dP crosses LDS once as a [cell][gene] image of three bf16 terms per plane and is
//    read back column-wise by ds_read_b64_tr_b16 (the hardware transpose read).
//  * W_out's tile is split ONCE per workgroup into a [h][gene] bf16 x 3 image per plane: the forward product reads it transposed
//    (ds_read_b64_tr_b16), d d reads its rows.
//  * d's operands never change: a wave's two views of d (its cells' rows for the forward product, its H rows' columns for dW) are
//    split once and stay in 48 + 48 registers (zinbd: the second is parked in an L2-resident table and fetched per unit); no LDS
//    image of d (144 KB of LDS are the images above, two slots of each).
//  * the k index of every product is permuted consistently in both operands (lane group g, element e <-> k = 4 g + e for
//    e < 4, 16 + 4 g + e - 4 otherwise) so that the transposed reads of a 32-lane half touch 8 consecutive rows.
//  * the per-wave queue of the non-zero counts (smx_loss.h) lives in the wave's own rows of the dP image.
// Deterministic: no atomics; per-workgroup d d slabs are summed by bigk_reduce_kernel in workgroup order.
// Round 4's kernel walked whole tiles of 32 genes with two barriers per tile (one phase kind between them); round 5's (below) walks
// units of 16 genes through two slots of each image with one barrier per unit -- profiles/r05_head_fused_experiments.txt has the A/B.
#include <stdlib.h>

#include <algorithm>

#include <hip/hip_ext.h>
#include "smx_internal.h"
#include "smx_loss.h"
#include "../../include/sisua_hip.h"

namespace smx {

typedef float hf_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 hf_bf16x4 __attribute__((ext_vector_type(4)));
typedef short hf_s16x4 __attribute__((ext_vector_type(4)));
#define HF_LDS3(p) ((__attribute__((address_space(3))) hf_s16x4*)(p))

struct Split4 { hf_bf16x4 t0, t1, t2; };
__device__ inline Split4 split3x4(const float (&x)[4]) {
  Split4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const __bf16 a = (__bf16)x[k];
    const float r1 = x[k] - (float)a;   // exact
    const __bf16 b = (__bf16)r1;
    const __bf16 c = (__bf16)(r1 - (float)b);
    o.t0[k] = a; o.t1[k] = b; o.t2[k] = c;
  }
  return o;
}
__device__ inline smx_bf16x8 cat8(hf_bf16x4 a, hf_bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ inline hf_bf16x4 lo4(smx_bf16x8 v) { return __builtin_shufflevector(v, v, 0, 1, 2, 3); }
__device__ inline hf_bf16x4 hi4(smx_bf16x8 v) { return __builtin_shufflevector(v, v, 4, 5, 6, 7); }
__device__ inline hf_bf16x4 as_bf(hf_s16x4 v) { return __builtin_bit_cast(hf_bf16x4, v); }

// acc += A B over 32 k on 16 x 16 tiles, operands split three ways (smallest terms first, as mfma_bf16x3)
__device__ inline hf_f32x4 mfma16_bf16x3(const Split8& a, const Split8& b, hf_f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t2, b.t0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t0, b.t2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t1, b.t1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t1, b.t0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t0, b.t1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t0, b.t0, acc, 0, 0, 0);
  return acc;
}

// Raw buffer accesses: one 32-bit lane offset + a scalar offset per instruction against a descriptor in scalar registers.  With plain
// pointers the compiler forms every loop-invariant 64-bit lane address ONCE, ahead of the tile loop -- ~60 register pairs for the table
// entries, the dW rows and the W segments -- and spills them around it (cdna_hip_programming.md: "a lane-constant address hoisted to
// kernel entry is spilled around the tile loop").  Loads beyond num_records return 0, stores there are dropped.
typedef unsigned int hf_u32x4 __attribute__((ext_vector_type(4)));
__device__ inline int hf_hide(int x) { asm volatile("" : "+v"(x)); return x; }
// LDS byte address = opaque lane base + constant: sub-image si (term x planes + plane) of an image, `more` further bytes
#define HF_AT(bases, hf, si, more) (lds + (bases)[(si) >= 4][hf] + ((si) - 4 * ((si) >= 4)) * 8192 + (more))
__device__ inline __amdgpu_buffer_rsrc_t hf_rsrc(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0xFFFFFFFFL ? 0xFFFFFFFFL : bytes), 0x00020000);
}
__device__ inline float4 hf_load4(__amdgpu_buffer_rsrc_t r, int vo, int so) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0));
}
__device__ inline smx_bf16x8 hf_load8h(__amdgpu_buffer_rsrc_t r, int vo, int so) {
  return __builtin_bit_cast(smx_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0));
}
__device__ inline void hf_store8h(smx_bf16x8 v, __amdgpu_buffer_rsrc_t r, int vo, int so) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(hf_u32x4, v), r, vo, so, 0);
}
__device__ inline void hf_store1(float v, __amdgpu_buffer_rsrc_t r, int vo, int so) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, vo, so, 0);
}

#ifndef SMX_HF_STAMP_WAVE
#define SMX_HF_STAMP_WAVE 0
#endif
// x = t0 + t1 + t2 (bf16 each; split3x8's arithmetic, bit for bit) with the conversions two values at a time: one v_cvt_pk_bf16_f32 per
// pair and term IS the packed operand register; its two halves go back to f32 with a shift and a mask (5.5 vector instructions per value
// where the element-wise spelling above compiles to 8.5).  Since round 6 smx_device.h's split3x8 is this form too (the whole library).
typedef float hf_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 hf_bf16x2 __attribute__((ext_vector_type(2)));
__device__ inline void hf_split_pair(float x0, float x1, unsigned& t0, unsigned& t1, unsigned& t2) {
  const unsigned a = __builtin_bit_cast(unsigned, __builtin_convertvector(hf_f32x2{x0, x1}, hf_bf16x2));
  const float r0 = x0 - __uint_as_float(a << 16), r1 = x1 - __uint_as_float(a & 0xFFFF0000u);      // exact
  const unsigned b = __builtin_bit_cast(unsigned, __builtin_convertvector(hf_f32x2{r0, r1}, hf_bf16x2));
  const float s0 = r0 - __uint_as_float(b << 16), s1 = r1 - __uint_as_float(b & 0xFFFF0000u);
  t0 = a; t1 = b; t2 = __builtin_bit_cast(unsigned, __builtin_convertvector(hf_f32x2{s0, s1}, hf_bf16x2));
}
__device__ inline Split4 hf_split4(const float (&x)[4]) {
  typedef unsigned int hf_u32x2 __attribute__((ext_vector_type(2)));
  unsigned a[2], b[2], c[2];
  hf_split_pair(x[0], x[1], a[0], b[0], c[0]);
  hf_split_pair(x[2], x[3], a[1], b[1], c[1]);
  Split4 o;
  o.t0 = __builtin_bit_cast(hf_bf16x4, hf_u32x2{a[0], a[1]}); o.t1 = __builtin_bit_cast(hf_bf16x4, hf_u32x2{b[0], b[1]}); o.t2 = __builtin_bit_cast(hf_bf16x4, hf_u32x2{c[0], c[1]});
  return o;
}
__device__ inline Split8 hf_split8(const float (&x)[8]) {
  unsigned a[4], b[4], c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) hf_split_pair(x[2 * k], x[2 * k + 1], a[k], b[k], c[k]);
  Split8 o;
  o.t0 = __builtin_bit_cast(smx_bf16x8, hf_u32x4{a[0], a[1], a[2], a[3]}); o.t1 = __builtin_bit_cast(smx_bf16x8, hf_u32x4{b[0], b[1], b[2], b[3]});
  o.t2 = __builtin_bit_cast(smx_bf16x8, hf_u32x4{c[0], c[1], c[2], c[3]});
  return o;
}

// v_permlane32_swap on both registers of two 4 x bf16 values: lanes 0-31 end with (a of their own, a of lane + 32), lanes 32-63 with (b of
// lane - 32, b of their own) -- lane groups g and g + 2 of a wave exchange halves
__device__ inline smx_bf16x8 hf_swap32(hf_bf16x4 a, hf_bf16x4 b) {
  typedef unsigned int hf_u32x2 __attribute__((ext_vector_type(2)));
  const hf_u32x2 ua = __builtin_bit_cast(hf_u32x2, a), ub = __builtin_bit_cast(hf_u32x2, b);
  // (vdst = a, src0 = b: lanes 32-63 of vdst <-> lanes 0-31 of src0)
  const auto r0 = __builtin_amdgcn_permlane32_swap(ua[0], ub[0], false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(ua[1], ub[1], false, false);
  const hf_u32x4 o = hf_u32x4{r0[0], r1[0], r0[1], r1[1]};
  return __builtin_bit_cast(smx_bf16x8, o);
}

// the wave's queue of non-zero counts inside rows 16 w .. 16 w + 15 of the first four sub-images of a dP slot (32-byte rows: 512 bytes each)
struct Hf2Queue {
  unsigned char* base;
  __device__ float2& operator[](int k) const { return *reinterpret_cast<float2*>(base + (k >> 6) * 4096 + (k & 63) * 8); }
  __device__ explicit operator bool() const { return true; }
};

// =====================================================================================================================================
// Round 5: the same head at HALF-tile granularity (16 genes), software-pipelined so that the matrix and the vector phases overlap.
//
// Round 4's kernel kept its two waves per SIMD in the same phase: two workgroup barriers per tile, ONE phase kind between them (matrix:
// forward product, d d, dW; vector: W split, likelihood, dP split) -- the pipes followed each other (profiles/r04_pipe_occupancy_head_fused.txt:
// any-instruction 47-52 %).  Here a workgroup walks units of 16 genes through TWO W-image slots and TWO dP-image slots (the same 144 KB):
//   interval i:   F / L / D of unit i   (reads W slot i & 1, writes dP slot i & 1 -- the wave's own rows)
//                 dW of unit i - 1      (reads dP slot (i - 1) & 1 -- complete since the barrier)
//                 S of unit i + 1       (splits the next W tile into W slot (i + 1) & 1 -- free since the barrier)
//   ONE barrier per interval; the three parts are independent of each other, so the two waves of a SIMD drift apart inside an interval
//   instead of being put back in step by a barrier after every phase.  (Measured, profiles/r05_head_fused_experiments.txt: giving the
//   halves of the workgroup different orders -- F L D dW S against S dW F L D --, s_setprio around the matrix phases and operands
//   requested two steps ahead all leave the launch where it is, 51.6-52.1 us: the kernel is ISSUE-bound -- ~1600 instructions per wave
//   and unit, any-instruction busy 74 % -- and what one wave gains beside its partner the partner loses.)
// Images: [term 3][plane NP][row 128][16 genes] bf16 -- 32-byte rows of four 8-byte slots (slot s = genes 4 s .. 4 s + 3) in the order
// 0, 2, 1, 3: the transposed reads (a half-wave: 8 consecutive rows x 32 bytes) and the 16-byte row accesses (slots s, s + 2 of a row; a
// ds_read_b128 group: rows 0-3 / 12-15 first half, rows 4-11 second half) are both conflict-free without a swizzle.
// d d sums over rho = (plane, gene): a lane holds dP of 4 genes x NP planes of its cell; v_permlane32_swap between the lane groups g and
// g + 2 turns that into 8 genes of ONE plane (groups 0, 1: plane 0; groups 2, 3: plane 1) -- the 16-byte chunk that is written to the dP
// image, and the B operand of a k step of 32 against ONE 16-byte row read of W.  The third plane is a k step of its own (groups 2, 3: zeros).
// The units of a workgroup are CONSECUTIVE (its 64-byte row segments of W / dW and its 32-byte pieces of the counts pair up in L2), and
// 1250 units over 250 workgroups x 5 quantise better than 625 tiles over 209 x 3.  The likelihood partial is one per cell and WORKGROUP.
// =====================================================================================================================================
// ACC: the launch ADDS its dW / db to what is there (the second 128 cells of a minibatch of up to 256: launch_head_fused) -- the sum of
// squares it leaves is then the one of the finished gradient.
template <int LK, int U16, int ACC>
__global__ __launch_bounds__(512, 2) void head_fused_kernel(HeadFusedArgs a) {
  // both views of d resident (48 + 48 registers), except zinbd (104 bytes of scratch that way): its view for dW is parked in the L2-resident
  // table (a.dtab, 96 KB, the same bytes from every workgroup) and fetched per unit
  constexpr int RES = LK != SMX_LLK_ZINBD;
  constexpr int NP = (LK == SMX_LLK_ZINB || LK == SMX_LLK_ZINBD) ? 3 : 2;
  constexpr int SUBI = 4096;               // bytes of one (term, plane) sub-image: [128 rows][16 genes] bf16
  constexpr int IMG = 3 * NP * SUBI;       // one slot
  constexpr int DBX = 8 * 16 * NP;         // floats of one slot of bias-gradient partials [share 8][plane][16 genes]
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // lds: W slots 0, 1 | dP slots 0, 1 | bias partials, slots 0, 1
  float* const dbx = reinterpret_cast<float*>(lds + 4 * IMG);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4, q4 = j >> 2, pp = j & 3;
  // lane parts of the LDS addresses (opaque: see hf_hide); every other part is a slot base (one add per phase) or an immediate
#define HF2_SLOT(s) (16 * ((s) & 1) + 8 * ((s) >> 1))                             /* byte offset of slot s within a 32-byte row */
  const int tbl = hf_hide(32 * (4 * g + q4) + HF2_SLOT(pp));                     // transposed read: row 4 g + q4 (+ 32 ks + 16 rd), slot pp
  const int rbl = hf_hide(32 * j + 16 * (g & 1));                                // 16-byte row access: row j (+ 16 hs), slots g & 1 and (g & 1) + 2 ...
  const int rbp = hf_hide(32 * j + 16 * (g & 1) + (g >> 1) * SUBI);              // ... of plane g >> 1 (the k step over planes 0 and 1)
  const int wt = tid >> 2, wc = tid & 3;                                         // W split: row wt, genes 4 wc .. 4 wc + 3
  const int wwl = hf_hide(32 * wt + HF2_SLOT(wc));
  const int wgo = (int)((wt * a.ldw + 4 * wc) * 4);
  const long wbytes = 128L * a.ldw * 4;
  const __amdgpu_buffer_rsrc_t rW = hf_rsrc(a.W, wbytes), rdW = hf_rsrc(a.dW, wbytes);
  const __amdgpu_buffer_rsrc_t rbias = hf_rsrc(a.bias, a.ldw * 4), rdb = hf_rsrc(a.db, a.ldw * 4);
  const __amdgpu_buffer_rsrc_t rtab = hf_rsrc(a.dtab, SMX_HEAD_FUSED_TAB_BYTES);
  const int u0 = blockIdx.x * a.per_wg, nu = min(a.per_wg, a.n_gt - u0);   // this workgroup's units: u0 .. u0 + nu - 1 (nu >= 1)

  float4 wreg[NP];
  auto load_w = [&](int u) {
#pragma unroll
    for (int p = 0; p < NP; ++p) wreg[p] = hf_load4(rW, wgo, (int)((u * 16 + (long)p * a.Gp) * 4));
  };
  load_w(u0);
  float4 wnext[NP];   // (unit 1's tile, requested now: its HBM latency runs under the split of d instead of in front of the first W split of waves 4-7)
#pragma unroll
  for (int p = 0; p < NP; ++p) wnext[p] = nu > 1 ? hf_load4(rW, wgo, (int)(((u0 + 1) * 16 + (long)p * a.Gp) * 4)) : zero4();
#ifdef SMX_HF_STAMPS
  if (a.dbg && tid == 64 * SMX_HF_STAMP_WAVE && (blockIdx.x == 0 || blockIdx.x == 100)) a.dbg[(blockIdx.x ? 64 : 0) + 63] = clock64();
#endif

  // ---- both views of d, split once ------------------------------------------------------------------------------
  Split8 dB[4], dA[4];
  hf_f32x4 accDD[8];
  const int w = wave;
  const int cell = 16 * w + j;
  const int dw_vo = (int)(((16 * w + 4 * g) * a.ldw + j) * 4);
  const int cellc = min(cell, a.B - 1);
  const bool cell_ok = cell < a.B;
  // (the cell's dataset row: requested here, used by load_xb behind the split of d.  No branch on `rows`: inside one the compiler put the first use of the
  // loaded word -- its sign extension -- together with a wait for it, one memory round trip at the head of every workgroup before any of d's 40 requests)
  const bool tab = a.rows != nullptr;
  const int32_t* const rp = tab ? a.rows : reinterpret_cast<const int32_t*>(a.D);
  const int src_w = rp[tab ? cellc : 0];
  {
    const float* dp = a.D + (long)cellc * a.ldd;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float4 lo = *reinterpret_cast<const float4*>(dp + 32 * ks + 4 * g);
      const float4 hi = *reinterpret_cast<const float4*>(dp + 32 * ks + 16 + 4 * g);
      const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      dB[ks] = hf_split8(x);
    }
    const float* da = a.D + 16 * w + j;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = 32 * ks + (e < 4 ? 4 * g + e : 16 + 4 * g + e - 4);
        x[e] = da[(long)min(c, a.B - 1) * a.ldd];
      }
      dA[ks] = hf_split8(x);
    }
#pragma unroll
    for (int hs = 0; hs < 8; ++hs) accDD[hs] = hf_f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int tab_vo = lane * 16, tabA = wave * 12 * 1024;
  if (!RES) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      hf_store8h(dA[ks].t0, rtab, tab_vo, tabA + (3 * ks + 0) * 1024); hf_store8h(dA[ks].t1, rtab, tab_vo, tabA + (3 * ks + 1) * 1024); hf_store8h(dA[ks].t2, rtab, tab_vo, tabA + (3 * ks + 2) * 1024);
    }
#define HF_V(x) "v"(x.t0), "v"(x.t1), "v"(x.t2)
    // the wave's own table entries are written before it reads them back.  The stored registers are operands of the wait, so that nothing
    // overwrites them before the stores have completed.  ISA fact (gfx950, measured: tools/dev/store_hazard.hip, profiles/r06_hazards.txt):
    // `buffer_store_dwordx4 v[a:a+3], voff, srsrc, sN offen` followed DIRECTLY by a vector instruction that writes v[a] stores the NEW value
    // in 0.15 % of the lanes; one wait state cures it.  The same store with an IMMEDIATE soffset needs two, which the compiler inserts
    // (the ">64-bit VMEM store -> VALU write of its data" hazard); with a REGISTER soffset -- these stores: soffset = the wave's table
    // base -- LLVM's GCNHazardRecognizer::createsVALUHazard assumes none are needed and emitted `buffer_store_dwordx4 v[14:17] ...;
    // v_pk_add_f32 v[14:15], ...` (the split of the next k step) back to back.  Round 4 saw the table's garbage and held the registers
    // without knowing why; tools/isa_lint.py rule R2 now rejects that pair in any kernel of the library at build time
    // (-DSMX_HF_NOHOLD rebuilds the failing form: the lint stops the build).
#ifdef SMX_HF_NOHOLD
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(0)" :: HF_V(dA[0]), HF_V(dA[1]), HF_V(dA[2]), HF_V(dA[3]) : "memory");
#endif
#undef HF_V
  }
  auto load_view = [&]() {
    const int vo = hf_hide(tab_vo);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      dA[ks].t0 = hf_load8h(rtab, vo, tabA + (3 * ks + 0) * 1024); dA[ks].t1 = hf_load8h(rtab, vo, tabA + (3 * ks + 1) * 1024); dA[ks].t2 = hf_load8h(rtab, vo, tabA + (3 * ks + 2) * 1024);
    }
  };
  float ssq = 0.f, lsum = 0.f;
  const bool gate = __builtin_amdgcn_readfirstlane(hf_hide(1)) != 0;
#ifdef SMX_HF_STAMPS
  int dbg_n = 0;
#define HF2_STAMP() do { if (a.dbg && tid == 64 * SMX_HF_STAMP_WAVE && (blockIdx.x == 0 || blockIdx.x == 100) && dbg_n < 62) { a.dbg[(blockIdx.x ? 64 : 0) + dbg_n] = clock64(); ++dbg_n; } } while (0)
#else
#define HF2_STAMP() do { if (!gate) asm volatile("s_nop 0"); } while (0)
#endif
  HF2_STAMP();

  // ---- prefetched operands of a unit: counts of the lane's cell (4 genes), biases of the lane's 4 genes ---------------------
  uint2 xraw16; float4 xraw32 = zero4(); float4 bq[NP];
  xraw16.x = 0; xraw16.y = 0;
  auto load_xb = [&](int u) {
    const long o = (long)(tab ? src_w : cellc) * a.ldx + (long)u * 16 + 4 * g;
    if (U16) xraw16 = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(a.X) + o);
    else xraw32 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.X) + o);
#pragma unroll
    for (int p = 0; p < NP; ++p) bq[p] = hf_load4(rbias, 16 * g, (p * a.Gp + u * 16) * 4);
  };

  // transposed operand of k step ks over the 16 genes of plane p: rows 32 ks + 16 rd + 4 g + q4 of the image at `base` (slot + tbl)
  auto tr_read = [&](int base, int p, int ks) {
    Split8 r;
    r.t0 = cat8(as_bf(__builtin_amdgcn_ds_read_tr16_b64_v4i16(HF_LDS3(lds + base + (0 * NP + p) * SUBI + 1024 * ks))), as_bf(__builtin_amdgcn_ds_read_tr16_b64_v4i16(HF_LDS3(lds + base + (0 * NP + p) * SUBI + 1024 * ks + 512))));
    r.t1 = cat8(as_bf(__builtin_amdgcn_ds_read_tr16_b64_v4i16(HF_LDS3(lds + base + (1 * NP + p) * SUBI + 1024 * ks))), as_bf(__builtin_amdgcn_ds_read_tr16_b64_v4i16(HF_LDS3(lds + base + (1 * NP + p) * SUBI + 1024 * ks + 512))));
    r.t2 = cat8(as_bf(__builtin_amdgcn_ds_read_tr16_b64_v4i16(HF_LDS3(lds + base + (2 * NP + p) * SUBI + 1024 * ks))), as_bf(__builtin_amdgcn_ds_read_tr16_b64_v4i16(HF_LDS3(lds + base + (2 * NP + p) * SUBI + 1024 * ks + 512))));
    return r;
  };
  // d d's A operand for rows 16 hs + j of the W image: k step 0 = the lane group's plane (g >> 1), genes of slots g & 1 and (g & 1) + 2;
  // k step 1 (three planes) = the same chunk of plane 2 (lane groups 2, 3 meet zeros in the B operand)
  auto row_read = [&](int wbase, int kk, int hs) {
    const int base = wbase + (kk == 0 ? rbp : rbl);
    const int pl = kk == 0 ? 0 : 2;
    Split8 r;
    r.t0 = *reinterpret_cast<const smx_bf16x8*>(lds + base + (0 * NP + pl) * SUBI + 512 * hs);
    r.t1 = *reinterpret_cast<const smx_bf16x8*>(lds + base + (1 * NP + pl) * SUBI + 512 * hs);
    r.t2 = *reinterpret_cast<const smx_bf16x8*>(lds + base + (2 * NP + pl) * SUBI + 512 * hs);
    return r;
  };

  // ---- S: the W tile of unit u (in wreg) -> bf16 x 3 image in W slot u & 1; then the loads of the tile after it ------------
  auto phase_S = [&](int i, bool more) {
    const int base = ((i & 1) ? IMG : 0) + wwl;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const float x[4] = {wreg[p].x, wreg[p].y, wreg[p].z, wreg[p].w};
      const Split4 s = hf_split4(x);
      *reinterpret_cast<hf_bf16x4*>(lds + base + (0 * NP + p) * SUBI) = s.t0;
      *reinterpret_cast<hf_bf16x4*>(lds + base + (1 * NP + p) * SUBI) = s.t1;
      *reinterpret_cast<hf_bf16x4*>(lds + base + (2 * NP + p) * SUBI) = s.t2;
    }
    if (more) load_w(u0 + i + 1);
  };

  // ---- dW of unit i: A = the share's columns of d, B = dP read transposed; stores; the bias gradient ----------------------
  auto phase_dW = [&](int i) {
    const int n0 = (u0 + i) * 16;
    const int base = 2 * IMG + ((i & 1) ? IMG : 0) + tbl;
    if (!RES) load_view();
    hf_f32x4 accW[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int r = 0; r < 4; ++r)   // (ACC: the accumulators start from the gradient the launch before left)
        accW[p][r] = ACC ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdW, dw_vo, (int)((r * a.ldw + (long)p * a.Gp + n0) * 4), 0)) : 0.f;
    {
      // (operands requested TWO steps ahead: a wave runs this phase alone on its SIMD -- its partner is in a vector phase -- and one step of
      // six MFMAs, 96 cycles, does not cover an LDS round trip under load)
      Split8 cur = tr_read(base, 0, 0), nx1 = tr_read(base, 1 % NP, 1 / NP), nx2 = nx1;
#pragma unroll
      for (int n = 0; n < 4 * NP; ++n) {
        if (n + 2 < 4 * NP) nx2 = tr_read(base, (n + 2) % NP, (n + 2) / NP);
        accW[n % NP] = mfma16_bf16x3(dA[n / NP], cur, accW[n % NP]);
        cur = nx1; nx1 = nx2;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x = accW[p][r];
        hf_store1(x, rdW, dw_vo, (int)((r * a.ldw + (long)p * a.Gp + n0) * 4));
        ssq += x * x;
      }
    if (tid < 16 * NP) {   // bias gradient: the eight shares' partials in order
      const float* bx = dbx + (i & 1) * DBX;
      float t = ACC ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdb, (tid & 15) * 4, ((tid >> 4) * a.Gp + n0) * 4, 0)) : 0.f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) t += bx[ww * 16 * NP + tid];
      hf_store1(t, rdb, (tid & 15) * 4, ((tid >> 4) * a.Gp + n0) * 4);
    }
  };

  // ---- F, L, D of unit i ------------------------------------------------------------------------------------------------
  auto phase_FLD = [&](int i, bool more) {
    const int n0 = (u0 + i) * 16;
    const int wbase = ((i & 1) ? IMG : 0);
    const int pbase = 2 * IMG + ((i & 1) ? IMG : 0);
    hf_f32x4 accP[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) accP[p] = hf_f32x4{bq[p].x, bq[p].y, bq[p].z, bq[p].w};
    float xs[4];
    if (U16) {
      xs[0] = (float)(xraw16.x & 0xFFFFu); xs[1] = (float)(xraw16.x >> 16); xs[2] = (float)(xraw16.y & 0xFFFFu); xs[3] = (float)(xraw16.y >> 16);
    } else {
      xs[0] = xraw32.x; xs[1] = xraw32.y; xs[2] = xraw32.z; xs[3] = xraw32.w;
    }
    // (Round 5 pinned this unpacking ahead of the forward product with an empty asm because, sunk behind it, the nbd / u16 instantiation
    // lost the term x / (mu + eps) of dP for the last gene of a unit in waves 4-7, "timing-dependent, cause not established".  The cause
    // (round 6; tools/dev/hf_hazard.hip ran hand-edited ISA variants of the kernel, tools/dev/pk_opsel_hazard.hip / pk_opsel_forms.hip
    // reproduce it in 60 lines; numbers in profiles/r06_hazards.txt): with the unpacking sunk, the SLP vectoriser paired
    // rcp(mu + eps) * x and inv * x of element 3 into `v_pk_mul_f32 v[154:155], v[206:207], v[166:167] op_sel:[0,1]` -- and on gfx950 a
    // packed-f32 instruction whose op_sel takes src1's HIGH dword for the LOW result (src0's bit clear) reads that operand as 0 in lanes
    // 48-63 whenever the SIMD's OTHER wave issues a bf16 MFMA in the same cycles (v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16; not the
    // f32 MFMA, not op_sel_hi, not the high result).  Lanes 48-63 of element 3 are gene 15 of the unit; waves 4-7 trail their SIMD partners
    // 0-3 behind each barrier and meet the partners' dW products in their likelihood.  No wait state in the issuing wave can help, and the
    // pin only moved the vectoriser's choice.  The library is built with -fno-slp-vectorize (sisua_amd/build.py; also 1-2 us per step
    // faster) and tools/isa_lint.py rule R1 rejects the instruction form in every kernel at build time: the pin is gone.)
    {
      const int base = wbase + tbl;
      Split8 cur = tr_read(base, 0, 0), nx1 = tr_read(base, 1 % NP, 1 / NP), nx2 = nx1;
#pragma unroll
      for (int n = 0; n < 4 * NP; ++n) {
        if (n + 2 < 4 * NP) nx2 = tr_read(base, (n + 2) % NP, (n + 2) / NP);
        accP[n % NP] = mfma16_bf16x3(cur, dB[n / NP], accP[n % NP]);
        cur = nx1; nx1 = nx2;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) load_xb(u0 + i + 1);   // (in flight under the likelihood and everything behind it)
    HF2_STAMP();   // forward done
    // likelihood on the accumulators: element r is gene n0 + 4 g + r of the lane's cell
    float dpv[NP][4];
    {
      float v0[4], v1[4], v2[4], llk[4], d0[4], d1[4], d2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v0[r] = accP[0][r]; v1[r] = accP[1][r]; v2[r] = NP == 3 ? accP[NP - 1][r] : 0.f; }
      if (!gate) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { llk[r] = v0[r] * xs[r]; d0[r] = v0[r]; d1[r] = v1[r]; d2[r] = v2[r]; }
      } else
      count_elem_vec<LK, 0, 4>(xs, v0, v1, v2, llk, d0, d1, d2, Hf2Queue{lds + pbase + 512 * w});
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = cell_ok && (n0 + 4 * g + r) < a.G;
        lsum += ok ? llk[r] : 0.f;
        dpv[0][r] = ok ? d0[r] * a.grad_scale : 0.f;
        dpv[1][r] = ok ? d1[r] * a.grad_scale : 0.f;
        if (NP == 3) dpv[NP - 1][r] = ok ? d2[r] * a.grad_scale : 0.f;
      }
    }
    HF2_STAMP();   // likelihood done
    // dP: bias-gradient partials over the share's 16 cells; the lane groups g and g + 2 exchange halves (permlane32_swap) so that a lane
    // holds 8 genes of one plane: the 16-byte chunk of the [cell][gene] image AND the B operand of d d^T += W dP^T
    constexpr int NK = (NP + 1) / 2;
    Split8 bop[NK];
    {
      float* bx = dbx + (i & 1) * DBX + w * 16 * NP;
      Split4 sp[NP];
      // (all row sums first, as straight-line code the scheduler can interleave -- a store per value between them made every sum a
      // basic block of its own: twelve dependent DPP chains one after the other)
      float rs[NP][4];
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = dpv[p][e];
#define HF_DPP_ADD(ctrl) t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), ctrl, 0xF, 0xF, false))
          HF_DPP_ADD(0xB1); HF_DPP_ADD(0x4E); HF_DPP_ADD(0x141); HF_DPP_ADD(0x140);
#undef HF_DPP_ADD
          rs[p][e] = t;
        }
      if (j == 0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<float4*>(bx + 16 * p + 4 * g) = make_float4(rs[p][0], rs[p][1], rs[p][2], rs[p][3]);
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) sp[p] = hf_split4(dpv[p]);
      const hf_bf16x4 z = hf_bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        const bool two = 2 * kk + 1 < NP;
        bop[kk].t0 = hf_swap32(sp[2 * kk].t0, two ? sp[two ? 2 * kk + 1 : 0].t0 : z);
        bop[kk].t1 = hf_swap32(sp[2 * kk].t1, two ? sp[two ? 2 * kk + 1 : 0].t1 : z);
        bop[kk].t2 = hf_swap32(sp[2 * kk].t2, two ? sp[two ? 2 * kk + 1 : 0].t2 : z);
      }
      // image: k step 0 -> plane g >> 1, k step 1 -> plane 2 (lane groups 0, 1 only); row 16 w + j, chunk g & 1
      {
        const int base = pbase + rbp + 512 * w;
        *reinterpret_cast<smx_bf16x8*>(lds + base + (0 * NP) * SUBI) = bop[0].t0;
        *reinterpret_cast<smx_bf16x8*>(lds + base + (1 * NP) * SUBI) = bop[0].t1;
        *reinterpret_cast<smx_bf16x8*>(lds + base + (2 * NP) * SUBI) = bop[0].t2;
      }
      if (NP == 3 && g < 2) {
        const int base = pbase + rbl + 512 * w;
        *reinterpret_cast<smx_bf16x8*>(lds + base + (0 * NP + 2) * SUBI) = bop[NK - 1].t0;
        *reinterpret_cast<smx_bf16x8*>(lds + base + (1 * NP + 2) * SUBI) = bop[NK - 1].t1;
        *reinterpret_cast<smx_bf16x8*>(lds + base + (2 * NP + 2) * SUBI) = bop[NK - 1].t2;
      }
    }
    HF2_STAMP();   // dP image written
    {
      Split8 cur = row_read(wbase, 0, 0), nx1 = row_read(wbase, 0, 1), nx2 = nx1;
#pragma unroll
      for (int n = 0; n < 8 * NK; ++n) {
        if (n + 2 < 8 * NK) nx2 = row_read(wbase, (n + 2) / 8, (n + 2) % 8);
        accDD[n % 8] = mfma16_bf16x3(cur, bop[n / 8], accDD[n % 8]);
        cur = nx1; nx1 = nx2;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- prologue: unit 0's W image, unit 1's W under way, unit 0's counts and biases ---------------------------------------
  phase_S(0, false);
#pragma unroll
  for (int p = 0; p < NP; ++p) wreg[p] = wnext[p];
  load_xb(u0);
  __syncthreads();
  HF2_STAMP();
  for (int i = 0; i <= nu; ++i) {
    if (i < nu) { phase_FLD(i, i + 1 < nu); HF2_STAMP(); }
    if (i > 0) { phase_dW(i - 1); HF2_STAMP(); }
    if (i + 1 < nu) { phase_S(i + 1, i + 2 < nu); HF2_STAMP(); }
    __syncthreads();
    HF2_STAMP();
  }

  // ---- this workgroup's slab of d d, its likelihood partial per cell, its sum-of-squares slots -----------------------------
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  if (cell_ok) {
    if (g == 0) a.llk_part[(long)cell * a.n_chunks + blockIdx.x] = lsum;
    if (a.part_colmajor) {   // [column][128 cells]: the BatchNorm-backward launch sums the slabs itself (bn_wide_bwd_kernel)
      float* op = a.part + (long)blockIdx.x * a.slab_stride + (long)(4 * g) * 128 + cell;
#pragma unroll
      for (int hs = 0; hs < 8; ++hs)
#pragma unroll
        for (int e = 0; e < 4; ++e) op[(16 * hs + e) * 128] = accDD[hs][e];
    } else {
      float* op = a.part + (long)blockIdx.x * a.slab_stride + (long)cell * 128 + 4 * g;
#pragma unroll
      for (int hs = 0; hs < 8; ++hs) *reinterpret_cast<float4*>(op + 16 * hs) = make_float4(accDD[hs][0], accDD[hs][1], accDD[hs][2], accDD[hs][3]);
    }
  }
  if (a.sq_part) {
    ssq = wave_sum(ssq);
    if (lane == 0) a.sq_part[(long)blockIdx.x * 8 + wave] = ssq;
  }
#undef HF2_STAMP
#undef HF2_SLOT
}

int head_fused_min_genes() {
  static const int v = std::max(512, (int)tuning("head_fused_min_genes", SMX_HEAD_FUSED_MIN_GENES));
  return v;
}
bool head_fused_supported(int B, int Hp, int Gp, int k) {
  return B > 0 && B <= SMX_HEAD_FUSED_MAX_CELLS && Hp == 128 && Gp % 32 == 0 && Gp >= head_fused_min_genes() && (k == 2 || k == 3) && !tuning_on("no_head_fused");
}
// workgroups: one per CU at most, every one with the same number of units (+- 1); a unit = 16 genes
static int hf_units(int Gp) { return Gp / 16; }
static int hf_rounds(int Gp) {
  static const int cap = std::max((int)tuning("head_fused_grid", 256), 1);
  return (hf_units(Gp) + cap - 1) / cap;
}
int head_fused_grid(int Gp) {
  const int rounds = hf_rounds(Gp);
  return (hf_units(Gp) + rounds - 1) / std::max(rounds, 1);
}
// likelihood partials per cell the launch leaves in llk_part
int head_fused_chunks(int Gp) { return head_fused_grid(Gp); }

template <int LK>
static int launch_hf(hipStream_t st, const HeadFusedArgs& a, int grid, int acc, hipEvent_t stop) {
  constexpr int NP = (LK == SMX_LLK_ZINB || LK == SMX_LLK_ZINBD) ? 3 : 2;
  { const int rc = head_fused_prepare(); if (rc != SMX_OK) return rc; }
  const size_t lds = (size_t)4 * 3 * NP * 4096 + (size_t)2 * 8 * 16 * NP * 4;
  // stop: an event that is recorded when THIS launch completes -- as the completion signal of the launch's own packet (hipExtLaunchKernelGGL),
  // not as a marker packet behind it (hipEventRecord), which the next launch of the stream would have to wait for
#define SMX_HF_GO(U, A)                                                                                                                   \
  do {                                                                                                                                    \
    if (stop) hipExtLaunchKernelGGL((head_fused_kernel<LK, U, A>), dim3((unsigned)grid), dim3(512), (uint32_t)lds, st, nullptr, stop, 0u, a); \
    else hipLaunchKernelGGL((head_fused_kernel<LK, U, A>), dim3((unsigned)grid), dim3(512), lds, st, a);                                  \
  } while (0)
  if (a.x_u16) { if (acc) SMX_HF_GO(1, 1); else SMX_HF_GO(1, 0); }
  else { if (acc) SMX_HF_GO(0, 1); else SMX_HF_GO(0, 0); }
#undef SMX_HF_GO
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// every instantiation's dynamic-LDS limit, once per process: at model creation rather than at the first launch, which may sit inside a
// stream capture
int head_fused_prepare() {
  static bool done = false;
  if (done) return SMX_OK;
#define SMX_HF_ATTR2(K, NP) SMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 3 * NP * 4096 + 2 * 8 * 16 * NP * 4))
#define SMX_HF_ATTR(LK, NP)                                                                        \
  SMX_HF_ATTR2((head_fused_kernel<LK, 0, 0>), NP); SMX_HF_ATTR2((head_fused_kernel<LK, 0, 1>), NP);                       \
  SMX_HF_ATTR2((head_fused_kernel<LK, 1, 0>), NP); SMX_HF_ATTR2((head_fused_kernel<LK, 1, 1>), NP)
  SMX_HF_ATTR(SMX_LLK_NB, 2); SMX_HF_ATTR(SMX_LLK_ZINB, 3); SMX_HF_ATTR(SMX_LLK_NBD, 2); SMX_HF_ATTR(SMX_LLK_ZINBD, 3);
#undef SMX_HF_ATTR
#undef SMX_HF_ATTR2
  done = true;
  return SMX_OK;
}

// The fused head of a minibatch of up to SMX_HEAD_FUSED_MAX_CELLS cells: one launch per 128 cells, every launch over the whole panel; the
// launches after the first ADD their dW / db (W_out's tiles are read once per launch, the weight gradient is read and written once more).
// *n_slabs workgroups leave a slab of d d each in a.part ([workgroup][B][128]: a launch fills its cells' rows), *n_sq sum-of-squares slots
// are written (by the last launch: the finished gradient's), llk_part holds head_fused_chunks(Gp) partials per cell.
int launch_head_fused(hipStream_t st, const HeadFusedArgs& a_in, int* n_slabs, int* n_sq, hipEvent_t stop) {
  const int k = llk_planes(a_in.likelihood);
  if (!head_fused_supported(a_in.B, 128, a_in.Gp, k) || !a_in.D || !a_in.W || !a_in.bias || !a_in.X || !a_in.dW || !a_in.db || !a_in.part || !a_in.llk_part ||
      !a_in.dtab || (a_in.ldd % 4) || (a_in.ldw % 4) || (a_in.ldx % 8) || a_in.slab_stride < (long)a_in.B * 128 || (a_in.slab_stride % 4) ||
      false) {
    set_error("head_fused: bad shapes");
    return SMX_ERR_INVALID;
  }
  if (a_in.part_colmajor && (a_in.B > 128 || a_in.slab_stride < 128L * 128)) { set_error("head_fused: column-major slabs take at most 128 cells"); return SMX_ERR_INVALID; }
  const int grid = head_fused_grid(a_in.Gp);
  if (n_sq) *n_sq = grid * 8;
  if (n_slabs) *n_slabs = grid;
  for (int c0 = 0; c0 < a_in.B; c0 += 128) {
    HeadFusedArgs a = a_in;
    a.n_gt = hf_units(a.Gp);
    a.per_wg = hf_rounds(a.Gp);
    a.n_chunks = head_fused_chunks(a.Gp);
    a.B = std::min(128, a_in.B - c0);
    a.D = a_in.D + (long)c0 * a.ldd;
    if (a.rows) a.rows = a_in.rows + c0;
    else a.X = a.x_u16 ? (const void*)(reinterpret_cast<const uint16_t*>(a_in.X) + (long)c0 * a.ldx) : (const void*)(reinterpret_cast<const float*>(a_in.X) + (long)c0 * a.ldx);
    a.part = a_in.part + (long)c0 * 128;
    a.llk_part = a_in.llk_part + (long)c0 * a.n_chunks;
    if (c0 + 128 < a_in.B) a.sq_part = nullptr;   // (only the finished gradient's sum of squares)
    const hipEvent_t ev = (c0 + 128 < a_in.B) ? nullptr : stop;   // (the last launch's completion)
    int rc;
    switch (a.likelihood) {
      case SMX_LLK_NB: rc = launch_hf<SMX_LLK_NB>(st, a, grid, c0 > 0, ev); break;
      case SMX_LLK_ZINB: rc = launch_hf<SMX_LLK_ZINB>(st, a, grid, c0 > 0, ev); break;
      case SMX_LLK_NBD: rc = launch_hf<SMX_LLK_NBD>(st, a, grid, c0 > 0, ev); break;
      case SMX_LLK_ZINBD: rc = launch_hf<SMX_LLK_ZINBD>(st, a, grid, c0 > 0, ev); break;
      default: set_error("head_fused: unknown likelihood"); return SMX_ERR_INVALID;
    }
    if (rc != SMX_OK) return rc;
  }
  return SMX_OK;
}
// ... and the ordered sum of its d d slabs into dd_out [B][128] (smx_bigk.hip's reduce launch)
int launch_head_fused_reduce(hipStream_t st, const HeadFusedArgs& a, int n_slabs, float* dd_out) {
  return launch_bigk_reduce(st, a.part, a.slab_stride, n_slabs, ((long)a.B * 128) >> 2, dd_out);
}
// algorithmic bytes of the head of B cells: W_out and bias read, dW_out and db written (once more each per further 128 cells), the decoder
// output, the counts (as 4-byte values, like the other entries of bench.py's roofline), d d and the likelihood partials
long head_fused_bytes(int B, int G, int Gp, int k) {
  const long passes = (B + 127) / 128, wb = 4L * 128 * k * Gp + 4L * k * Gp;
  return wb * (2 * passes + (passes - 1)) + 4L * B * 128 + 4L * B * G + 4L * B * 128 + 4L * B * head_fused_chunks(Gp);
}

}  // namespace smx
