// smx_headfused.hip -- the WHOLE output head of a training step at a wide gene panel in ONE launch (SURVEY.md 8 rows a-9,
// a-10 / a-11 and the head's part of a-16; BASELINE.json configs[4]: 20 000 genes, 128 cells per GPU and step):
//
//   P = d W_out + b  ->  NB / ZINB / NBD / ZINBD log-likelihood of x, d llk / d P (scaled) = dP
//   dW_out = d^T dP,  db_out = colsum(dP),  sum of squares of dW_out (clipnorm),  d d = dP W_out^T (per-workgroup slabs)
//
// Neither P nor dP exists in memory.  Before: out_head_loss_kernel (P in registers, dP stored: 31 MB) -> bigk_kernel +
// bigk_reduce_kernel (d d: dP and W_out read again) -> panel_dw_kernel (dW: dP read a third time): 35 + 20 + 5 + 18 us at
// 128 x 20 000 with three launch boundaries between them.  Here a workgroup OWNS a tile of 32 genes (all k planes, all <= 128
// cells) from the raw weights to their gradients and walks its tiles; per tile W_out is read once and dW_out written once.
//
// Shape of the work (the flash-attention-backward decomposition, cdna_hip_programming.md "Attention backward": the gene plays
// the key, the cell the query):
//  * 8 waves (2 per SIMD, 256 registers each); wave w owns the cells 16 w .. 16 w + 15 in the forward product and in d d, and
//    the rows 16 w .. 16 w + 15 of H in dW.  All products are v_mfma_f32_16x16x32_bf16 on three-way split operands (six of
//    the nine cross products: f32 accuracy, smx_device.h).
//  * forward with the CELL ON THE LANE: P^T[rho][cell] = sum_h W[h][rho] d[cell][h] (rho = 32 plane + gene of the tile).  A lane
//    then holds, for ONE cell, 8 genes x k planes: the likelihood runs on the accumulators where they are, and its result dP^T
//    is, register for register, the B operand of d d^T[h][cell] += sum_rho W[h][rho] dP^T[rho][cell] (a product that sums over
//    the accumulator's ROW index takes it without any lane movement), accumulated in 32 registers for the whole launch.
//  * dW = d^T dP sums over the cell = the LANE index: dP crosses LDS once, written as the lane's two runs of 4 genes (16 bytes per term) into a
//    [cell][rho] image (three bf16 terms) and read back column-wise by ds_read_b64_tr_b16 (the hardware transpose read).
//  * W_out's tile is split ONCE per workgroup into a [h][rho] bf16 x 3 image: the forward product reads it transposed
//    (ds_read_b64_tr_b16), d d reads its rows.  Both images: 64-byte rows per (term, plane) = 4 slots of 16 bytes holding columns
//    4 s .. 4 s + 3 of both 16-column halves side by side (a lane's two runs of a row read or write are ONE 16-byte access), the slot
//    XOR-ed with row bits (row reads conflict-free, transposed reads 2-way).
//  * d's operands never change: a wave's two views of d (its cells' rows for the forward product, its H rows' columns for dW) are
//    split once; the first stays in 48 registers, the second is parked in an L2-resident table and fetched per tile (below); no LDS
//    image of d (144 KB of LDS are the two images above).
//  * the k index of every product is permuted consistently in both operands (lane group g, element e <-> k = 4 g + e for
//    e < 4, 16 + 4 g + e - 4 otherwise) so that the transposed reads of a 32-lane half touch 8 consecutive rows.
//  * 2 workgroup barriers per tile; the next tile's W (global -> registers) and counts are in flight under the current tile.
//  * the per-wave queue of the non-zero counts (smx_loss.h) lives in the wave's own rows of the dP image.
// Deterministic: no atomics; per-workgroup d d slabs are summed by bigk_reduce_kernel in workgroup order.
#include <stdlib.h>

#include <algorithm>

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

// the wave's queue of non-zero counts (smx_loss.h) inside rows 16 w .. 16 w + 15 of the dP image's first four sub-images
struct HfQueue {
  unsigned char* base;
  __device__ float2& operator[](int k) const { return *reinterpret_cast<float2*>(base + (k >> 7) * 8192 + (k & 127) * 8); }
  __device__ explicit operator bool() const { return true; }
};

// VW: the eight 16-cell / 16-row shares of a workgroup are carried by 8 / VW hardware waves.
// What decides it is the register file: hipcc gives the MFMAs' A / B operands vector registers only (the accumulator half of the
// file takes accumulators), so the two resident views of d (96 registers per share) sit in the half that the likelihood needs.
//   VW = 2 (256 threads, one wave per SIMD, 512 registers): both views resident; measured 103 us at 128 x 20 000 zinb (117 registers
//   in scratch, every LDS round trip and every dependent MFMA chain exposed -- nothing else runs on the SIMD).
//   VW = 1 (512 threads, two waves per SIMD, 256 registers): the view for dW is resident only while that product runs.  Each wave writes
//   it once to a table in global memory (a.dtab, 96 KB, the same bytes from every workgroup: L2-resident) and reads it back -- 12
//   coalesced 16-byte loads -- one phase before it is needed, ahead of the next tile's W so that the wait for it does not also wait
//   for that (vector-memory results return in order).  (Both views that way: +1-2 us -- 123 MB of L2 reads per launch.)
#ifndef SMX_HF_STAMP_WAVE
#define SMX_HF_STAMP_WAVE 0
#endif
template <int LK, int U16, int VW>
__global__ __launch_bounds__(512 / VW, 2 / VW) void head_fused_kernel(HeadFusedArgs a) {
  constexpr int NP = (LK == SMX_LLK_ZINB || LK == SMX_LLK_ZINBD) ? 3 : 2;
  constexpr int NSUB = 2 * NP;             // 16-column groups of a tile's rho axis
  constexpr int IMG = 3 * NP * 8192;       // bytes of one image (three terms x NP planes x [128][32] bf16)
  constexpr int T = 512 / VW;              // threads
  constexpr int RPT = T / 8;               // W rows one pass of the workgroup's float4 loads covers (8 threads per 128-byte row segment)
  constexpr int UPP = 128 / RPT;           // passes per plane
  constexpr int NWL = NP * UPP;            // float4 of W per thread and tile
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // lds + 0: the W image [term][plane][h 128][32]
  unsigned char* const Pimg = lds + IMG;                // [term][plane][cell 128][32]
  float* const dbx = reinterpret_cast<float*>(lds + 2 * IMG);   // [8 shares][32 NP]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;               // column of a 16 x 16 tile | lane group (k octet / row quad)
  const int q4 = j >> 2, pp = j & 3;                    // transposed read: this lane addresses row q4, slot pp of its group's block
  // Lane parts of every LDS address, formed once; everything else of an address is a compile-time constant (an immediate
  // of the ds instruction).  The swizzle's row bits are lane bits in every access shape:
  //   transposed reads: row = 32 ks + 16 rd + 4 g + q4 -> bits 2, 3 of the row are bits 0, 1 of g
  //   row reads of d d / the dP writes: row = 16 hs + j (16 w + j) -> bits 2, 3 of j
  //   the W image's writes: row = RPT u + (tid >> 3) -> bits 2, 3 of tid >> 3
  // ... and every base is made OPAQUE to the optimiser (hf_hide): knowing that a base's bits below 1024 are the only ones set it turns
  // base + 8192 n into base | 8192 n, no longer folds the constant into the instruction's offset field, and keeps one register per
  // distinct address instead (~100 of them, spilled around the tile loop).  The offset field holds 16 bits: two bases per image half
  // (sub-images 0..3 and 4..8).
  int tb[2][2][2], rb[2][2][2];   // transposed reads / row accesses: [image: W, dP][sub-images 4.. ][column half]
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    // (a row = 64 bytes = 4 slots of 16 bytes; slot s holds columns 4 s .. 4 s + 3 of BOTH halves side by side, so that a lane's two runs of
    // a row read are ONE 16-byte access; the slot is XOR-ed with f(row >> 2), f = 0, 2, 3, 1: every 16-lane group of ds_read_b128 then
    // touches 16 different slots; the transposed reads are 2-way on it -- rows r and r + 4 of a half-wave share a bank pair)
    const int t = 64 * (4 * g + q4) + 16 * (pp ^ ((0x78 >> (2 * g)) & 3)) + 8 * hf;
    const int r = 64 * j + 16 * (g ^ ((0x78 >> (2 * ((j >> 2) & 3))) & 3));
#pragma unroll
    for (int im = 0; im < 2; ++im)
#pragma unroll
      for (int hi = 0; hi < 2; ++hi) { tb[im][hi][hf] = hf_hide(t + im * IMG + hi * 32768); rb[im][hi][hf] = hf_hide(r + im * IMG + hi * 32768); }
  }
  const int wt = tid >> 3, wc4 = tid & 7;
  int wwb[2];
#pragma unroll
  for (int hi = 0; hi < 2; ++hi) wwb[hi] = hf_hide(64 * wt + 16 * ((wc4 & 3) ^ ((0x78 >> (2 * ((wt >> 2) & 3))) & 3)) + 8 * (wc4 >> 2) + hi * 32768);
  const int wgo = (int)((wt * a.ldw + 4 * wc4) * 4);   // this thread's float4 of a W tile (bytes): rows wt, wt + RPT, ... of every plane
  const long wbytes = 128L * a.ldw * 4;
  const __amdgpu_buffer_rsrc_t rW = hf_rsrc(a.W, wbytes), rdW = hf_rsrc(a.dW, wbytes);
  const __amdgpu_buffer_rsrc_t rbias = hf_rsrc(a.bias, a.ldw * 4), rdb = hf_rsrc(a.db, a.ldw * 4);
  const __amdgpu_buffer_rsrc_t rllk = hf_rsrc(a.llk_part, (long)a.B * a.n_gt * 4), rtab = hf_rsrc(a.dtab, SMX_HEAD_FUSED_TAB_BYTES);

  // ---- loads of a tile: this thread's NWL float4 of W (row-major segments of 128 bytes), the lanes' counts ---------------
  float4 wreg[NWL];
  auto load_w = [&](int tile) {
#pragma unroll
    for (int u = 0; u < NWL; ++u)   // u = UPP plane + pass
      wreg[u] = hf_load4(rW, wgo, (int)((tile * 32 + (long)(RPT * (u % UPP)) * a.ldw + (long)(u / UPP) * a.Gp) * 4));
  };
  int tile = blockIdx.x;
  load_w(tile);   // (requested before the prologue: the first tile's HBM latency runs under the split of d)
#ifdef SMX_HF_STAMPS
  if (a.dbg && tid == 64 * SMX_HF_STAMP_WAVE && (blockIdx.x == 0 || blockIdx.x == 100)) a.dbg[(blockIdx.x ? 64 : 0) + 63] = clock64();   // kernel entry
#endif

  // ---- both views of d per share, split once -----------------------------------------------------------------------------
  // dB[v][ks]: B operand of the forward product, d[cell][k], k = 32 ks + (e < 4 ? 4 g + e : 16 + 4 g + e - 4)
  // dA[v][ks]: A operand of dW, d[k = cell'][h = 16 w + j] over the same k order
  Split8 dB[VW][4], dA[VW][4];
  hf_f32x4 accDD[VW][8];
  int cell[VW], dw_vo[VW]; bool cell_ok[VW]; long src[VW];
#pragma unroll
  for (int v = 0; v < VW; ++v) {
    const int w = VW * wave + v;
    cell[v] = 16 * w + j;                               // forward / d d: the lane's cell
    dw_vo[v] = (int)(((16 * w + 4 * g) * a.ldw + j) * 4);   // dW: row 16 w + 4 g (+ r), column j (+ the tile's) in bytes
    const int cellc = min(cell[v], a.B - 1);
    cell_ok[v] = cell[v] < a.B;
    src[v] = a.rows ? (long)a.rows[cellc] : (long)cellc;
    const float* dp = a.D + (long)cellc * a.ldd;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float4 lo = *reinterpret_cast<const float4*>(dp + 32 * ks + 4 * g);
      const float4 hi = *reinterpret_cast<const float4*>(dp + 32 * ks + 16 + 4 * g);
      const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      dB[v][ks] = split3x8(x);
    }
    const float* da = a.D + 16 * w + j;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = 32 * ks + (e < 4 ? 4 * g + e : 16 + 4 * g + e - 4);
        x[e] = da[(long)min(c, a.B - 1) * a.ldd];       // (a cell beyond the minibatch meets dP = 0)
      }
      dA[v][ks] = split3x8(x);
    }
#pragma unroll
    for (int hs = 0; hs < 8; ++hs) accDD[v][hs] = hf_f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // the table of the view for dW: [wave 8][ks 4][term 3][lane 64] x 16 bytes
  constexpr bool RELOAD = VW == 1 && LK != SMX_LLK_NB;   // (two planes, the lighter likelihood: both views fit -- 43.6 -> 41.0 us at 128 x 20 000)
  const int tab_vo = lane * 16, tabA = wave * 12 * 1024;   // (scalar byte offset of the wave's entries)
  if (RELOAD) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      hf_store8h(dA[0][ks].t0, rtab, tab_vo, tabA + (3 * ks + 0) * 1024); hf_store8h(dA[0][ks].t1, rtab, tab_vo, tabA + (3 * ks + 1) * 1024); hf_store8h(dA[0][ks].t2, rtab, tab_vo, tabA + (3 * ks + 2) * 1024);
    }
    // the wave's own table entries are written before it reads them back.  The stored registers are operands of the wait: measured on
    // MI355X, a 16-byte buffer store whose data registers are reused by the instructions right behind it (they are dead once stored)
    // wrote garbage -- with the registers held until the stores have completed the table is right (tools/headfused_try.py)
#define HF_V(x) "v"(x.t0), "v"(x.t1), "v"(x.t2)
    asm volatile("s_waitcnt vmcnt(0)" :: HF_V(dA[0][0]), HF_V(dA[0][1]), HF_V(dA[0][2]), HF_V(dA[0][3]) : "memory");
#undef HF_V
  }
  // (the lane offset goes through hf_hide at every call: a load from a loop-invariant address is otherwise hoisted out of the tile loop --
  // the view would be resident again, and requested right behind its own stores)
  auto load_view = [&](int tab, Split8 (&dst)[4]) {
    const int vo = hf_hide(tab_vo);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      dst[ks].t0 = hf_load8h(rtab, vo, tab + (3 * ks + 0) * 1024); dst[ks].t1 = hf_load8h(rtab, vo, tab + (3 * ks + 1) * 1024); dst[ks].t2 = hf_load8h(rtab, vo, tab + (3 * ks + 2) * 1024);
    }
  };
  float ssq = 0.f;
  // `gate` is always true, and opaque: HF_STAMP() at the seam of two phases is a (never taken) branch on it, which makes every phase a
  // basic block of its own.  Without real block boundaries there the compiler merges the phases of a tile into one scheduling region --
  // across sched_barrier(0) too -- to the point of ~70 registers in scratch, whose reloads (a memory round trip each, with two waves
  // per SIMD to hide it) were half of a tile's time.
  const bool gate = __builtin_amdgcn_readfirstlane(hf_hide(1)) != 0;
#ifdef SMX_HF_STAMPS   // (development: cycle stamps of the phases of workgroups 0 and 100, read back by smx_k_head_fused under the knob hf_dbg)
  int dbg_n = 0;
#define HF_STAMP() do { if (a.dbg && tid == 64 * SMX_HF_STAMP_WAVE && (blockIdx.x == 0 || blockIdx.x == 100)) { a.dbg[(blockIdx.x ? 64 : 0) + dbg_n] = clock64(); ++dbg_n; } } while (0)
#else
#define HF_STAMP() do { if (!gate) asm volatile("s_nop 0"); } while (0)
#endif
  HF_STAMP();

  uint2 xraw16[VW][2]; float4 xraw32[VW][2];
  auto load_x = [&](int tile) {
#pragma unroll
    for (int v = 0; v < VW; ++v)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const long o = src[v] * a.ldx + (long)tile * 32 + 16 * hf + 4 * g;
        if (U16) xraw16[v][hf] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(a.X) + o);
        else xraw32[v][hf] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.X) + o);
      }
  };

  // operand group n = NSUB ks + s of a transposed sweep over image im: rows (k) 32 ks + 16 rd + 4 g + q4, columns of group s
  auto tr_read = [&](int im, int n) {
    const int ks = n / NSUB, sb = n % NSUB, p = sb >> 1, hf = sb & 1;
    Split8 r;
    r.t0 = cat8(as_bf(__builtin_amdgcn_ds_read_tr16_b64_v4i16(HF_LDS3(HF_AT(tb[im], hf, p, 2048 * ks)))), as_bf(__builtin_amdgcn_ds_read_tr16_b64_v4i16(HF_LDS3(HF_AT(tb[im], hf, p, 2048 * ks + 1024)))));
    r.t1 = cat8(as_bf(__builtin_amdgcn_ds_read_tr16_b64_v4i16(HF_LDS3(HF_AT(tb[im], hf, NP + p, 2048 * ks)))), as_bf(__builtin_amdgcn_ds_read_tr16_b64_v4i16(HF_LDS3(HF_AT(tb[im], hf, NP + p, 2048 * ks + 1024)))));
    r.t2 = cat8(as_bf(__builtin_amdgcn_ds_read_tr16_b64_v4i16(HF_LDS3(HF_AT(tb[im], hf, 2 * NP + p, 2048 * ks)))), as_bf(__builtin_amdgcn_ds_read_tr16_b64_v4i16(HF_LDS3(HF_AT(tb[im], hf, 2 * NP + p, 2048 * ks + 1024)))));
    return r;
  };
  // operand group n = 8 p + hs of d d: rows 16 hs + j of the W image, the lane group's two 8-byte runs of plane p
  auto row_read = [&](int n) {
    const int p = n / 8, hs = n % 8;
    Split8 r;
    r.t0 = *reinterpret_cast<const smx_bf16x8*>(HF_AT(rb[0], 0, p, 1024 * hs));
    r.t1 = *reinterpret_cast<const smx_bf16x8*>(HF_AT(rb[0], 0, NP + p, 1024 * hs));
    r.t2 = *reinterpret_cast<const smx_bf16x8*>(HF_AT(rb[0], 0, 2 * NP + p, 1024 * hs));
    return r;
  };

  load_x(tile);
  for (; tile < a.n_gt; tile += gridDim.x) {
    const int n0 = tile * 32;
    HF_STAMP();   // top
    // biases of the lane's 8 genes: the accumulators of the forward product start from them.  (Requested here, after the wait for
    // W's registers and ahead of the next tile's counts: vector-memory results return in order, so a wait for the biases also waits
    // for everything requested before them.)
    float4 bq[NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) bq[p][hf] = hf_load4(rbias, 16 * g, (p * a.Gp + n0 + 16 * hf) * 4);
    // ---- W tile -> bf16 x 3 image (the previous tile's readers are past their last barrier) ---------------------------
#pragma unroll
    for (int u = 0; u < NWL; ++u) {
      const float x[4] = {wreg[u].x, wreg[u].y, wreg[u].z, wreg[u].w};
      const Split4 s = split3x4(x);
      const int pl = u / UPP, more = 64 * RPT * (u % UPP);
      *reinterpret_cast<hf_bf16x4*>(lds + wwb[(pl) >= 4] + (pl - 4 * (pl >= 4)) * 8192 + more) = s.t0;
      *reinterpret_cast<hf_bf16x4*>(lds + wwb[(NP + pl) >= 4] + (NP + pl - 4 * ((NP + pl) >= 4)) * 8192 + more) = s.t1;
      *reinterpret_cast<hf_bf16x4*>(lds + wwb[(2 * NP + pl) >= 4] + (2 * NP + pl - 4 * ((2 * NP + pl) >= 4)) * 8192 + more) = s.t2;
    }
    // this tile's counts out of their raw registers
    float xs[VW][8];
#pragma unroll
    for (int v = 0; v < VW; ++v)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        if (U16) {
          xs[v][4 * hf] = (float)(xraw16[v][hf].x & 0xFFFFu); xs[v][4 * hf + 1] = (float)(xraw16[v][hf].x >> 16);
          xs[v][4 * hf + 2] = (float)(xraw16[v][hf].y & 0xFFFFu); xs[v][4 * hf + 3] = (float)(xraw16[v][hf].y >> 16);
        } else {
          xs[v][4 * hf] = xraw32[v][hf].x; xs[v][4 * hf + 1] = xraw32[v][hf].y; xs[v][4 * hf + 2] = xraw32[v][hf].z; xs[v][4 * hf + 3] = xraw32[v][hf].w;
        }
      }
    HF_STAMP();   // W written
    __syncthreads();   // (A) the W image is complete; every wave has left the previous tile's dW (the dP image, dbx are free)
    HF_STAMP();   // barrier A passed
    const int next = tile + gridDim.x;

#pragma unroll
    for (int v = 0; v < VW; ++v) {
      const int w = VW * wave + v;
      // ---- forward: P^T[rho][cell], A = W^T read transposed from the image, B = the share's rows of d -----------------
      hf_f32x4 accP[NSUB];
#pragma unroll
      for (int s = 0; s < NSUB; ++s) accP[s] = hf_f32x4{bq[s >> 1][s & 1].x, bq[s >> 1][s & 1].y, bq[s >> 1][s & 1].z, bq[s >> 1][s & 1].w};
      if (v == 0 && next < a.n_gt) load_x(next);
      // (one wave per SIMD: nothing hides an LDS round trip but this wave's own MFMAs -- the operands of step n + 1 are requested
      // before the products of step n are issued)
      {
        Split8 cur = tr_read(0, 0), nxt = cur;
#pragma unroll
        for (int n = 0; n < 4 * NSUB; ++n) {
          if (n + 1 < 4 * NSUB) nxt = tr_read(0, n + 1);
          accP[n % NSUB] = mfma16_bf16x3(cur, dB[v][n / NSUB], accP[n % NSUB]);
          cur = nxt;
          __builtin_amdgcn_sched_barrier(0);
        }
      }

      HF_STAMP();   // fwd done
      // ---- likelihood on the accumulators: element e = 4 half + r is gene n0 + 16 half + 4 g + r of the lane's cell ---
      float lsum = 0.f;
      float dpv[NP][8];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {   // (four elements at a time)
        float x4[4], v0[4], v1[4], v2[4], llk[4], d0[4], d1[4], d2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          x4[r] = xs[v][4 * hf + r];
          v0[r] = accP[hf][r];
          v1[r] = accP[2 + hf][r];
          v2[r] = NP == 3 ? accP[2 * (NP - 1) + hf][r] : 0.f;
        }
        if (!gate) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { llk[r] = v0[r] * x4[r]; d0[r] = v0[r]; d1[r] = v1[r]; d2[r] = v2[r]; }
        } else
        count_elem_vec<LK, 0, 4>(x4, v0, v1, v2, llk, d0, d1, d2, HfQueue{Pimg + 1024 * w});
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = cell_ok[v] && (n0 + 16 * hf + 4 * g + r) < a.G;
          lsum += ok ? llk[r] : 0.f;
          dpv[0][4 * hf + r] = ok ? d0[r] * a.grad_scale : 0.f;
          dpv[1][4 * hf + r] = ok ? d1[r] * a.grad_scale : 0.f;
          if (NP == 3) dpv[NP - 1][4 * hf + r] = ok ? d2[r] * a.grad_scale : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (v == VW - 1 && RELOAD) {
        load_view(tabA, dA[0]);
      }
      HF_STAMP();   // likelihood done
      // per-cell partial of the tile: the four lane groups hold its 32 genes
      lsum += __shfl_xor(lsum, 16, 64);
      lsum += __shfl_xor(lsum, 32, 64);
      if (g == 0 && cell_ok[v]) hf_store1(lsum, rllk, cell[v] * a.n_gt * 4, tile * 4);

      // ---- dP, plane by plane: bias-gradient partials over the share's 16 cells, the [cell][rho] image, d d^T += W dP^T --------
      // (A = W[h = 16 hs + j][rho = 32 p + (e < 4 ? 4 g + e : 16 + 4 g + e - 4)]: two 8-byte row reads per term; B = the split plane)
      Split8 cur = row_read(0), nxt = cur;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = dpv[p][e];
#define HF_DPP_ADD(ctrl) t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), ctrl, 0xF, 0xF, false))
          HF_DPP_ADD(0xB1); HF_DPP_ADD(0x4E); HF_DPP_ADD(0x141); HF_DPP_ADD(0x140);   // sum over the 16 lanes of the row
#undef HF_DPP_ADD
          if (j == 0) dbx[w * 32 * NP + 32 * p + 16 * (e >> 2) + 4 * g + (e & 3)] = t;
        }
        const Split8 sp = split3x8(dpv[p]);
        {
          const int wo = 1024 * w;   // (the share's rows: a scalar)
          *reinterpret_cast<smx_bf16x8*>(HF_AT(rb[1], 0, p, 0) + wo) = sp.t0;
          *reinterpret_cast<smx_bf16x8*>(HF_AT(rb[1], 0, NP + p, 0) + wo) = sp.t1;
          *reinterpret_cast<smx_bf16x8*>(HF_AT(rb[1], 0, 2 * NP + p, 0) + wo) = sp.t2;
        }
#pragma unroll
        for (int hs = 0; hs < 8; ++hs) {
          const int n = 8 * p + hs;
          if (n + 1 < 8 * NP) nxt = row_read(n + 1);
          accDD[v][hs] = mfma16_bf16x3(cur, sp, accDD[v][hs]);
          cur = nxt;
          __builtin_amdgcn_sched_barrier(0);
        }
        HF_STAMP();
      }
    }
    HF_STAMP();   // dd done
    if (next < a.n_gt) load_w(next);   // (as late as dW still covers it: its 24 registers are not free before)
    __syncthreads();   // (B) the dP image and the bias partials are complete; every wave is done with the W image
    HF_STAMP();   // barrier B passed

    // ---- dW[h = 16 w + 4 g + r][rho] = sum over the cells: A = the share's columns of d, B = dP read transposed ----------
#pragma unroll
    for (int v = 0; v < VW; ++v) {
      (void)wave;
      hf_f32x4 accW[NSUB];
#pragma unroll
      for (int s = 0; s < NSUB; ++s) accW[s] = hf_f32x4{0.f, 0.f, 0.f, 0.f};
      {
        Split8 cur = tr_read(1, 0), nxt = cur;
#pragma unroll
        for (int n = 0; n < 4 * NSUB; ++n) {
          if (n + 1 < 4 * NSUB) nxt = tr_read(1, n + 1);
          accW[n % NSUB] = mfma16_bf16x3(dA[v][n / NSUB], cur, accW[n % NSUB]);
          cur = nxt;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int s = 0; s < NSUB; ++s) {
        const int p = s >> 1, hf = s & 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = accW[s][r];
          hf_store1(x, rdW, dw_vo[v], (int)((r * a.ldw + (long)p * a.Gp + n0 + 16 * hf) * 4));
          ssq += x * x;
        }
      }
    }
    HF_STAMP();   // dW + stores done
    if (tid < 32 * NP) {   // bias gradient: the eight shares' partials in order
      float t = dbx[tid];
#pragma unroll
      for (int ww = 1; ww < 8; ++ww) t += dbx[ww * 32 * NP + tid];
      hf_store1(t, rdb, (tid & 31) * 4, ((tid >> 5) * a.Gp + n0) * 4);
    }
  }

  HF_STAMP();
  // ---- this workgroup's slab of d d: accumulator register r of tile hs is h = 16 hs + 4 g + r of the lane's cell -----------
#pragma unroll
  for (int v = 0; v < VW; ++v)
    if (cell_ok[v]) {
      float* op = a.part + (long)blockIdx.x * a.slab_stride + (long)cell[v] * 128 + 4 * g;
#pragma unroll
      for (int hs = 0; hs < 8; ++hs) *reinterpret_cast<float4*>(op + 16 * hs) = make_float4(accDD[v][hs][0], accDD[v][hs][1], accDD[v][hs][2], accDD[v][hs][3]);
    }
  if (a.sq_part) {
    ssq = wave_sum(ssq);
    if (lane == 0) a.sq_part[(long)blockIdx.x * (8 / VW) + wave] = ssq;
  }
}

bool head_fused_supported(int B, int Hp, int Gp, int k) {
  return B > 0 && B <= 128 && Hp == 128 && Gp % 32 == 0 && Gp >= SMX_HEAD_FUSED_MIN_GENES && (k == 2 || k == 3) && !tuning_on("no_head_fused");
}
// workgroups: one per CU at most, every one with the same number of tiles (+- 1)
int head_fused_grid(int Gp) {
  static const int cap = std::max((int)tuning("head_fused_grid", 256), 1);
  const int tiles = Gp / 32, rounds = (tiles + cap - 1) / cap;
  return (tiles + rounds - 1) / std::max(rounds, 1);
}

#define SMX_HF_VW 1   // one share per hardware wave, two waves per SIMD (the kernel's comment)
template <int LK>
static int launch_hf(hipStream_t st, const HeadFusedArgs& a, int grid) {
  constexpr int NP = (LK == SMX_LLK_ZINB || LK == SMX_LLK_ZINBD) ? 3 : 2;
  const size_t lds = (size_t)2 * 3 * NP * 8192 + (size_t)8 * 32 * NP * 4;
  { const int rc = head_fused_prepare(); if (rc != SMX_OK) return rc; }
  if (a.x_u16) hipLaunchKernelGGL((head_fused_kernel<LK, 1, SMX_HF_VW>), dim3((unsigned)grid), dim3(512 / SMX_HF_VW), lds, st, a);
  else hipLaunchKernelGGL((head_fused_kernel<LK, 0, SMX_HF_VW>), dim3((unsigned)grid), dim3(512 / SMX_HF_VW), lds, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// every instantiation's dynamic-LDS limit, once per process: at model creation rather than at the first launch, which may sit inside a
// stream capture
int head_fused_prepare() {
  static bool done = false;
  if (done) return SMX_OK;
#define SMX_HF_ATTR(LK, NP)                                                                                                                         \
  SMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&head_fused_kernel<LK, 0, SMX_HF_VW>), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                              2 * 3 * NP * 8192 + 8 * 32 * NP * 4));                                                                                \
  SMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&head_fused_kernel<LK, 1, SMX_HF_VW>), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                              2 * 3 * NP * 8192 + 8 * 32 * NP * 4))
  SMX_HF_ATTR(SMX_LLK_NB, 2); SMX_HF_ATTR(SMX_LLK_ZINB, 3); SMX_HF_ATTR(SMX_LLK_NBD, 2); SMX_HF_ATTR(SMX_LLK_ZINBD, 3);
#undef SMX_HF_ATTR
  done = true;
  return SMX_OK;
}

// the fused launch; *n_slabs workgroups leave a slab of d d each in a.part, *n_sq sum-of-squares slots are written
int launch_head_fused(hipStream_t st, const HeadFusedArgs& a_in, int* n_slabs, int* n_sq) {
  HeadFusedArgs a = a_in;
  const int k = llk_planes(a.likelihood);
  if (!head_fused_supported(a.B, 128, a.Gp, k) || !a.D || !a.W || !a.bias || !a.X || !a.dW || !a.db || !a.part || !a.llk_part || !a.dtab ||
      (a.ldd % 4) || (a.ldw % 4) || (a.ldx % 8) || a.slab_stride < (long)a.B * 128 || (a.slab_stride % 4)) {
    set_error("head_fused: bad shapes");
    return SMX_ERR_INVALID;
  }
  a.n_gt = a.Gp / 32;
  const int grid = head_fused_grid(a.Gp);
  if (n_sq) *n_sq = grid * (8 / SMX_HF_VW);
  if (n_slabs) *n_slabs = grid;
  switch (a.likelihood) {
    case SMX_LLK_NB: return launch_hf<SMX_LLK_NB>(st, a, grid);
    case SMX_LLK_ZINB: return launch_hf<SMX_LLK_ZINB>(st, a, grid);
    case SMX_LLK_NBD: return launch_hf<SMX_LLK_NBD>(st, a, grid);
    case SMX_LLK_ZINBD: return launch_hf<SMX_LLK_ZINBD>(st, a, grid);
    default: set_error("head_fused: unknown likelihood"); return SMX_ERR_INVALID;
  }
}
// ... and the ordered sum of its d d slabs into dd_out [B][128] (smx_bigk.hip's reduce launch)
int launch_head_fused_reduce(hipStream_t st, const HeadFusedArgs& a, int n_slabs, float* dd_out) {
  return launch_bigk_reduce(st, a.part, a.slab_stride, n_slabs, ((long)a.B * 128) >> 2, dd_out);
}
// algorithmic bytes of one launch: W_out and bias read, dW_out and db written, the decoder output, the counts (as 4-byte values, like
// the other entries of bench.py's roofline), d d and the likelihood partials
long head_fused_bytes(int B, int G, int Gp, int k) {
  return 2L * (4L * 128 * k * Gp + 4L * k * Gp) + 4L * B * 128 + 4L * B * G + 4L * B * 128 + 4L * B * (Gp / 32);
}

}  // namespace smx
