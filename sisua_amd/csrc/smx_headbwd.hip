// smx_headbwd.hip -- the two backward products of the output head in ONE launch (count heads with raw parameter
// planes; SURVEY.md 8 row a-16, the largest weight gradient):
//
//   role 0   dW_out = d^T dP   [H][k Gp]   (+ db = column sums of dP, + per-tensor sum of squares for clipnorm)
//   role 1   dd     = dP W_out^T  [B][H]   as split-K slabs over the k Gp axis (summed by the BatchNorm-backward launch)
//
// Same shape of work as the fused forward (smx_headloss.hip): 8 waves per workgroup split K, every operand goes
// straight from global memory into MFMA operand registers with ALL loads of a K slab in flight (the k index of
// v_mfma_f32_32x32x2_f32 is free to permute), the 8 partial tiles meet in LDS and every wave finishes two
// accumulator registers.  The grouped LDS-tiled form this replaces took 14.1 us for 0.39 GFLOP and 12 MB: its
// workgroups walked K in serial load -> LDS -> barrier -> MFMA rounds at two workgroups per CU.
//   role 0: tile = 32 rows of H x 32 genes x k planes; K = the minibatch (16 cells per wave and 128-cell slab: lane
//           (i, hh) supplies cells 8 hh + s); both operands are coalesced 128-byte row segments (d[cell][h0 + i],
//           dP[cell][g0 + i]); the (H / 32) workgroups of a gene tile are 8 blocks apart (same XCD).
//   role 1: tile = 32 cells x 32 columns of H; K = this slice of the k Gp axis (64 per wave and 512-wide slab: lane
//           (i, hh) supplies 32 consecutive k: one whole 128-byte line of row i of dP and of row h0 + i of W_out).
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "smx_internal.h"
#include "smx_panel.h"
#include "smx_dgemm.h"
#include "../../include/sisua_hip.h"

namespace smx {
SMX_STAMP_TABLE


typedef float f32x16 __attribute__((ext_vector_type(16)));

// SEP: the planes are separate tensors (scvi's heads: W_p [Hp][Gp] each, their own bias / clipnorm)
// B3: both products from bf16 MFMAs on three-way split operands (smx_device.h), split in registers after the loads: in
// either role a lane's loads are runs of 8 consecutive k, i.e. whole operands of v_mfma_f32_32x32x16_bf16
template <int NP, int SEP = 0, int B3 = 0>
__global__ __launch_bounds__(512) void out_head_bwd_kernel(HeadBwdArgs a) {
  // ONE plane's eight partial tiles at a time (32 KB); the bf16 form's role 1 turns its operands through wave-private tiles
  // first (2 x 8 x 32 x 36 floats = 72 KB, the partial tiles then go over them): two workgroups per CU either way
  __shared__ __attribute__((aligned(16))) float red[B3 ? 2 * 8 * 32 * 36 : 2 * 8 * 1024];   // (>= two planes' eight partial tiles)
  preload(a.D, a.ldd, a.dP, a.ldp, a.W, a.ldw, a.dW, a.db, a.slab, a.slab_stride, a.B, a.Hp, a.Gp, a.n_slices, a.k_chunk, a.sq_part,
          a.n_w, a.n_ht, a.n_gt, a.n_ct, a.n_extra, a.diag, a.dd_colmajor, a.skip_dd);   // (the argument fields in one batch: smx_device.h)
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = lane & 31, hh = lane >> 5;
  // accumulator register r of a 32 x 32 tile is row (r & 3) + 8 (r >> 2) + 4 hh, column i; wave q finishes r = 2q, 2q + 1
  int rowof[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int r = 2 * q + j; rowof[j] = (r & 3) + 8 * (r >> 2) + 4 * hh; }

  if ((int)blockIdx.x < a.n_w) {
    // ======================= role 0: dW tile (+ db, + sum of squares) ==========================================
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int ht = idx % a.n_ht, gt = (idx / a.n_ht) * 8 + xcd;
    if (gt >= a.n_gt || (a.diag & 1)) return;
    const int h0 = ht * 32, g0 = gt * 32;
    f32x16 acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    float csum[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) csum[p] = 0.f;
    SMX_STAMP(4, 0);   // entry (role 0: a dW tile)
    for (int kc = 0; kc < a.B; kc += 128) {
      const int k0 = kc + 16 * q + 8 * hh;
      if (kc + 16 * q >= a.B) break;   // wave-uniform
      float av[8], bv[NP][8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int cell = min(k0 + s, a.B - 1);
        av[s] = a.D[(long)cell * a.ldd + h0 + i];
#pragma unroll
        for (int p = 0; p < NP; ++p) bv[p][s] = a.dP[(long)cell * a.ldp + (long)p * a.Gp + g0 + i];
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (B3) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const bool on = k0 + s < a.B;     // K is the (ragged) minibatch axis: cells beyond it contribute nothing
          av[s] = on ? av[s] : 0.f;
#pragma unroll
          for (int p = 0; p < NP; ++p) { bv[p][s] = on ? bv[p][s] : 0.f; csum[p] += bv[p][s]; }
        }
        const Split8 sa = split3x8(av);
#pragma unroll
        for (int p = 0; p < NP; ++p) acc[p] = mfma_bf16x3(sa, split3x8(bv[p]), acc[p]);
      } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const bool on = k0 + s < a.B;       // K is the (ragged) minibatch axis: cells beyond it contribute nothing
          const float av_s = on ? av[s] : 0.f;
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            const float b = on ? bv[p][s] : 0.f;
            csum[p] += b;
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av_s, b, acc[p], 0, 0, 0);
          }
        }
      }
    }
    SMX_STAMP(4, 1);   // the products over the minibatch (loads in flight included)
    float sq = 0.f;
    // planes 0 and 1 go through LDS TOGETHER (two slots of eight partial tiles), the third plane behind them: three barriers for three planes
    // (one for two) instead of five (three); the partial tiles of a plane are added in the same order as one plane at a time
    auto park = [&](int p, float* slot) {
#pragma unroll
      for (int r = 0; r < 16; ++r) slot[(q * 16 + r) * 64 + lane] = acc[p][r];
    };
    auto finish = [&](int p, const float* slot) {
      if (SEP) sq = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = 2 * q + j;
        float t = slot[(0 * 16 + r) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) t += slot[(w * 16 + r) * 64 + lane];
        const int h = h0 + rowof[j];
        if (SEP) a.dWp[p][(long)h * a.ldw + g0 + i] = t;
        else a.dW[(long)h * a.ldw + (long)p * a.Gp + g0 + i] = t;   // rows >= H and columns >= G are zero by construction
        sq += t * t;
      }
      if (SEP && a.sqp[p]) {
        const float sw = wave_sum(sq);
        if (lane == 0) a.sqp[p][((long)ht * a.n_gt + gt) * 8 + q] = sw;
      }
    };
    park(0, red); park(1, red + 8 * 1024);
    __syncthreads();
    finish(0, red); finish(1, red + 8 * 1024);
    if constexpr (NP == 3) {
      __syncthreads();
      park(2, red);
      __syncthreads();
      finish(2, red);
    }
    if (!SEP && a.sq_part) {
      sq = wave_sum(sq);
      if (lane == 0) a.sq_part[((long)ht * a.n_gt + gt) * 8 + q] = sq;   // 8 slots per (H tile, gene tile): <= 4 per 32 x 32 tile
    }
    if (ht == 0) {   // bias gradient: column sums of dP over the whole minibatch (each lane holds its K slice's part)
      __syncthreads();
#pragma unroll
      for (int p = 0; p < NP; ++p) red[(q * NP + p) * 64 + lane] = csum[p];
      __syncthreads();
      if (q < NP && lane < 32) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[(w * NP + q) * 64 + lane] + red[(w * NP + q) * 64 + 32 + lane];
        if (SEP) {   // (q is wave-uniform; the pointer array is indexed with constants only)
          float* dbq = q == 0 ? a.dbp[0] : (q == 1 ? a.dbp[1] : a.dbp[2]);
          dbq[g0 + lane] = t;
        } else a.db[(long)q * a.Gp + g0 + lane] = t;
      }
    }
    SMX_STAMP(4, 2);   // the planes' partial tiles summed, dW / db / sum of squares stored
    return;
  }

  // ========================= role 1: one split-K slab tile of dd = dP W^T ===========================================
  if (a.diag & 2) return;
  // blocks 8 apart share an XCD: give them ALL tiles of one K slice, so that the slice's columns of dP and W are
  // fetched into one L2 only (without this FETCH_SIZE showed 20.8 MB for this launch against 9.3 MB of operands)
  const int b1 = blockIdx.x - a.n_w;
  const int tiles = a.n_ct * a.n_ht;
  const int xcd1 = b1 & 7, idx1 = b1 >> 3;
  const int z = (idx1 / tiles) * 8 + xcd1, t1 = idx1 % tiles;
  if (z >= a.n_slices + a.n_extra) return;
  const int ct = t1 / a.n_ht, ht = t1 % a.n_ht;
  const int m0 = ct * 32, h0 = ht * 32;
  long kbeg = (long)z * a.k_chunk, kend = min((long)a.ldp, kbeg + a.k_chunk);
  const float* Ab = a.dP; long lda_b = a.ldp; const float* Wb = a.W; int ldw_b = a.ldw;
  const bool extra = z >= a.n_slices;   // block-uniform: a label head's d Y W_lab^T (the array entries picked with constants)
  if (extra) {
    const int e = z - a.n_slices;
    Ab = e == 0 ? a.xA[0] : e == 1 ? a.xA[1] : e == 2 ? a.xA[2] : a.xA[3];
    lda_b = e == 0 ? a.xlda[0] : e == 1 ? a.xlda[1] : e == 2 ? a.xlda[2] : a.xlda[3];
    Wb = e == 0 ? a.xW[0] : e == 1 ? a.xW[1] : e == 2 ? a.xW[2] : a.xW[3];
    ldw_b = e == 0 ? a.xldw[0] : e == 1 ? a.xldw[1] : e == 2 ? a.xldw[2] : a.xldw[3];
    kbeg = 0; kend = e == 0 ? a.xK[0] : e == 1 ? a.xK[1] : e == 2 ? a.xK[2] : a.xK[3];
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if constexpr (B3) {
    // Both operands are k-contiguous rows (dP's cells, W's rows of H).  They are read COALESCED -- 8 lanes per 128-byte line, 4
    // loads of 16 bytes per lane and round of 32 k -- and turned into MFMA operand vectors (row i, 8 consecutive k per lane)
    // through a tile in LDS that is private to the wave; the waves walk their k = kbeg + 32 q, + 256, ... independently, the
    // next round's loads in flight under the current one's products (smx_dgemm.hip has the measurement behind this: a lane
    // reading its own row's line in 16-byte pieces, the f32 form below, sends every piece to L2 again)
    constexpr int LD = 36;
    const int qs = __builtin_amdgcn_readfirstlane(q);
    float* ta = red + qs * (32 * LD);
    float* tb = red + (8 + qs) * (32 * LD);
    const int lr = lane >> 3, lc = lane & 7;
    const float* arow[4];
    long wrow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      arow[j] = Ab + (long)min(m0 + 8 * j + lr, a.B - 1) * lda_b + 4 * lc;   // rows beyond the minibatch compute garbage that nobody reads
      wrow[j] = (long)(h0 + 8 * j + lr) * ((SEP && !extra) ? a.ldw : ldw_b) + 4 * lc;
    }
    float4 a0, a1, a2, a3, b0, b1, b2, b3;
    auto load_round = [&](long kb) {
      const float* wb = Wb;
      long kw = kb;
      if (SEP && !extra) {   // the plane of this round's 32 k (Gp is a multiple of 32: never straddled; wave-uniform)
        const int pl = (int)(kb / a.Gp);
        wb = pl == 0 ? a.Wp[0] : (pl == 1 ? a.Wp[1] : a.Wp[2]);
        kw = kb - (long)pl * a.Gp;
      }
      a0 = *reinterpret_cast<const float4*>(arow[0] + kb); a1 = *reinterpret_cast<const float4*>(arow[1] + kb);
      a2 = *reinterpret_cast<const float4*>(arow[2] + kb); a3 = *reinterpret_cast<const float4*>(arow[3] + kb);
      b0 = *reinterpret_cast<const float4*>(wb + wrow[0] + kw); b1 = *reinterpret_cast<const float4*>(wb + wrow[1] + kw);
      b2 = *reinterpret_cast<const float4*>(wb + wrow[2] + kw); b3 = *reinterpret_cast<const float4*>(wb + wrow[3] + kw);
    };
    float* wa = ta + lr * LD + 4 * lc;
    float* wbt = tb + lr * LD + 4 * lc;
    if (kbeg + 32 * qs < kend) load_round(kbeg + 32 * qs);
    for (long kb = kbeg + 32 * qs; kb < kend; kb += 256) {   // kend - kbeg is a multiple of 32: a wave's 32 k are all in
      *reinterpret_cast<float4*>(wa) = a0; *reinterpret_cast<float4*>(wa + 8 * LD) = a1;
      *reinterpret_cast<float4*>(wa + 16 * LD) = a2; *reinterpret_cast<float4*>(wa + 24 * LD) = a3;
      *reinterpret_cast<float4*>(wbt) = b0; *reinterpret_cast<float4*>(wbt + 8 * LD) = b1;
      *reinterpret_cast<float4*>(wbt + 16 * LD) = b2; *reinterpret_cast<float4*>(wbt + 24 * LD) = b3;
      if (kb + 256 < kend) load_round(kb + 256);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < 2; ++t) {   // step t: lane (i, hh) supplies k = 16 t + 8 hh + s of the wave's 32, for both operands
        const float* pa = ta + i * LD + 16 * t + 8 * hh;
        const float* pb = tb + i * LD + 16 * t + 8 * hh;
        const float4 x0 = *reinterpret_cast<const float4*>(pa), x1 = *reinterpret_cast<const float4*>(pa + 4);
        const float4 y0 = *reinterpret_cast<const float4*>(pb), y1 = *reinterpret_cast<const float4*>(pb + 4);
        const float ax[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        const float bx[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
        acc = mfma_bf16x3(split3x8(ax), split3x8(bx), acc);
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);   // the reads are done before the next round overwrites the tile
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();   // every wave is done with its operand tiles: the partial tiles go over them
  } else {
  const int cell = min(m0 + i, a.B - 1);       // rows beyond the minibatch compute garbage that nobody reads
  const float* ap = Ab + (long)cell * lda_b;
  const float* bp = (SEP && !extra) ? a.Wp[0] : Wb + (long)(h0 + i) * ldw_b;
  // one 512-wide slab per iteration: lane (i, hh) of wave q supplies the 32 consecutive k at 64 q + 32 hh -- one whole
  // 128-byte line of row i of dP and of row h0 + i of W per lane (eight 16-byte loads each, unconditional: a
  // predicated 16-byte load is split into four 4-byte loads by the compiler, 4x the instructions at 32 lines each)
  for (long kb = kbeg; kb < kend; kb += 512) {
    const long k0 = kb + 64 * q + 32 * hh;
    const bool on = k0 < kend;                       // kend is a multiple of 32: a lane's 32 k are all in or all out
    const long kl = on ? k0 : kbeg;                  // (loads of an 'out' lane read valid memory and are zeroed below)
    float4 a4[8], b4[8];
    const float* bq = bp + kl;
    if (SEP && !extra) {   // plane of this lane's 32 k (Gp is a multiple of 32: never straddled)
      const int pl = (int)(kl / a.Gp);
      const float* wb = pl == 0 ? a.Wp[0] : (pl == 1 ? a.Wp[1] : a.Wp[2]);
      bq = wb + (long)(h0 + i) * a.ldw + (kl - (long)pl * a.Gp);
    }
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      a4[v] = *reinterpret_cast<const float4*>(ap + kl + 4 * v);
      b4[v] = *reinterpret_cast<const float4*>(bq + 4 * v);
    }
    __builtin_amdgcn_sched_barrier(0);
    const float m = on ? 1.f : 0.f;
    if constexpr (B3) {
      // step t: the lane's k 8 t .. 8 t + 7 of its 32 (both lane halves' runs form the step's 16 k; A and B agree)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float ax[8] = {a4[2 * t].x * m, a4[2 * t].y * m, a4[2 * t].z * m, a4[2 * t].w * m,
                             a4[2 * t + 1].x * m, a4[2 * t + 1].y * m, a4[2 * t + 1].z * m, a4[2 * t + 1].w * m};
        const float bx[8] = {b4[2 * t].x, b4[2 * t].y, b4[2 * t].z, b4[2 * t].w, b4[2 * t + 1].x, b4[2 * t + 1].y, b4[2 * t + 1].z, b4[2 * t + 1].w};
        acc = mfma_bf16x3(split3x8(ax), split3x8(bx), acc);
      }
    } else {
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[v].x * m, b4[v].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[v].y * m, b4[v].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[v].z * m, b4[v].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[v].w * m, b4[v].w, acc, 0, 0, 0);
      }
    }
  }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(q * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  float* slab = a.slab + (long)z * a.slab_stride;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = 2 * q + j;
    float t = red[(0 * 16 + r) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) t += red[(w * 16 + r) * 64 + lane];
    const int row = m0 + rowof[j];
    if (a.dd_colmajor) { if (row < 128) slab[(long)(h0 + i) * 128 + row] = row < a.B ? t : 0.f; }
    else if (row < a.B) slab[(long)row * a.ldd + h0 + i] = t;
  }
}

bool head_bwd_supported(int B, int Hp, int Gp) { return B > 0 && Hp % 32 == 0 && Gp % 32 == 0; }

// slices of the k Gp axis for dd: whole 512-wide slabs per slice, at most `max_slabs` slices
int head_bwd_slices(long ldp, int max_slabs, int* k_chunk) {
  int chunk = (int)((ldp + max_slabs - 1) / max_slabs);
  chunk = (chunk + 511) / 512 * 512;
  if (k_chunk) *k_chunk = chunk;
  return (int)((ldp + chunk - 1) / chunk);
}

int launch_out_head_bwd(hipStream_t st, const HeadBwdArgs& a_in) {
  HeadBwdArgs a = a_in;
  bool ptrs = a.W && (a.skip_dw || (a.dW && a.db));
  if (a.sep) {
    ptrs = a.n_planes >= 2 && a.n_planes <= 3;
    for (int p = 0; p < a.n_planes && ptrs; ++p) ptrs = a.Wp[p] && a.dWp[p] && a.dbp[p];
  }
  if (!head_bwd_supported(a.B, a.Hp, a.Gp) || !a.D || !a.dP || !ptrs || (!a.slab && !a.skip_dd) || (a.ldp % 4) || (a.ldw % 4) ||
      a.n_slices < 1 || a.k_chunk % 512) {
    set_error("out_head_bwd: bad shapes");
    return SMX_ERR_INVALID;
  }
  if (a.dd_colmajor && (a.B > 128 || a.n_extra != 0 || a.skip_dd || a.slab_stride < (long)a.Hp * 128)) { set_error("out_head_bwd: column-major d d slabs take at most 128 cells and no label slabs"); return SMX_ERR_INVALID; }
  a.n_ht = a.Hp / 32; a.n_gt = a.Gp / 32; a.n_ct = (a.B + 31) / 32;
  a.n_w = a.skip_dw ? 0 : a.n_ht * ((a.n_gt + 7) / 8 * 8);
  { static const int dg = (int)tuning("head_bwd_diag", 0); if (dg) a.diag = dg; }   // (timing only: 1 = role 0 returns at once, 2 = role 1 does)
  if (a.n_extra < 0 || a.n_extra > SMX_MAX_LABELS) { set_error("out_head_bwd: bad label riders"); return SMX_ERR_INVALID; }
  for (int e = 0; e < a.n_extra; ++e)
    if (!a.xA[e] || !a.xW[e] || a.xK[e] <= 0 || (a.xK[e] % 32) || (a.xlda[e] % 4) || (a.xldw[e] % 4)) { set_error("out_head_bwd: bad label riders"); return SMX_ERR_INVALID; }
  const int n_d = a.skip_dd ? 0 : a.n_ct * a.n_ht * ((a.n_slices + a.n_extra + 7) / 8 * 8);
  if (a.sq_count && !a.skip_dw) *a.sq_count = a.n_ht * a.n_gt * 8;
  dim3 grid((unsigned)(a.n_w + n_d));
  if (a.sep) {
    for (int p = 0; p < a.n_planes; ++p)
      if (a.sqp[p] && a.sq_countp[p]) *a.sq_countp[p] = a.n_ht * a.n_gt * 8;
    if (a.n_planes == 3) { if (a.bf16x3) hipLaunchKernelGGL((out_head_bwd_kernel<3, 1, 1>), grid, dim3(512), 0, st, a); else hipLaunchKernelGGL((out_head_bwd_kernel<3, 1>), grid, dim3(512), 0, st, a); }
    else { if (a.bf16x3) hipLaunchKernelGGL((out_head_bwd_kernel<2, 1, 1>), grid, dim3(512), 0, st, a); else hipLaunchKernelGGL((out_head_bwd_kernel<2, 1>), grid, dim3(512), 0, st, a); }
    SMX_HIP(hipGetLastError());
    return SMX_OK;
  }
  if (a.n_planes == 3) { if (a.bf16x3) hipLaunchKernelGGL((out_head_bwd_kernel<3, 0, 1>), grid, dim3(512), 0, st, a); else hipLaunchKernelGGL((out_head_bwd_kernel<3>), grid, dim3(512), 0, st, a); }
  else if (a.n_planes == 2) { if (a.bf16x3) hipLaunchKernelGGL((out_head_bwd_kernel<2, 0, 1>), grid, dim3(512), 0, st, a); else hipLaunchKernelGGL((out_head_bwd_kernel<2>), grid, dim3(512), 0, st, a); }
  else { set_error("out_head_bwd: 2 or 3 planes"); return SMX_ERR_INVALID; }
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

}  // namespace smx

// =====================================================================================================================
// Weight gradients whose contraction axis is the minibatch (C = A^T Bm, K = B cells), grouped in one launch: the first
// encoder layer (A = log1p of the gathered count rows), the latent head (+ bias gradient), the first decoder layer.
// Same scheme as role 0 above: 32 x 32 tile per workgroup, 8 waves split the cells, both operands are coalesced
// 128-byte row segments loaded straight into MFMA operand registers.
// =====================================================================================================================
namespace smx {

// one 32 x 32 tile of problem P: `red` 8 partial tiles (32 KB), `sqs` 8 floats
__device__ __forceinline__ void wgrad_tile_body(const WgradGroup& Gr, const WgradProblem& P, float* red, float* sqs) {
  preload(P.A, P.lda, P.a_mode, P.log1p, P.rows, P.Bm, P.ldb, P.C, P.ldc, P.M, P.n_mt, P.n_nt, P.start, P.colsum, P.sq_part, Gr.B);   // (one batch: smx_device.h)
  const int local = blockIdx.x - P.start;
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = lane & 31, hh = lane >> 5;
  // blocks 8 apart share an XCD: they take the column tiles of ONE row tile (the gathered A rows are fetched once)
  const int xcd = local & 7, idx = local >> 3;
  const int nt = idx % P.n_nt, mt = (idx / P.n_nt) * 8 + xcd;
  if (mt >= P.n_mt) return;
  const int m0 = mt * 32, n0 = nt * 32;
  const int B = Gr.B;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float csum = 0.f;
  for (int kc = 0; kc < B; kc += 128) {
    if (kc + 16 * q >= B) break;   // wave-uniform
    const int k0 = kc + 16 * q + 8 * hh;
    long arow[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int cell = min(k0 + s, B - 1);
      arow[s] = (P.a_mode && P.rows) ? (long)P.rows[cell] : (long)cell;
    }
    float av[8], bv[8];
    // (the store's format outside the request loop: `a_mode == 2 ? u16 : f32` per element put a wait and the conversion behind every uint16 request --
    // eight round trips in a row for the compact store; the f32 arm is the loop as it was)
    if (P.a_mode == 2) {
      uint16_t raw[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        bv[s] = P.Bm[(long)min(k0 + s, B - 1) * P.ldb + n0 + i];
        raw[s] = reinterpret_cast<const uint16_t*>(P.A)[arow[s] * P.lda + m0 + i];
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) av[s] = (float)raw[s];
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        bv[s] = P.Bm[(long)min(k0 + s, B - 1) * P.ldb + n0 + i];
        av[s] = P.A[arow[s] * P.lda + m0 + i];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (Gr.b3) {   // (launch-uniform) the product from bf16 MFMAs on three-way split operands: the lane's 8 cells are one operand
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const bool on = k0 + s < B;
        float a_s = av[s];
        if (P.a_mode && P.log1p) a_s = log1p_count(a_s);
        av[s] = on ? a_s : 0.f;
        bv[s] = on ? bv[s] : 0.f;
        csum += bv[s];
      }
      acc = mfma_bf16x3(split3x8(av), split3x8(bv), acc);
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const bool on = k0 + s < B;
        float a_s = av[s];
        if (P.a_mode && P.log1p) a_s = log1p_count(a_s);
        a_s = on ? a_s : 0.f;
        const float b_s = on ? bv[s] : 0.f;
        csum += b_s;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_s, b_s, acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(q * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = 2 * q + j;
    float t = red[(0 * 16 + r) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) t += red[(w * 16 + r) * 64 + lane];
    const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
    if (row < P.M) { P.C[(long)row * P.ldc + n0 + i] = t; sq += t * t; }
  }
  if (P.sq_part) {   // 4 slots per 32 x 32 tile (the layout of the per-tensor slot table): waves pair up
    sq = wave_sum(sq);
    if (lane == 0) sqs[q] = sq;
  }
  if (P.colsum && mt == 0) {
    __syncthreads();
    red[q * 64 + lane] = csum;
  }
  __syncthreads();
  if (P.sq_part && threadIdx.x < 4) P.sq_part[((long)mt * P.n_nt + nt) * 4 + threadIdx.x] = sqs[2 * threadIdx.x] + sqs[2 * threadIdx.x + 1];
  if (P.colsum && mt == 0 && q == 0 && lane < 32) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += red[w * 64 + lane] + red[w * 64 + 32 + lane];
    P.colsum[n0 + lane] = t;
  }
}


// the problem of this workgroup: every start in one scalar load (a `while (blockIdx.x >= p[pi + 1].start)` walk is a scalar load, a wait and a
// branch per problem: up to seven dependent scalar-cache round trips at the head of every workgroup)
__device__ __forceinline__ int wgrad_problem_of(const WgradGroup& Gr) {
  int pi = 0;
#pragma unroll
  for (int k = 1; k < SMX_GROUP_MAX; ++k) pi += ((int)blockIdx.x >= Gr.starts[k]) ? 1 : 0;
  return pi;
}

__global__ __launch_bounds__(512) void wgrad_group_kernel(WgradGroup G) {
  __shared__ float red[8 * 1024];
  __shared__ float sqs[8];
  // the descriptor through the kernarg segment pointer: indexing the by-value struct with a run-time problem id would
  // make the compiler copy all of it to scratch (see gemm_group_kernel)
  const WgradGroup& Gr = *(const WgradGroup*)__builtin_amdgcn_kernarg_segment_ptr();
  const int pi = wgrad_problem_of(Gr);
  wgrad_tile_body(Gr, Gr.p[pi], red, sqs);
}

// The same group with wide problems in it (a gene panel as M, N <= 128): those take the panel form (smx_panel.h, role 0), one
// workgroup per 32 rows with every column tile; the group's small problems keep the 32 x 32 tiles.  A kernel of its own: the
// panel form's 128 registers would otherwise cap the occupancy of every launch of the tile kernel (48 registers).
template <int ONE>
__global__ __launch_bounds__(512, ONE ? 4 : 2) void wgrad_panel_group_kernel(WgradGroup G) {   // (one chunk: 128 registers, two workgroups per CU)
  __shared__ __attribute__((aligned(16))) float red[SMX_PANEL_SMEM_FLOATS];
  __shared__ float sqs[8];
  const WgradGroup& Gr = *(const WgradGroup*)__builtin_amdgcn_kernarg_segment_ptr();
  const int pi = wgrad_problem_of(Gr);
  const WgradProblem& P = Gr.p[pi];
  if (!P.panel) { wgrad_tile_body(Gr, P, red, sqs); return; }   // (block-uniform)
  PanelProblem pp;
  pp.big = P.A; pp.ld_big = P.lda; pp.big_mode = P.a_mode; pp.log1p = P.a_mode ? P.log1p : 0; pp.rows = P.a_mode ? P.rows : nullptr;
  pp.S = P.Bm; pp.ldS = P.ldb; pp.n_st = P.n_nt; pp.out = P.C; pp.ld_out = P.ldc;
  pp.s_colsum = P.colsum; pp.sq_part = P.sq_part; pp.n_wt = P.n_mt; pp.B = Gr.B;
  if (P.a_mode == 2) panel_body<0, 2, ONE>(pp, (int)blockIdx.x - P.start, P.panel, red);
  else if (P.a_mode == 1) panel_body<0, 1, ONE>(pp, (int)blockIdx.x - P.start, P.panel, red);
  else panel_body<0, 0, ONE>(pp, (int)blockIdx.x - P.start, P.panel, red);
}

// The group beside ONE product of the direct-operand form (smx_dgemm.h): a layer's weight gradient and its input gradient both
// read the layer's d pre-activation and nothing of each other -- FactorVAE's 1000-wide discriminator layers, where each of the
// two is a latency chain of ~9 us that leaves the chip half idle.  Workgroups [0, n_w) are the group's, the rest the product's.
template <int ONE>
__global__ __launch_bounds__(512, ONE ? 4 : 2) void wgrad_dgemm_kernel(WgradGroup G, GemmArgs g, int n_w) {
  constexpr int SM = SMX_DG_SMEM_FLOATS(1) > SMX_PANEL_SMEM_FLOATS ? SMX_DG_SMEM_FLOATS(1) : SMX_PANEL_SMEM_FLOATS;
  __shared__ __attribute__((aligned(16))) float red[SM];
  __shared__ float sqs[8];
  if ((int)blockIdx.x >= n_w) { dgemm_body<1>(g, (int)blockIdx.x - n_w, red); return; }   // (block-uniform)
  const WgradGroup& Gr = *(const WgradGroup*)__builtin_amdgcn_kernarg_segment_ptr();
  const int pi = wgrad_problem_of(Gr);
  const WgradProblem& P = Gr.p[pi];
  if (!P.panel) { wgrad_tile_body(Gr, P, red, sqs); return; }
  PanelProblem pp;
  pp.big = P.A; pp.ld_big = P.lda; pp.big_mode = P.a_mode; pp.log1p = P.a_mode ? P.log1p : 0; pp.rows = P.a_mode ? P.rows : nullptr;
  pp.S = P.Bm; pp.ldS = P.ldb; pp.n_st = P.n_nt; pp.out = P.C; pp.ld_out = P.ldc;
  pp.s_colsum = P.colsum; pp.sq_part = P.sq_part; pp.n_wt = P.n_mt; pp.B = Gr.B;
  if (P.a_mode == 2) panel_body<0, 2, ONE>(pp, (int)blockIdx.x - P.start, P.panel, red);
  else if (P.a_mode == 1) panel_body<0, 1, ONE>(pp, (int)blockIdx.x - P.start, P.panel, red);
  else panel_body<0, 0, ONE>(pp, (int)blockIdx.x - P.start, P.panel, red);
}

bool wgrad_supported(const GemmArgs& g, int B) {
  // C = A^T Bm with A stored [K][M] (k-major), Bm [K][N], K = the minibatch; optional gather + log1p of A; no input dropout
  if (!g.a_kmajor || g.b_nmajor || g.K != B || g.split_k > 1 || g.epi != 0 || g.bias) return false;
  if (g.N % 32 || g.M % 32 || g.M <= 0) return false;   // (every weight has its rows padded to 32)
  if (g.use_xform && (g.xf.drop_p > 0.f || g.xf.inj_mask)) return false;
  return true;
}

int launch_wgrad_group(hipStream_t st, const GemmArgs* list, int n, int B, int bf16x3, const GemmArgs* beside) {
  if (n < 1 || n > SMX_GROUP_MAX) { set_error("wgrad group: 1..SMX_GROUP_MAX problems"); return SMX_ERR_INVALID; }
  WgradGroup G;
  memset(&G, 0, sizeof(G));
  G.n = n; G.B = B; G.b3 = bf16x3;
  for (int k = 0; k < SMX_GROUP_MAX; ++k) G.starts[k] = 0x7FFFFFFF;
  int total = 0;
  bool any_panel = false;
  for (int k = 0; k < n; ++k) {
    const GemmArgs& g = list[k];
    if (!wgrad_supported(g, B)) { set_error("wgrad group: unsupported problem"); return SMX_ERR_INVALID; }
    WgradProblem& P = G.p[k];
    G.starts[k] = total;
    P.A = g.A; P.lda = g.lda; P.a_mode = g.use_xform ? (g.xf.u16 ? 2 : 1) : 0; P.log1p = g.use_xform ? g.xf.log1p : 0;
    P.rows = g.use_xform ? g.xf.rows : nullptr;
    P.Bm = g.B; P.ldb = g.ldb; P.C = g.C; P.ldc = g.ldc; P.M = g.M; P.N = g.N;
    P.colsum = g.colsum; P.sq_part = g.sq_part;
    P.n_mt = (g.M + 31) / 32; P.n_nt = g.N / 32;
    P.start = total;
    P.panel = (bf16x3 && (g.M >= SMX_PANEL_MIN_WIDE || g.panel_hint) && g.N <= 128 && !tuning_on("no_panel")) ? panel_grid(P.n_mt) : 0;
    if (P.panel) {
      any_panel = true;
      total += P.panel;
      if (g.sq_part && g.sq_count) *g.sq_count = P.n_mt * 8;
      continue;
    }
    total += P.n_nt * ((P.n_mt + 7) / 8 * 8);
    if (g.sq_part && g.sq_count) *g.sq_count = P.n_mt * P.n_nt * 4;
  }
  if (beside && !(dgemm_supported(*beside) && beside->b_nmajor)) { set_error("wgrad group: the product beside it is not of the direct-operand form"); return SMX_ERR_INVALID; }
  if (beside && any_panel) {   // (without a panel problem the two stay two launches: the tile kernel's 48 registers are worth more)
    const GemmArgs& d = *beside;
    const int d_grid = ((d.M + 31) / 32) * ((d.N / 32 + 7) / 8 * 8);
    if (B <= 128) hipLaunchKernelGGL(wgrad_dgemm_kernel<1>, dim3((unsigned)(total + d_grid)), dim3(512), 0, st, G, d, total);
    else hipLaunchKernelGGL(wgrad_dgemm_kernel<0>, dim3((unsigned)(total + d_grid)), dim3(512), 0, st, G, d, total);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
  }
  if (any_panel && B <= 128) hipLaunchKernelGGL(wgrad_panel_group_kernel<1>, dim3((unsigned)total), dim3(512), 0, st, G);
  else if (any_panel) hipLaunchKernelGGL(wgrad_panel_group_kernel<0>, dim3((unsigned)total), dim3(512), 0, st, G);
  else hipLaunchKernelGGL(wgrad_group_kernel, dim3((unsigned)total), dim3(512), 0, st, G);
  SMX_HIP(hipGetLastError());
  if (beside) return launch_dgemm(st, *beside);
  return SMX_OK;
}


// ---- the output head's dW / db at a wide panel (smx_panel.h, role 1): one workgroup per (gene tile, plane) ----------------
template <int ONE>
__global__ __launch_bounds__(512, ONE ? 4 : 2) void panel_dw_kernel(PanelProblem P) {   // (one chunk: 4 waves per SIMD = 128 registers, two workgroups per CU)
  __shared__ __attribute__((aligned(16))) float smem[SMX_PANEL_SMEM_FLOATS];
  panel_body<1, 0, ONE>(P, (int)blockIdx.x, (int)gridDim.x, smem);
}

int panel_grid(int units) {
  // the workgroups are resident two per CU (128 registers, 45 KB of LDS) and each walks units first, first + grid, ...: as
  // many rounds as 512 workgroups need, then the grid that fills those rounds evenly (1875 units: 4 rounds of 469)
  static const int cap = std::max((int)tuning("panel_grid", 512), 1);
  const int rounds = (units + cap - 1) / cap;
  return (units + rounds - 1) / std::max(rounds, 1);
}

bool panel_dw_supported(const HeadBwdArgs& a) {
  return !a.sep && a.B > 0 && a.Hp % 32 == 0 && a.Hp <= 128 && a.Gp % 32 == 0 && a.Gp >= SMX_PANEL_MIN_WIDE &&
         a.n_planes >= 1 && a.n_planes <= 3 && !tuning_on("no_panel");
}

int launch_panel_dw(hipStream_t st, const HeadBwdArgs& a) {
  if (!panel_dw_supported(a) || !a.D || !a.dP || !a.dW || !a.db) { set_error("panel_dw: bad shapes"); return SMX_ERR_INVALID; }
  PanelProblem P;
  P.big = a.dP; P.ld_big = a.ldp; P.sub_stride = a.Gp; P.n_sub = a.n_planes;
  P.S = a.D; P.ldS = a.ldd; P.n_st = a.Hp / 32;
  P.out = a.dW; P.ld_out = a.ldw; P.big_colsum = a.db; P.sq_part = a.sq_part;
  P.n_wt = a.Gp / 32; P.B = a.B;
  const int total = P.n_wt * P.n_sub;
  if (a.sq_part && a.sq_count) *a.sq_count = total * 8;
  if (a.B <= 128) hipLaunchKernelGGL(panel_dw_kernel<1>, dim3((unsigned)panel_grid(total)), dim3(512), 0, st, P);
  else hipLaunchKernelGGL(panel_dw_kernel<0>, dim3((unsigned)panel_grid(total)), dim3(512), 0, st, P);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

}  // namespace smx

#ifdef SMX_STAMPS
extern "C" int smx_dbg_stamps_headbwd(long long* out) {   // development builds only (tools/c2_stamps.sh): this unit's stamp table [16][16]
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(smx::smx_tu_stamps), sizeof(long long) * 256) == hipSuccess ? 0 : -1;
}
#endif
