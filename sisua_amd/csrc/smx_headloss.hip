// smx_headloss.hip -- output-head product FUSED with the count likelihood, wide (training step of VAE / DCA /
// SISUA count outputs; SURVEY.md 8 rows a-9 + a-10 / a-11):
//
//   P = d W_out + b  ->  NB / ZINB / NBD / ZINBD log-likelihood of x and d llk / d P, scaled  ->  dP, llk partials
//
// The parameter planes P never exist in memory: they live in MFMA accumulators, the likelihood runs on them and
// only dP (which the two backward products read) and one partial sum per (cell, gene tile) are written.  Against
// the product + loss kernel pair that removes the P write (4kG B/cell), the P read and one kernel boundary.
//
// Shape of the work: a workgroup owns a 32-cell x 32-gene tile with ALL k planes of it and
// splits K = H over its 4 waves, so that the grid is (B / 32) x (Gp / 32) = 252 workgroups at the benchmark size
// (one per CU) -- the earlier fused head (smx_head.hip: 16 genes x whole batch, with dW / db folded in) had 126.
//  * no LDS in the main loop: with K split over the waves no operand element is used by two waves, so both go
//    straight from global memory to the MFMA operand registers.  The k index of v_mfma_f32_32x32x2_f32 is free
//    to permute: lane (i, h) supplies k = 16 h + s in step s, i.e. 16 CONSECUTIVE floats of row i of d (four
//    16-byte loads) and, for W, rows 16 h + s of a 128-byte column segment (coalesced);
//  * the 4 partial tiles meet in LDS (k x 16 KB), wave q finishes accumulator registers 4q .. 4q+3 of every plane
//    = cells 8q .. 8q+7 of the tile: sum in wave order, bias, likelihood, gradient stores;
//  * the counts x are gathered BEFORE the product (they do not depend on it);
//  * XCD-aware mapping: the (B / 32) workgroups that share a W tile are 8 blocks apart (same XCD under round-robin
//    dispatch), so each W tile is fetched into one L2 only.
#include <stdlib.h>

#include "smx_internal.h"
#include "smx_loss.h"
#include "../../include/sisua_hip.h"

namespace smx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// EPI: 1 likelihood epilogue (the training kernel); 0 product only -- stores P instead of dP, reads no counts
// (what the standalone product kernel does; used to attribute the fused kernel's time to the likelihood);
// 2 likelihood values only (scoring over stacked posterior draws: no gradient, no stores but the partial sums).
// NW waves per workgroup split every 128-deep slab of K: lane (i, h) of wave q supplies k = q KS + (KS/2) h + s in
// MFMA step s (KS = 128 / NW) and finishes accumulator registers q RPW .. q RPW + RPW - 1 (RPW = 16 / NW).
// SLAB: depth of K a workgroup takes per round of loads (128 in training; the scoring form with 1 / 2 waves per
// workgroup uses 32 / 64 so that a lane still holds 16 k-steps of operands).
// B3: the product from bf16 MFMAs on three-way split operands (smx_device.h: split3x8 / mfma_bf16x3), split in registers
// right after the loads -- a lane's 8 consecutive k of a slab ARE its operand of one v_mfma_f32_32x32x16_bf16 (8 waves, K
// slab 128: KH = 8).  Same operand ownership, same exchange, same epilogue as the f32 form; the wide output heads take it
// (launch_hl), where the f32 MFMAs' vector-pipe time dominates the launch.
template <int LK, int U16, int EPI, int NW, int SLAB = 128, int B3 = 0>
__global__ __launch_bounds__(64 * NW) void out_head_loss_kernel(HeadLossArgs a) {
  static_assert(!B3 || SLAB / NW == 16, "the bf16 form needs 8 consecutive k per lane half");
  constexpr int NP = (LK == SMX_LLK_ZINB || LK == SMX_LLK_ZINBD) ? 3 : 2;
  constexpr int KS = SLAB / NW, KH = KS / 2, RPW = 16 / NW;
  // ONE plane's NW partial tiles at a time (32 KB at 8 waves) | training epilogue: the waves' queues of their non-zero counts
  // (smx_loss.h: one float2 per element of the tile = 8 KB) -- ONE array (a second __shared__ object can cost waits)
  constexpr bool QUEUE = EPI == 1 && NW >= 4;
  constexpr int QOFF = NW > 1 ? NW * 1024 : 0;
  __shared__ float red[(NW > 1 ? NW * 1024 : 1) + (QUEUE ? 2 * 1024 : 0)];
  float2* const lq = reinterpret_cast<float2*>(red + QOFF);
  preload(a.H, a.ldh, a.W, a.ldw, a.bias, a.X, a.ldx, a.rows, a.dP, a.ldp, a.plane_stride, a.llk_part, a.B, a.G, a.Gp, a.Hp,
          a.grad_scale, a.n_ct, a.n_gt, a.x_u16, a.row_mod, a.product_only, a.llk_only, a.likelihood);   // (the argument fields in one batch: smx_device.h)
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  // blocks b, b + 8, b + 16, ... share an XCD: give them the cell tiles of ONE gene tile
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int ct = idx % a.n_ct, gt = (idx / a.n_ct) * 8 + xcd;
  if (gt >= a.n_gt) return;
  const int m0 = ct * 32, n0 = gt * 32;
  const int col = n0 + i;
  // accumulator register r of a 32 x 32 tile is row (r & 3) + 8 (r >> 2) + 4 h, column i
  int rowof[RPW];
#pragma unroll
  for (int j = 0; j < RPW; ++j) { const int r = q * RPW + j; rowof[j] = m0 + (r & 3) + 8 * (r >> 2) + 4 * h; }

  // ---- loads, oldest first: row ids of this wave's finishing cells -> operands of the first K slab -> the counts
  // (they do not depend on the product; their latency hides under the MFMAs) -----------------------------------------
  long src[RPW];
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    int cell = min(rowof[j], a.B - 1);
    if (EPI == 2) cell %= a.row_mod;
    src[j] = (EPI && a.rows) ? a.rows[cell] : cell;
  }
  float bias[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) bias[p] = a.bias[(long)p * a.Gp + col];
  __builtin_amdgcn_sched_barrier(0);

  f32x16 acc[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
  const int arow = min(m0 + i, a.B - 1);   // rows beyond the batch compute garbage that is never stored
  float av[KH], bv[NP][KH];
  // ALL operand loads of a slab are issued before its first MFMA (left alone the compiler waits for each W load
  // right before the MFMA that uses it: ~30 serial L2 / HBM round trips, 9.8 us for the product alone)
  auto load_slab = [&](int kc, bool mine) {
    const int k0 = mine ? kc + KS * q + KH * h : KH * h;   // first k of this lane half in this slab (a wave without a slice: any valid rows, unused)
    if (EPI == 2) {   // k-major H (smx_score.hip): lanes i = 32 consecutive rows of one k, as for W
      const float* ap = a.H + (long)k0 * a.ldh + arow;
#pragma unroll
      for (int s = 0; s < KH; ++s) av[s] = ap[(long)s * a.ldh];
    } else {
      const float* ap = a.H + (long)arow * a.ldh + k0;
#pragma unroll
      for (int v = 0; v < KH / 4; ++v) {
        const float4 t = *reinterpret_cast<const float4*>(ap + 4 * v);
        av[4 * v] = t.x; av[4 * v + 1] = t.y; av[4 * v + 2] = t.z; av[4 * v + 3] = t.w;
      }
    }
    const float* wp = a.W + (long)k0 * a.ldw + col;
#pragma unroll
    for (int s = 0; s < KH; ++s)
#pragma unroll
      for (int p = 0; p < NP; ++p) bv[p][s] = wp[(long)s * a.ldw + (long)p * a.Gp];
  };
  auto mfma_slab = [&]() {
    if constexpr (B3) {
      const Split8 sa = split3x8(*reinterpret_cast<const float (*)[8]>(&av[0]));
#pragma unroll
      for (int p = 0; p < NP; ++p) acc[p] = mfma_bf16x3(sa, split3x8(*reinterpret_cast<const float (*)[8]>(&bv[p][0])), acc[p]);
    } else {
#pragma unroll
      for (int s = 0; s < KH; ++s)
#pragma unroll
        for (int p = 0; p < NP; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[p][s], acc[p], 0, 0, 0);
    }
  };
  const bool active = KS * q < a.Hp;              // wave-uniform: this wave has a K slice in the first slab
  // (no branch around the requests: behind its join the compiler's wait for the row ids -- which the counts' addresses need -- had to fit the path
  // WITHOUT the slab's 26 requests and became a wait for all of them: the counts left one round trip late)
  load_slab(0, active);
  __builtin_amdgcn_sched_barrier(0);
  float xs[RPW];
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    xs[j] = 0.f;
    if (EPI) xs[j] = U16 ? (float)reinterpret_cast<const uint16_t*>(a.X)[src[j] * a.ldx + col] : a.X[src[j] * a.ldx + col];
  }
  __builtin_amdgcn_sched_barrier(0);
  if (active) mfma_slab();
  for (int kc = SLAB; kc + KS * q < a.Hp; kc += SLAB) {
    load_slab(kc, true);
    __builtin_amdgcn_sched_barrier(0);
    mfma_slab();
  }

  // ---- the NW partial tiles meet in LDS, plane by plane; wave q finishes registers q RPW .. q RPW + RPW - 1 ---------
  float v[NP][RPW];
  if (NW == 1) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int j = 0; j < RPW; ++j) v[p][j] = acc[p][j] + bias[p];
  }
#pragma unroll
  for (int p = 0; p < NP && NW > 1; ++p) {
    if (p) __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(q * 16 + r) * 64 + lane] = acc[p][r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int r = q * RPW + j;
      float t = red[(0 * 16 + r) * 64 + lane];
#pragma unroll
      for (int w = 1; w < NW; ++w) t += red[(w * 16 + r) * 64 + lane];
      v[p][j] = t + bias[p];
    }
  }

  // ---- likelihood + gradient on the tile: the RPW elements of a lane as interleaved straight-line chains ------------
  const bool live = col < a.G;
  float llk[RPW], d0[RPW], d1[RPW], d2[RPW];
  if (EPI) {
    float p2[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) p2[j] = NP == 3 ? v[NP - 1][j] : 0.f;
    constexpr int CH = RPW > 4 ? 4 : RPW;   // interleaved chains at a time (all 16 of a one-wave workgroup would spill)
#pragma unroll
    for (int c = 0; c < RPW; c += CH) {
      typedef float Vec[CH];
      static_assert(!QUEUE || CH == RPW, "the queue form: one call per tile (a wave's stretch holds its RPW elements per lane)");
      count_elem_vec<LK, 0, CH>(*(const Vec*)(xs + c), *(const Vec*)(v[0] + c), *(const Vec*)(v[1] + c), *(const Vec*)(p2 + c),
                                *(Vec*)(llk + c), *(Vec*)(d0 + c), *(Vec*)(d1 + c), *(Vec*)(d2 + c), QUEUE ? lq : nullptr);
    }
  }
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int cell = rowof[j];
    const bool ok = live && cell < a.B;
    float d[3];
    if (EPI) { d[0] = ok ? d0[j] * a.grad_scale : 0.f; d[1] = ok ? d1[j] * a.grad_scale : 0.f; d[2] = ok ? d2[j] * a.grad_scale : 0.f; }
    else { d[0] = v[0][j]; d[1] = v[1][j]; d[2] = NP == 3 ? v[NP - 1][j] : 0.f; }
    if (EPI != 2 && cell < a.B) {
      float* dp = a.dP + (long)cell * a.ldp + col;
#pragma unroll
      for (int p = 0; p < NP; ++p) dp[(long)p * a.plane_stride] = d[p];
    }
    if (EPI) {
      // per-cell partial over the tile's 32 genes: the 32 lanes of this half hold them
      const float t = half_wave_sum(ok ? llk[j] : 0.f);
      if (i == 0 && cell < a.B) a.llk_part[(long)cell * a.n_gt + gt] = t;
    }
  }
}

bool head_loss_supported(int B, int Hp, int Gp) { return B > 0 && Hp % 32 == 0 && Gp % 32 == 0; }
int head_loss_chunks(int Gp) { return Gp / 32; }

template <int LK, int NW, int B3 = 0>
static void launch_hl_w(hipStream_t st, const HeadLossArgs& a, dim3 grid) {
  if (a.product_only) {
    hipLaunchKernelGGL((out_head_loss_kernel<LK, 0, 0, NW, 128, B3>), grid, dim3(64 * NW), 0, st, a);
  } else if (a.x_u16) {
    hipLaunchKernelGGL((out_head_loss_kernel<LK, 1, 1, NW, 128, B3>), grid, dim3(64 * NW), 0, st, a);
  } else {
    hipLaunchKernelGGL((out_head_loss_kernel<LK, 0, 1, NW, 128, B3>), grid, dim3(64 * NW), 0, st, a);
  }
}
// bf16 x 3 products: SMX_BF16X3 = 1 always, 0 never (the exact-f32 MFMA forms); default: SMX_BF16X3_MIN_WORK (smx_internal.h)
bool use_bf16x3(long work) {
  static const int forced = (int)tuning("bf16x3", -1);
  return forced >= 0 ? forced != 0 : work >= SMX_BF16X3_MIN_WORK;
}
template <int LK>
static void launch_hl(hipStream_t st, const HeadLossArgs& a, dim3 grid) {
  if (a.llk_only) {
    // 2 waves per workgroup (K slab 64): measured at 128 cells x 1000 draws, 8 000 genes with 1 / 2 / 4 / 8 / 16 waves:
    // 3.45 / 3.28 / 3.42 / 3.84 / 5.77 ms -- with hundreds of cell tiles the chip is full without a deep split, and every
    // wave less means fewer partial tiles through LDS (profiles/r02_scoring_path.txt)
    if (a.x_u16) hipLaunchKernelGGL((out_head_loss_kernel<LK, 1, 2, 2, 64>), grid, dim3(128), 0, st, a);
    else hipLaunchKernelGGL((out_head_loss_kernel<LK, 0, 2, 2, 64>), grid, dim3(128), 0, st, a);
    return;
  }
  if (a.bf16x3) launch_hl_w<LK, 8, 1>(st, a, grid);   // (8 waves: 4 / 16 measured slower at every width, 7.3 / 8.5 / 8.0 us at C2; forms removed in round 4)
  else launch_hl_w<LK, 8>(st, a, grid);
}

int launch_out_head_loss(hipStream_t st, const HeadLossArgs& a_in) {
  HeadLossArgs a = a_in;
  if (!head_loss_supported(a.B, a.Hp, a.Gp) || (!a.llk_only && (a.ldh % 4)) || (a.llk_only && a.ldh < a.B) || !a.H || !a.W || !a.bias || (!a.dP && !a.llk_only) || (!a.product_only && (!a.X || !a.llk_part)) ||
      (a.llk_only && (a.product_only || a.row_mod <= 0))) {
    set_error("out_head_loss: bad shapes");
    return SMX_ERR_INVALID;
  }
  a.n_ct = (a.B + 31) / 32;
  a.n_gt = a.Gp / 32;
  const int gt8 = (a.n_gt + 7) / 8 * 8;
  dim3 grid((unsigned)(a.n_ct * gt8));
  switch (a.likelihood) {
    case SMX_LLK_NB: launch_hl<SMX_LLK_NB>(st, a, grid); break;
    case SMX_LLK_ZINB: launch_hl<SMX_LLK_ZINB>(st, a, grid); break;
    case SMX_LLK_NBD: launch_hl<SMX_LLK_NBD>(st, a, grid); break;
    case SMX_LLK_ZINBD: launch_hl<SMX_LLK_ZINBD>(st, a, grid); break;
    default: set_error("out_head_loss: unknown likelihood"); return SMX_ERR_INVALID;
  }
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

}  // namespace smx
