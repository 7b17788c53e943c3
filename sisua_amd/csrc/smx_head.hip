// smx_head.hip -- the output head of a training step as ONE kernel (VAE / DCA / SISUA count output):
//
//   P = d W_out + b      ->   NB / ZINB log-likelihood and dP (a-10 / a-11)   ->
//   dW_out = d^T dP,  db = colsum(dP)          (a-16, the largest weight gradient)
//
// A workgroup owns 16 genes (all k parameter planes of them) and the whole minibatch, so the parameter planes
// never leave the chip: P lives in MFMA accumulators (v_mfma_f32_16x16x4_f32: the lane that holds P[b][g] of one
// plane holds it for every plane), the likelihood runs on those registers, dP goes to LDS as the B operand
// of the weight-gradient product and to HBM once for the input-gradient product (dd = dP W^T, a separate
// split-K launch).  HBM traffic per step: W (read once), x gather, dP (written once), dW (written once) --
// the P write + read and one dP read of the three-kernel form are gone, with two kernel boundaries.
//
// LDS (124.9 KB of the 160 KB): d [128][148] (row-major; the stride makes both the [b][k] reads of the forward
// product and the [k][h] reads of the weight gradient conflict-free per 32-lane half), W tile [128][48],
// dP tile [128][48].
#include "smx_internal.h"
#include "smx_loss.h"
#include "../../include/sisua_hip.h"

namespace smx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HD_ROWS = 128;      // minibatch rows per workgroup (whole batch)
constexpr int HD_K = 128;         // decoder width limit
constexpr int HD_LDH = 148;
constexpr int HD_GENES = 16;
constexpr int HD_LDW = 48;        // 3 planes x 16 genes
constexpr int HD_THREADS = 512;
constexpr int HD_LDS_FLOATS = HD_ROWS * HD_LDH + HD_K * HD_LDW + HD_ROWS * HD_LDW + 8 * HD_LDW;

template <int LK>
__global__ __launch_bounds__(HD_THREADS) void out_head_train_kernel(OutHeadArgs a) {
  constexpr int K = (LK == SMX_LLK_ZINB || LK == SMX_LLK_ZINBD) ? 3 : 2;
  extern __shared__ __attribute__((aligned(16))) float hd_lds[];
  float* Hs = hd_lds;                                  // [128][148]
  float* Ws = Hs + HD_ROWS * HD_LDH;                   // [Hp][48]
  float* Ds = Ws + HD_K * HD_LDW;                      // [128][48]
  float* Cs = Ds + HD_ROWS * HD_LDW;                   // [8 waves][48] column sums of dP
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lc = lane & 15, lq = lane >> 4;            // MFMA: column / k (operands), column / row quad (result)
  const int g0 = blockIdx.x * HD_GENES;
  const int Hp = a.Hp;

  // ---- stage d (zero rows beyond the batch) and the W tile; start the x gather ----------------
  for (int f = tid; f < HD_ROWS * (HD_K / 4); f += HD_THREADS) {
    const int r = f / (HD_K / 4), kq = f % (HD_K / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < a.B && kq * 4 < Hp) v = *reinterpret_cast<const float4*>(a.H + (long)r * a.ldh + kq * 4);
    *reinterpret_cast<float4*>(&Hs[r * HD_LDH + kq * 4]) = v;
  }
  for (int f = tid; f < HD_K * K * 4; f += HD_THREADS) {
    const int k = f / (K * 4), rest = f % (K * 4), c = rest / 4, q = rest % 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < Hp) v = *reinterpret_cast<const float4*>(a.W + (long)k * a.ldw + (long)c * a.Gp + g0 + q * 4);
    *reinterpret_cast<float4*>(&Ws[k * HD_LDW + c * 16 + q * 4]) = v;
  }
  const int gene = g0 + lc;
  float xs[4], bias[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = 16 * w + 4 * lq + i;
    xs[i] = 0.f;
    if (b < a.B) {
      const long src = a.rows ? a.rows[b] : b;
      xs[i] = a.x_u16 ? (float)reinterpret_cast<const uint16_t*>(a.X)[src * a.ldx + gene] : a.X[src * a.ldx + gene];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) bias[c] = (c < K) ? a.bias[(long)c * a.Gp + gene] : 0.f;
  __syncthreads();

  // ---- forward: rows 16w..16w+15 of P for the K planes -----------------------------------------
  f32x4 acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const float* ap = Hs + (16 * w + lc) * HD_LDH + lq;
    const float* bp = Ws + lq * HD_LDW + lc;
    const int fsteps = (a.diag & 1) ? 1 : Hp / 4;
    for (int s = 0; s < fsteps; ++s) {
      const float av = ap[4 * s];
#pragma unroll
      for (int c = 0; c < K; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[4 * s * HD_LDW + c * 16], acc[c], 0, 0, 0);
    }
  }

  // ---- likelihood on the accumulators; dP to LDS (B operand of dW) and to HBM (for dd = dP W^T) ----
  float csum[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = 16 * w + 4 * lq + i;
    float llk = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f;
    const bool live = (b < a.B) && (gene < a.G);
    if (a.diag & 2) { d0 = acc[0][i] + xs[i]; d1 = acc[1][i]; d2 = acc[2][i]; llk = d0; }
    else if (live) count_elem<LK, 0>(xs[i], acc[0][i] + bias[0], acc[1][i] + bias[1], K == 3 ? acc[2][i] + bias[2] : 0.f, llk, d0, d1, d2);
    else llk = 0.f;
    d0 = live ? d0 * a.grad_scale : 0.f; d1 = live ? d1 * a.grad_scale : 0.f; d2 = live ? d2 * a.grad_scale : 0.f;
    csum[0] += d0; csum[1] += d1; csum[2] += d2;
    float* ds = Ds + b * HD_LDW + lc;
    ds[0] = d0; ds[16] = d1;
    if (K == 3) ds[32] = d2;
    if (b < a.B) {
      float* dp = a.dP + (long)b * a.ldp + gene;
      dp[0] = d0; dp[a.plane_stride] = d1;
      if (K == 3) dp[2 * a.plane_stride] = d2;
    }
    // per-cell partial over this tile's 16 genes
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) llk += __shfl_xor(llk, off, 64);
    if (lc == 0 && b < a.B) a.llk_part[(long)b * a.n_chunks + blockIdx.x] = llk;
  }
  // column sums of dP over this wave's 16 rows (bias gradient)
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    csum[c] += __shfl_xor(csum[c], 16, 64);
    csum[c] += __shfl_xor(csum[c], 32, 64);
  }
  if (lq == 0) {
#pragma unroll
    for (int c = 0; c < K; ++c) Cs[w * HD_LDW + c * 16 + lc] = csum[c];
  }
  __syncthreads();

  // ---- weight gradient: rows 16w..16w+15 of dW (decoder units) for this tile's K x 16 columns ----
  if (16 * w < Hp) {
    f32x4 gw[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) gw[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* ap = Hs + lq * HD_LDH + 16 * w + lc;      // A[m = unit][k = cell] = d[cell][unit]
    const float* bp = Ds + lq * HD_LDW + lc;               // B[k = cell][n] = dP[cell][n]
    const int ksteps = (a.diag & 4) ? 1 : (a.B + 3) / 4;
    for (int s = 0; s < ksteps; ++s) {
      const float av = ap[4 * s * HD_LDH];
#pragma unroll
      for (int c = 0; c < K; ++c) gw[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[4 * s * HD_LDW + c * 16], gw[c], 0, 0, 0);
    }
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int unit = 16 * w + 4 * lq + i;
      float* o = a.dW + (long)unit * a.ldw + gene;
#pragma unroll
      for (int c = 0; c < K; ++c) { o[(long)c * a.Gp] = gw[c][i]; sq += gw[c][i] * gw[c][i]; }
    }
    (void)sq;
  }
  if (tid < K * 16) {
    float s = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) s += Cs[ww * HD_LDW + tid];
    a.db[(long)(tid / 16) * a.Gp + g0 + (tid % 16)] = s;
  }
}

bool out_head_supported(int B, int Hp, int Gp) { return B >= 1 && B <= HD_ROWS && Hp <= HD_K && (Hp % 16) == 0 && (Gp % HD_GENES) == 0; }
int out_head_chunks(int Gp) { return Gp / HD_GENES; }

int launch_out_head_train(hipStream_t st, const OutHeadArgs& a) {
  if (!out_head_supported(a.B, a.Hp, a.Gp) || (a.ldh % 4) || (a.ldw % 4)) {
    set_error("out_head: unsupported shape");
    return SMX_ERR_INVALID;
  }
  static bool attr_done = false;
  const int lds_bytes = HD_LDS_FLOATS * (int)sizeof(float);
  if (!attr_done) {
    SMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(out_head_train_kernel<SMX_LLK_NB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    SMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(out_head_train_kernel<SMX_LLK_ZINB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    SMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(out_head_train_kernel<SMX_LLK_NBD>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    SMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(out_head_train_kernel<SMX_LLK_ZINBD>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_done = true;
  }
  const dim3 grid(a.Gp / HD_GENES), block(HD_THREADS);
  switch (a.likelihood) {
    case SMX_LLK_NB: hipLaunchKernelGGL(out_head_train_kernel<SMX_LLK_NB>, grid, block, lds_bytes, st, a); break;
    case SMX_LLK_ZINB: hipLaunchKernelGGL(out_head_train_kernel<SMX_LLK_ZINB>, grid, block, lds_bytes, st, a); break;
    case SMX_LLK_NBD: hipLaunchKernelGGL(out_head_train_kernel<SMX_LLK_NBD>, grid, block, lds_bytes, st, a); break;
    case SMX_LLK_ZINBD: hipLaunchKernelGGL(out_head_train_kernel<SMX_LLK_ZINBD>, grid, block, lds_bytes, st, a); break;
    default: set_error("out_head: unknown likelihood"); return SMX_ERR_INVALID;
  }
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

}  // namespace smx
