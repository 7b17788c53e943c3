// smx_score.hip -- importance-weighted log p(x) over STACKED posterior draws (SURVEY.md 8(f) row 1;
// posterior.py:941-976 marginal_log_prob, 100 draws by default).
//
// The draw-by-draw form runs the decoder once per draw on a batch of cells: 4 launches of 4-10 us for ~0.2 GFLOP, 25 us
// per draw.  In evaluation mode nothing couples the rows of a batch (BatchNorm applies the moving statistics, no
// dropout), so the S draws of B cells are laid side by side as S*B rows of ONE decoder pass:
//   score_draws_kernel     z[s B + b] = mu[b] + sigma[b] eps(s, b), and the latent part of log w for every row
//   product (+ bias / activation in its store path) and score_bn_act_kernel per decoder layer, S*B rows at a time
//   out_head_loss_kernel   EPI = 2: the output head's product with the count log-likelihood on its accumulators;
//                          only one partial sum per (row, 32 genes) is stored
//   iw_stack_kernel        per cell: log w of its draws, log-sum-exp folded into the running (max, sum) pair
// Same Philox counters as the draw-by-draw form (sample index in the stream word), so both see the same draws.
#include <stdlib.h>

#include "smx_internal.h"
#include "smx_device.h"
#include "smx_loss.h"

namespace smx {

// one wave per stacked row
__global__ __launch_bounds__(256) void score_draws_kernel(ScoreDrawArgs a) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (long)a.S * a.B) return;
  const int s = (int)(r / a.B), b = (int)(r % a.B);
  const uint32_t cell = a.cell_base + (uint32_t)(a.rows ? a.rows[b] : b);
  NoiseKey nk = a.nk;
  nk.stream = (a.nk.stream & 0xFFu) | (((uint32_t)(a.s0 + s) & 0xFFFFFFu) << 8);
  float lw = 0.f;
  float z_mine = 0.f;   // (mixture prior: Dp <= 64, lane d keeps z_d)
  for (int d = lane; d < a.Dp; d += 64) {
    float z = 0.f;
    if (d < a.D) {
      const float mu = a.lat[(long)b * a.ld + d];
      const float sig = softplusf(a.lat[(long)b * a.ld + a.Dp + d] + SMX_SOFTPLUS_INV_1);
      const float4 n = normal4(philox_block(nk, cell, (uint32_t)(d >> 2)));
      const float eps = (d & 3) == 0 ? n.x : (d & 3) == 1 ? n.y : (d & 3) == 2 ? n.z : n.w;
      z = mu + sig * eps;
      // log N(z; 0, I) - log N(z; mu, sigma), constants cancel (mixture prior: the q term only, the prior follows)
      lw += (a.pr_logits ? 0.f : -0.5f * z * z) + 0.5f * eps * eps + (a.pr_logits ? flog(sig) : logf(sig));
    }
    a.z[r * a.Dp + d] = z;
    z_mine = z;
  }
  lw = wave_sum(lw);
  if (a.pr_logits) {
    // log p(z) under the mixture, as scale_prior_fwd_kernel forms it: lane c keeps component c's joint log density
    const float lg = lane < a.C ? a.pr_logits[lane] : -3.0e38f;
    const float lmx = wave_max(lg);
    const float lse = lmx + flog(wave_sum(lane < a.C ? fexp(lg - lmx) : 0.f));
    float comp_mine = -3.0e38f;
    for (int c = 0; c < a.C; ++c) {
      float t = 0.f;
      if (lane < a.D) {
        const float sc = softplusf(a.pr_scale_raw[(long)c * a.Dp + lane] + SMX_SOFTPLUS_INV_1);
        const float u = (z_mine - a.pr_loc[(long)c * a.Dp + lane]) * frcp(sc);
        t = -0.5f * u * u - flog(sc);
      }
      t = wave_sum(t) + (a.pr_logits[c] - lse);
      if (lane == c) comp_mine = t;
    }
    const float cmx = wave_max(comp_mine);
    lw += cmx + flog(wave_sum(lane < a.C ? fexp(comp_mine - cmx) : 0.f));
  }
  if (lane != 0) return;
  if (a.latl) {
    const long src = a.lib_rows ? a.lib_rows[b] : b;
    const float mu = a.latl[(long)b * a.ld_l], sig = softplusf(a.latl[(long)b * a.ld_l + 1] + SMX_SOFTPLUS_INV_1);
    NoiseKey nl = a.nk_l;
    nl.stream = (a.nk_l.stream & 0xFFu) | (((uint32_t)(a.s0 + s) & 0xFFFFFFu) << 8);
    const float eps = normal4(philox_block(nl, cell, 0u)).x;
    const float l = mu + sig * eps;
    const float mp = a.library[src * 2], vp = a.library[src * 2 + 1];
    lw += -0.5f * (l - mp) * (l - mp) / vp - 0.5f * logf(vp) + 0.5f * eps * eps + logf(sig);
    a.l[r] = l;
  }
  a.lw[r] = lw;
}

int launch_score_draws(hipStream_t st, const ScoreDrawArgs& a) {
  if (a.S <= 0 || a.B <= 0 || a.Dp <= 0 || !a.lat || !a.z || !a.lw || (a.pr_logits && (a.Dp > 64 || a.C < 2 || a.C > 32 || !a.pr_loc || !a.pr_scale_raw))) {
    set_error("score_draws: bad arguments");
    return SMX_ERR_INVALID;
  }
  const long R = (long)a.S * a.B;
  hipLaunchKernelGGL(score_draws_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// evaluation-mode BatchNorm + activation of a decoder layer, in place, any number of rows (four columns per thread)
__global__ __launch_bounds__(256) void score_bn_act_kernel(ScoreBnArgs a) {
  const int q = a.Hp >> 2;
  const long total = a.R * q;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % q) * 4;
    float4 v = *reinterpret_cast<float4*>(a.h + (i / q) * a.Hp + c);
    float* e = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool live = c + j < a.H;
      const float inv = rsqrtf(a.moving_var[c + j] + a.eps);
      const float y = a.gamma[c + j] * ((e[j] - a.moving_mean[c + j]) * inv) + a.beta[c + j];
      float h = fmaxf(y, 0.f);
      if (a.leak != 0.f) h += a.leak * fminf(y, 0.f);
      e[j] = live ? h : 0.f;
    }
    *reinterpret_cast<float4*>(a.h + (i / q) * a.Hp + c) = v;
  }
}

// the LAST decoder layer: the same, written TRANSPOSED (out_t [Hp][ldt], k-major) -- the output head reads its A operand
// as 32 consecutive rows of one k per load instruction (2 cache lines) instead of 64 lanes in 64 different lines.
// gamma == nullptr: plain transpose (bias and activation already applied in the product's store path).
__global__ __launch_bounds__(256) void score_bn_act_t_kernel(ScoreBnArgs a) {
  __shared__ float tile[32][33];
  const long r0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 8 rows of 32 columns per pass
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const long r = r0 + ty + 8 * p;
    const int c = c0 + tx;
    float h = 0.f;
    if (r < a.R) {
      const float x = a.h[r * a.Hp + c];
      if (a.gamma) {
        const float inv = rsqrtf(a.moving_var[c] + a.eps);
        const float y = a.gamma[c] * ((x - a.moving_mean[c]) * inv) + a.beta[c];
        h = fmaxf(y, 0.f);
        if (a.leak != 0.f) h += a.leak * fminf(y, 0.f);
        if (c >= a.H) h = 0.f;
      } else h = x;
    }
    tile[ty + 8 * p][tx] = h;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int c = c0 + ty + 8 * p;
    const long r = r0 + tx;
    if (r < a.R) a.out_t[(long)c * a.ldt + r] = tile[tx][ty + 8 * p];
  }
}

__global__ void score_bn_act_split_kernel(ScoreBnArgs a);   // (below, beside the kernel that reads its output)
int launch_score_bn_act(hipStream_t st, const ScoreBnArgs& a) {
  if (a.out3) {
    if (a.R <= 0 || a.Hp <= 0 || (a.Hp % 4) || !a.h || (a.gamma && (!a.beta || !a.moving_mean || !a.moving_var))) {
      set_error("score_bn_act: bad arguments (split form)");
      return SMX_ERR_INVALID;
    }
    const long total = a.R * (a.Hp >> 2);
    hipLaunchKernelGGL(score_bn_act_split_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, st, a);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
  }
  if (a.out_t) {
    if (a.R <= 0 || a.Hp <= 0 || (a.Hp % 32) || !a.h || a.ldt < a.R || (a.gamma && (!a.beta || !a.moving_mean || !a.moving_var))) {
      set_error("score_bn_act: bad arguments (transposed form)");
      return SMX_ERR_INVALID;
    }
    hipLaunchKernelGGL(score_bn_act_t_kernel, dim3((unsigned)((a.R + 31) / 32), (unsigned)(a.Hp / 32)), dim3(256), 0, st, a);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
  }
  if (a.R <= 0 || a.Hp <= 0 || (a.Hp % 4) || !a.h || !a.gamma || !a.beta || !a.moving_mean || !a.moving_var) {
    set_error("score_bn_act: bad arguments");
    return SMX_ERR_INVALID;
  }
  const long total = a.R * (a.Hp >> 2);
  hipLaunchKernelGGL(score_bn_act_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Output head of a stacked pass: product + count log-likelihood, likelihood partials only.
//
// Two measured facts shape it (profiles/r02_scoring_path.txt, tools/coexec.hip):
//  * the training kernel's direct-operand form (smx_headloss.hip, EPI = 2) and an LDS-shared f32 form both stop at ~0.43
//    of the f32 MFMA peak here.  v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate and does not overlap with the
//    partner wave's vector instructions (an MFMA wave and a v_fma_f32 wave on one SIMD take the SUM of their times);
//    the likelihood is ~230 vector instructions per element, about as many cycles as the f32 MFMAs of the tile, and
//    the two add up.  bf16 MFMAs do run beside vector work.
//  * so the product is formed from bf16 MFMAs on operands split three ways, x = x0 + x1 + x2 (each the bf16 rounding
//    of what is left), keeping the six products x0y0, x0y1, x1y0, x0y2, x2y0, x1y1: what is dropped (x1y2, x2y1, x2y2)
//    is below 2^-23 of |x y| -- the rounding of one f32 multiply -- and the sum is accumulated in f32 as before.
//    6 MFMAs of 32 cycles per 16 k instead of 8 of 64: 0.375 of the cycles, on the matrix pipe, beside the likelihood.
// The operands arrive split: the last decoder layer's launch writes its output as three bf16 arrays [3][R][Hp]
// (score_bn_act_kernel), and W is split once per call into per-(gene tile, 32-deep slab) images in the exact order the
// MFMA B operand is read (score_split_w_kernel), so the hot loop converts nothing.  A workgroup of 4 waves takes 128
// rows x 32 genes x all planes: the gene tile's whole W image (K <= 128: up to 72 KB with 3 planes, two workgroups per
// CU) goes global -> LDS by LDS-DMA and each wave loads the 32 rows of its A operand into registers, ALL up front (a
// slab-by-slab double buffer was tried first: a memory round trip takes ~3 us under this load, more than the 0.5 us of
// MFMAs of a slab, and it waited once per slab: 4.0 ms per 128 x 1000 draws against 2.6 ms this way, 3.3 ms for the
// f32 forms).  What remains is the likelihood's vector work: ~230 instructions per element, 60 % of the vector peak.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float x, __bf16& x0, __bf16& x1, __bf16& x2) {
  x0 = (__bf16)x;
  const float r1 = x - (float)x0;   // exact
  x1 = (__bf16)r1;
  x2 = (__bf16)(r1 - (float)x1);
}

// evaluation-mode BatchNorm + activation of the LAST decoder layer, written as the three-way bf16 split [3][R][Hp]
// (gamma == nullptr: the split alone; bias and activation were applied in the product's store path)
__global__ __launch_bounds__(256) void score_bn_act_split_kernel(ScoreBnArgs a) {
  const int q = a.Hp >> 2;
  // a thread keeps ONE column quad (q divides 256 at Hp = 32, 64, 128): its four columns' BatchNorm constants are loaded once, ahead of the
  // rows, and the index arithmetic is a shift -- as a flat walk every element paid a 64-bit division and 16 loads of per-column constants
  // behind its row's load (11 us for 6.5 MB in, 9.8 MB out)
  const bool fixed = (256 % q) == 0;
  const int rpb = fixed ? 256 / q : 1;   // rows per block and pass
  const int c = fixed ? ((int)threadIdx.x % q) * 4 : 0;
  float g4[4] = {0.f, 0.f, 0.f, 0.f}, b4[4] = {0.f, 0.f, 0.f, 0.f}, m4[4] = {0.f, 0.f, 0.f, 0.f}, i4[4] = {0.f, 0.f, 0.f, 0.f};
  if (fixed && a.gamma) {
    float var[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { g4[j] = a.gamma[c + j]; b4[j] = a.beta[c + j]; m4[j] = a.moving_mean[c + j]; var[j] = a.moving_var[c + j]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) i4[j] = rsqrtf(var[j] + a.eps);
  }
  auto one = [&](long o, int cc, bool pre) {
    float4 v = *reinterpret_cast<const float4*>(a.h + o);
    float* e = reinterpret_cast<float*>(&v);
    __bf16 t[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float hval = e[j];
      if (a.gamma) {
        const float inv = pre ? i4[j] : rsqrtf(a.moving_var[cc + j] + a.eps);
        const float y = (pre ? g4[j] : a.gamma[cc + j]) * ((e[j] - (pre ? m4[j] : a.moving_mean[cc + j])) * inv) + (pre ? b4[j] : a.beta[cc + j]);
        hval = fmaxf(y, 0.f);
        if (a.leak != 0.f) hval += a.leak * fminf(y, 0.f);
        if (cc + j >= a.H) hval = 0.f;
      }
      split3(hval, t[0][j], t[1][j], t[2][j]);
    }
#pragma unroll
    for (int T = 0; T < 3; ++T) *reinterpret_cast<uint2*>(a.out3 + (long)T * a.R * a.Hp + o) = *reinterpret_cast<const uint2*>(t[T]);
  };
  if (fixed) {
    for (long r = (long)blockIdx.x * rpb + (int)threadIdx.x / q; r < a.R; r += (long)gridDim.x * rpb) one(r * a.Hp + c, c, true);
    return;
  }
  const long total = a.R * q;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cc = (int)(i % q) * 4;
    one((i / q) * a.Hp + cc, cc, false);
  }
}

// A ONE-layer decoder with BatchNorm in one launch: the draws, the product with the layer's [Dp][Hp] weights, evaluation-mode BatchNorm + activation and the
// three-way bf16 split of 32 stacked rows per workgroup -- score_draws_kernel + the product + score_bn_act_split_kernel (8.5 + 5.7 + 8.2 us of launches at 128 cells x
// 100 draws; z and the f32 activations never exist in memory, and the host, which paces the call's short launches, queues two fewer).  The draws are
// score_draws_kernel's, bit for bit (two rows per wave when Dp <= 32: half_wave_sum is wave_sum's tree over a half whose other half adds zeros); the product is
// plain f32 multiply-adds in ascending k (32 per output).
template <bool PAIR>   // two rows per pass, one per half of the wave (Dp <= 32), or one (Dp <= 64)
__global__ __launch_bounds__(256) void score_decoder1_kernel(ScoreDec1Args a) {
  __shared__ __attribute__((aligned(16))) float zs[32][68];
  extern __shared__ __attribute__((aligned(16))) float ws[];   // [Dp][Hp]
  const ScoreDrawArgs& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long R = (long)d.S * d.B, r0 = (long)blockIdx.x * 32;
  // ---- every request first: the rows' cells, their (mu, s_raw), the layer's weights, the thread's four columns' BatchNorm constants (pass by pass behind
  // one another -- the stores of lw between them kept the compiler from hoisting -- the rows' two dependent round trips came eight times over: 18.6 us) ----
  constexpr int NPASS = PAIR ? 4 : 8;
  const int half = lane >> 5, dd = PAIR ? (lane & 31) : lane;
  const bool d_ok = dd < d.D;
  int bs[NPASS], ss[NPASS]; uint32_t cells[NPASS]; bool row_ok[NPASS];
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    const long r = r0 + 8 * wv + (PAIR ? 2 * p + half : p);
    row_ok[p] = r < R;
    const long rc = row_ok[p] ? r : R - 1;
    ss[p] = (int)(rc / d.B); bs[p] = (int)(rc - (long)ss[p] * d.B);
  }
  if (d.rows) {
#pragma unroll
    for (int p = 0; p < NPASS; ++p) cells[p] = (uint32_t)d.rows[bs[p]];
  } else {
#pragma unroll
    for (int p = 0; p < NPASS; ++p) cells[p] = (uint32_t)bs[p];
  }
  float mus[NPASS], srs[NPASS];
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    const float* lp = d.lat + (long)bs[p] * d.ld + (d_ok ? dd : 0);
    mus[p] = lp[0]; srs[p] = lp[d.Dp];
  }
  const int q = a.Hp >> 2;
  for (int it = tid; it < d.Dp * q; it += 256) {
    const int k = it / q, c = (it - k * q) * 4;
    *reinterpret_cast<float4*>(ws + k * a.Hp + c) = *reinterpret_cast<const float4*>(a.W + (long)k * a.ldw + c);
  }
  const int cq = tid & 31, rg = tid >> 5, c0 = 4 * cq;
  const bool col_ok = cq < q;
  float g4[4], b4[4], m4[4], i4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = col_ok ? c0 + j : 0;
    g4[j] = a.gamma[c]; b4[j] = a.beta[c]; m4[j] = a.moving_mean[c]; i4[j] = a.moving_var[c];
  }
  // ---- the draws of the block's 32 rows: wave wv takes rows 8 wv .. 8 wv + 7 ----
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    const int rl = 8 * wv + (PAIR ? 2 * p + half : p);
    NoiseKey nk = d.nk;
    nk.stream = (d.nk.stream & 0xFFu) | (((uint32_t)(d.s0 + ss[p]) & 0xFFFFFFu) << 8);
    float lw = 0.f, z = 0.f;
    if (d_ok) {
      const float sig = softplusf(srs[p] + SMX_SOFTPLUS_INV_1);
      const float4 n = normal4(philox_block(nk, d.cell_base + cells[p], (uint32_t)(dd >> 2)));
      const float eps = (dd & 3) == 0 ? n.x : (dd & 3) == 1 ? n.y : (dd & 3) == 2 ? n.z : n.w;
      z = mus[p] + sig * eps;
      lw = -0.5f * z * z + 0.5f * eps * eps + logf(sig);
    }
    if (dd < d.Dp) zs[rl][dd] = z;
    lw = PAIR ? half_wave_sum(lw) : wave_sum(lw);
    if (dd == 0 && row_ok[p]) d.lw[r0 + rl] = lw;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) i4[j] = rsqrtf(i4[j] + a.eps);
  __syncthreads();
  if (!col_ok) return;
  // ---- product: thread (row group rg, column quad cq): rows rg, rg + 8, rg + 16, rg + 24 ----
  float acc[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[k][j] = 0.f;
  for (int k0 = 0; k0 < d.Dp; k0 += 4) {
    float4 w4[4], z4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w4[e] = *reinterpret_cast<const float4*>(ws + (k0 + e) * a.Hp + c0);
#pragma unroll
    for (int k = 0; k < 4; ++k) z4[k] = *reinterpret_cast<const float4*>(&zs[rg + 8 * k][k0]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float zz[4] = {z4[k].x, z4[k].y, z4[k].z, z4[k].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[k][0] = __builtin_fmaf(zz[e], w4[e].x, acc[k][0]);
        acc[k][1] = __builtin_fmaf(zz[e], w4[e].y, acc[k][1]);
        acc[k][2] = __builtin_fmaf(zz[e], w4[e].z, acc[k][2]);
        acc[k][3] = __builtin_fmaf(zz[e], w4[e].w, acc[k][3]);
      }
    }
  }
  // ---- BatchNorm (moving statistics), activation, split, stores ----
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long r = r0 + rg + 8 * k;
    if (r >= R) continue;
    __bf16 t[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float y = g4[j] * ((acc[k][j] - m4[j]) * i4[j]) + b4[j];
      float hval = fmaxf(y, 0.f);
      if (a.leak != 0.f) hval += a.leak * fminf(y, 0.f);
      if (c0 + j >= a.H) hval = 0.f;
      split3(hval, t[0][j], t[1][j], t[2][j]);
    }
#pragma unroll
    for (int T = 0; T < 3; ++T) *reinterpret_cast<uint2*>(a.out3 + (long)T * R * a.Hp + r * a.Hp + c0) = *reinterpret_cast<const uint2*>(t[T]);
  }
}

bool score_decoder1_supported(int Dp, int Hp) { return Dp > 0 && Dp % 4 == 0 && Dp <= 64 && Hp > 0 && Hp % 32 == 0 && Hp <= 128; }

int launch_score_decoder1(hipStream_t st, const ScoreDec1Args& a) {
  const ScoreDrawArgs& d = a.d;
  if (!score_decoder1_supported(d.Dp, a.Hp) || d.S <= 0 || d.B <= 0 || !d.lat || !d.lw || d.pr_logits || d.latl || !a.W || !a.gamma || !a.beta || !a.moving_mean ||
      !a.moving_var || !a.out3 || a.ldw < a.Hp) {
    set_error("score_decoder1: bad arguments");
    return SMX_ERR_INVALID;
  }
  const long R = (long)d.S * d.B;
  const dim3 grid((unsigned)((R + 31) / 32));
  const size_t lds = (size_t)d.Dp * a.Hp * sizeof(float);
  if (d.Dp <= 32) hipLaunchKernelGGL(score_decoder1_kernel<true>, grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL(score_decoder1_kernel<false>, grid, dim3(256), lds, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// W [Hp][k * Gp] -> images [gene tile][slab][term][plane][k block n][lane half h][column][8 k] of bf16: the 16 bytes at
// (term, plane, n, h, column i) are what lane (i, h) feeds to v_mfma_f32_32x32x16_bf16 for k = 32 slab + 16 n + 8 h ..+7
__global__ __launch_bounds__(256) void score_split_w_kernel(ScoreSplitWArgs a) {
  const long total = (long)a.n_gt * a.nslab * a.NP * 128;   // items of 8 k
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
    const int c = (int)(it & 31), h = (int)(it >> 5) & 1, n = (int)(it >> 6) & 1;
    long rest = it >> 7;
    const int p = (int)(rest % a.NP); rest /= a.NP;
    const int slab = (int)(rest % a.nslab), gt = (int)(rest / a.nslab);
    const float* src = a.W + (long)(32 * slab + 16 * n + 8 * h) * a.ldw + (long)p * a.Gp + 32 * gt + c;
    __bf16 t[3][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) split3(src[(long)k * a.ldw], t[0][k], t[1][k], t[2][k]);
#pragma unroll
    for (int T = 0; T < 3; ++T) {
      const long dst = (((((long)(gt * a.nslab + slab) * 3 + T) * a.NP + p) * 2 + n) * 2 + h) * 32 + c;
      *reinterpret_cast<uint4*>(a.img + dst * 8) = *reinterpret_cast<const uint4*>(t[T]);
    }
  }
}

int launch_score_split_w(hipStream_t st, const ScoreSplitWArgs& a) {
  if (a.n_gt <= 0 || a.nslab <= 0 || (a.NP != 2 && a.NP != 3) || !a.W || !a.img) { set_error("score_split_w: bad arguments"); return SMX_ERR_INVALID; }
  const long total = (long)a.n_gt * a.nslab * a.NP * 128;
  hipLaunchKernelGGL(score_split_w_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// The products of one wave's 32 x 32 tile from the W image in LDS and the wave's A operand in registers (smallest terms first).
template <int NP, int NSLAB>
__device__ inline void score_products(const uint4* bl, const uint4 (&av)[NSLAB][3][2], f32x16 (&acc)[NP], int i, int h) {
  constexpr int UNITS = 3 * NP * 128;
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
#pragma unroll
  for (int t = 0; t < NSLAB; ++t) {
    const uint4* bb = &bl[t * UNITS];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        uint4 braw[3];
#pragma unroll
        for (int T = 0; T < 3; ++T) braw[T] = bb[(((T * NP + p) * 2 + n) * 2 + h) * 32 + i];
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&braw[0]), b1 = *reinterpret_cast<const bf16x8*>(&braw[1]),
                     b2 = *reinterpret_cast<const bf16x8*>(&braw[2]);
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&av[t][0][n]), a1 = *reinterpret_cast<const bf16x8*>(&av[t][1][n]),
                     a2 = *reinterpret_cast<const bf16x8*>(&av[t][2][n]);
        // smallest terms first
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[p], 0, 0, 0);
      }
  }
}
// The likelihood of the wave's 32 x 32 tile and its per-row sums over the 32 columns: register r of plane p is row (r & 3) + 8 (r >> 2) + 4 h, column i.
// lgamma(x + r) - lgamma(r) through the per-wave queue of the NON-ZERO counts (smx_loss.h: 88-93 % of the counts are zero, for which the
// straight-line form still runs its 8-step recurrence -- a third of the vector instructions; lq == nullptr: the straight-line form)
template <int LK, int NP>
__device__ inline void score_tile_llk(const ScoreHeadArgs& a, const f32x16 (&acc)[NP], const float (&xs)[16], const float (&bias)[NP], bool live,
                                      float2* lq, int i, int h, int m0, int gt) {
  float L[16];
#pragma unroll
  for (int c = 0; c < 16; c += 8) {
    float p0[8], p1[8], p2[8], d0[8], d1[8], d2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      p0[j] = acc[0][c + j] + bias[0];
      p1[j] = acc[1][c + j] + bias[1];
      p2[j] = NP == 3 ? acc[NP - 1][c + j] + bias[NP - 1] : 0.f;
    }
    typedef float Vec[8];
    count_elem_vec<LK, 0, 8>(*(const Vec*)(xs + c), p0, p1, p2, *(Vec*)(L + c), d0, d1, d2, lq);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) L[r] = live ? L[r] : 0.f;
  // per-row sums over the 32 columns (the 32 lanes of this half): a halving exchange -- 8 + 4 + 2 + 1 + 1 shuffles
  // instead of 16 x 5; afterwards lane i holds row r = bits 4..1 of i
  float v8[8], v4[4], v2[2];
  {
    const bool up = (i & 16) != 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const float keep = up ? L[8 + q] : L[q], send = up ? L[q] : L[8 + q]; v8[q] = keep + __shfl_xor(send, 16, 64); }
  }
  {
    const bool up = (i & 8) != 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float keep = up ? v8[4 + q] : v8[q], send = up ? v8[q] : v8[4 + q]; v4[q] = keep + __shfl_xor(send, 8, 64); }
  }
  {
    const bool up = (i & 4) != 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) { const float keep = up ? v4[2 + q] : v4[q], send = up ? v4[q] : v4[2 + q]; v2[q] = keep + __shfl_xor(send, 4, 64); }
  }
  float tot;
  {
    const bool up = (i & 2) != 0;
    const float keep = up ? v2[1] : v2[0], send = up ? v2[0] : v2[1];
    tot = keep + __shfl_xor(send, 2, 64);
  }
  tot += __shfl_xor(tot, 1, 64);
  {
    const int r = (i >> 1) & 15;   // bit 4 chose r >= 8, bit 3 the upper half of those, ...
    const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if ((i & 1) == 0 && row < a.R) a.llk_part[(long)row * a.n_gt + gt] = tot;
  }
}
// the counts of a wave's 16 rows (of its 32: the lane half's) x its column, in two steps: the cells' dataset rows (to be requested BEFORE the tile's other
// loads: the counts' addresses wait for them, and a wait for the youngest request is a wait for everything in front of it), then the counts.  The storage
// format's branch is outside the loops: inside them it put a branch and a wait between every two requests (sixteen round trips in a row for u16 counts).
__device__ inline void score_count_rows(const ScoreHeadArgs& a, int m0, int h, int (&src)[16]) {
  const int base = m0 % a.row_mod;
  int cell[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    cell[r] = base + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (cell[r] >= a.row_mod) cell[r] -= a.row_mod;
    if (cell[r] >= a.row_mod) cell[r] %= a.row_mod;   // (fewer than 32 cells per draw)
  }
  // (no branch on `rows`: behind a join the compiler's wait for these requests becomes a wait for every request in flight -- the image and the A operand,
  // issued after them, included -- and the counts' requests, which need these, would leave one round trip late.  Without the table: a load of word 0 of the counts, unused)
  const bool tab = a.rows != nullptr;
  const int32_t* rp = tab ? a.rows : reinterpret_cast<const int32_t*>(a.X);
#pragma unroll
  for (int r = 0; r < 16; ++r) src[r] = rp[tab ? cell[r] : 0];
#pragma unroll
  for (int r = 0; r < 16; ++r) src[r] = tab ? src[r] : cell[r];
}
__device__ inline void score_count_values(const ScoreHeadArgs& a, const int (&src)[16], int col, float (&xs)[16]) {
  if (a.x_u16) {
    const uint16_t* X = reinterpret_cast<const uint16_t*>(a.X);
    uint16_t raw[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) raw[r] = X[(long)src[r] * a.ldx + col];
#pragma unroll
    for (int r = 0; r < 16; ++r) xs[r] = (float)raw[r];
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) xs[r] = a.X[(long)src[r] * a.ldx + col];
  }
}

template <int LK, int NSLAB>
__global__ __launch_bounds__(256) void score_head_kernel(ScoreHeadArgs a) {
  constexpr int NP = (LK == SMX_LLK_ZINB || LK == SMX_LLK_ZINBD) ? 3 : 2;
  constexpr int UNITS = 3 * NP * 128;             // 16-byte units of one slab image
  constexpr int ALL = NSLAB * UNITS;              // ... of the gene tile's whole K (72 KB at Hp = 128 with 3 planes)
  extern __shared__ uint4 bl[];
  const int lane = threadIdx.x & 63, w = (threadIdx.x >> 6) & 3;
  const int i = lane & 31, h = lane >> 5;
  // blocks 8 apart share an XCD: they take the row blocks of ONE gene tile (its W images stay in that L2), in groups
  // of row blocks over which ALL of the XCD's gene tiles pass before the next group (A stays in the L2 meanwhile)
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int per_grp = a.rb_group * a.gt_per_xcd;
  const int grp = idx / per_grp, rem = idx % per_grp;
  const int rb = grp * a.rb_group + rem % a.rb_group, gt = (rem / a.rb_group) * 8 + xcd;
  if (gt >= a.n_gt || rb >= a.n_rb) return;
  const int m0 = rb * 128 + 32 * w, n0 = gt * 32, col = n0 + i;
  const int arow = min(m0 + i, a.R - 1);   // rows beyond the pass compute garbage that is never stored
  const uint4* wimg = reinterpret_cast<const uint4*>(a.Wimg) + (long)gt * ALL;
  const long aterm = (long)a.R * a.Hp;   // bf16 elements between the terms of A

  // ---- every load of the tile is requested up front (K <= 128: the whole W image of the gene tile fits in LDS and the
  // wave's A operand in registers).  A memory round trip under this load takes ~3 us, longer than the MFMAs of a slab:
  // a slab-by-slab pipeline waited for it once per slab; this way a workgroup waits once and the CU's other workgroup
  // computes meanwhile ----
  int src[16];
  score_count_rows(a, m0, h, src);
  // W image: global -> LDS directly (no staging registers; one wave-instruction moves 1 KB to wave-uniform base + 16 lane)
  // (no branch around a last partial round -- a piece beyond the image re-reads its last unit into the slack the launcher allocates behind it: behind a
  // join the compiler's waits count nothing and become waits for everything)
  constexpr int UPT = (ALL + 255) / 256;
#pragma unroll
  for (int u = 0; u < UPT; ++u) {
    const int q0 = u * 256 + w * 64;   // first unit of this wave's piece (ALL is a multiple of 64)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wimg + min(q0 + lane, ALL - 1)),
                                     (__attribute__((address_space(3))) void*)(bl + q0), 16, 0, 0);
  }
  uint4 av[NSLAB][3][2];
  {
    const __bf16* ap = a.A3 + (long)arow * a.Hp + 8 * h;
#pragma unroll
    for (int t = 0; t < NSLAB; ++t)
#pragma unroll
      for (int T = 0; T < 3; ++T)
#pragma unroll
        for (int n = 0; n < 2; ++n) av[t][T][n] = *reinterpret_cast<const uint4*>(ap + T * aterm + 32 * t + 16 * n);
  }
  __builtin_amdgcn_sched_barrier(0);   // (the counts' addresses, which wait for the rows, behind the requests above)
  float xs[16];
  score_count_values(a, src, col, xs);
  __syncthreads();   // (waits for this wave's loads -- the image pieces included -- then for the other waves')

  f32x16 acc[NP];
  score_products<NP, NSLAB>(bl, av, acc, i, h);

  // the queue takes the place of the W image, which every wave has finished with behind this barrier (knob no_score_queue: the straight-line form)
  __syncthreads();
  float2* const lq = a.no_queue ? nullptr : reinterpret_cast<float2*>(bl);
  float bias[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) bias[p] = a.bias[(long)p * a.Gp + col];
  score_tile_llk<LK, NP>(a, acc, xs, bias, col < a.G, lq, i, h, m0, gt);
}

// The same tiles as a WALK: a workgroup of eight waves keeps its gene tile's W image in LDS and walks a range of 256-row blocks under it.  The wave's A
// operand of the NEXT block is requested as soon as the products of this one have issued (the registers are free from then on) and lands
// during the likelihood, so no wave waits for memory inside the walk; there is no workgroup barrier inside it either (the queue is each wave's own
// and sits behind the image), so the two waves of a SIMD drift apart and one's products run beside the other's likelihood.  score_head_kernel
// reloads 168 KB per 128 x 32 tile (1.06 GB per launch of 128 cells x 100 draws x 1998 genes) and its workgroup pairs keep step with each other:
// load, products and likelihood of both follow one another (32 k cycles per pair of tiles against 23 k of issue).
template <int LK, int NSLAB>
__global__ __launch_bounds__(512) void score_walk_kernel(ScoreHeadArgs a) {
  constexpr int NP = (LK == SMX_LLK_ZINB || LK == SMX_LLK_ZINBD) ? 3 : 2;
  constexpr int UNITS = 3 * NP * 128;
  constexpr int ALL = NSLAB * UNITS;
  extern __shared__ uint4 bl[];   // the image, then the eight waves' queues
  const int lane = threadIdx.x & 63, w = (threadIdx.x >> 6) & 7;
  const int i = lane & 31, h = lane >> 5;
  // blocks 8 apart share an XCD: it takes a contiguous stretch of the (row range, gene tile) pairs in row-range-major order, so that the A rows
  // its workgroups walk are the same few MB of its L2
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int pair = xcd * a.wg_per_xcd + idx;
  if (idx >= a.wg_per_xcd || pair >= a.n_gt * a.n_split) return;
  const int split = pair / a.n_gt, gt = pair - split * a.n_gt;
  const int rb0 = (int)((long)split * a.n_rb / a.n_split), rb1 = (int)((long)(split + 1) * a.n_rb / a.n_split);   // (n_rb: blocks of 256 rows here)
  const int col = gt * 32 + i;
  const uint4* wimg = reinterpret_cast<const uint4*>(a.Wimg) + (long)gt * ALL;
  const long aterm = (long)a.R * a.Hp;
  int src[16];
  score_count_rows(a, rb0 * 256 + 32 * w, h, src);
  constexpr int UPT = (ALL + 511) / 512, ALLP = UPT * 512;   // (the image's slack, as in score_head_kernel)
#pragma unroll
  for (int u = 0; u < UPT; ++u) {
    const int q0 = u * 512 + w * 64;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wimg + min(q0 + lane, ALL - 1)),
                                     (__attribute__((address_space(3))) void*)(bl + q0), 16, 0, 0);
  }
  uint4 av[NSLAB][3][2];
  // (with four slabs the last one's registers are requested at the top of the block instead, behind three slabs' MFMAs (~1.5 us): all four held through the
  // likelihood spilled, and so did requesting them between its halves -- an accumulator tuple stays allocated until its last element is read)
  constexpr int NEARLY = NSLAB == 4 ? 3 : NSLAB;
  auto load_a = [&](int rb, int t_lo, int t_hi) {
    const int arow = min(rb * 256 + 32 * w + i, a.R - 1);   // rows beyond the pass compute garbage that is never stored
    const __bf16* ap = a.A3 + (long)arow * a.Hp + 8 * h;
#pragma unroll
    for (int t = t_lo; t < t_hi; ++t)
#pragma unroll
      for (int T = 0; T < 3; ++T)
#pragma unroll
        for (int n = 0; n < 2; ++n) av[t][T][n] = *reinterpret_cast<const uint4*>(ap + T * aterm + 32 * t + 16 * n);
  };
  load_a(rb0, 0, NEARLY);
  __builtin_amdgcn_sched_barrier(0);   // (the counts' addresses, which wait for the rows, behind the requests above)
  float xs[16];
  int base = (rb0 * 256 + 32 * w) % a.row_mod;
  score_count_values(a, src, col, xs);
  float bias[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) bias[p] = a.bias[(long)p * a.Gp + col];
  const bool live = col < a.G;
  float2* const lq = a.no_queue ? nullptr : reinterpret_cast<float2*>(bl + ALLP);
  __syncthreads();   // (the image: this wave's pieces, then the other waves')
  // (measured and not kept: a head start of one matrix phase for the second wave of every SIMD, 223 against 226 us per call; a word per SIMD that a wave
  // takes for its products so that its partner is in its likelihood meanwhile, 216-225 against 218-226: neither beyond the spread between two runs)
  for (int rb = rb0; rb < rb1; ++rb) {
    const int m0 = rb * 256 + 32 * w;
    if (m0 >= a.R) break;   // (wave-uniform: the last block's waves beyond the pass)
    f32x16 acc[NP];
    if (NEARLY < NSLAB) {
      load_a(rb, NEARLY, NSLAB);
      __builtin_amdgcn_sched_barrier(0);   // (the requests stay here: the scheduler would sink them to their first use)
    }
    score_products<NP, NSLAB>(bl, av, acc, i, h);
    if (rb + 1 < rb1) load_a(rb + 1, 0, NEARLY);
    __builtin_amdgcn_sched_barrier(0);
    score_tile_llk<LK, NP>(a, acc, xs, bias, live, lq, i, h, m0, gt);
    // (the counts repeat from block to block when 256 is a multiple of the cells per draw: the usual case)
    const int base_n = (m0 + 256) % a.row_mod;
    if (rb + 1 < rb1 && base_n != base) {
      score_count_rows(a, m0 + 256, h, src);
      score_count_values(a, src, col, xs);
      base = base_n;
    }
  }
}

bool score_head_supported(int Hp, int Gp) { return Hp > 0 && Hp % 32 == 0 && Hp <= 128 && Gp % 32 == 0; }

template <int LK, int NSLAB>
static int launch_score_head_t(hipStream_t st, const ScoreHeadArgs& a, dim3 grid, bool walk) {
  constexpr int NP = (LK == SMX_LLK_ZINB || LK == SMX_LLK_ZINBD) ? 3 : 2;
  constexpr size_t units = (size_t)NSLAB * 3 * NP * 128;
  constexpr size_t lds0 = (units + 255) / 256 * 256 * 16;   // (whole rounds of the workgroup's pieces: the kernels' slack)
  constexpr size_t lds = lds0 > 4 * 64 * 8 * sizeof(float2) ? lds0 : 4 * 64 * 8 * sizeof(float2);   // (the W image, then the four waves' non-zero queues)
  constexpr size_t lds_walk = (units + 511) / 512 * 512 * 16 + 8 * 64 * 8 * sizeof(float2);          // (the W image and, behind it, the eight waves' queues)
  static bool raised = false, raised_walk = false;   // (above 64 KB of dynamic LDS a kernel needs the attribute once)
  ScoreHeadArgs b = a;
  b.no_queue = tuning_on("no_score_queue") ? 1 : 0;
  if (walk) {
    if (!raised_walk && lds_walk > 64 * 1024) {
      SMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&score_walk_kernel<LK, NSLAB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_walk));
      raised_walk = true;
    }
    hipLaunchKernelGGL((score_walk_kernel<LK, NSLAB>), grid, dim3(512), lds_walk, st, b);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
  }
  if (!raised && lds > 64 * 1024) {
    SMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&score_head_kernel<LK, NSLAB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    raised = true;
  }
  hipLaunchKernelGGL((score_head_kernel<LK, NSLAB>), grid, dim3(256), lds, st, b);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}
template <int LK>
static int launch_score_head_lk(hipStream_t st, const ScoreHeadArgs& a, dim3 grid, bool walk) {
  switch (a.Hp / 32) {
    case 1: return launch_score_head_t<LK, 1>(st, a, grid, walk);
    case 2: return launch_score_head_t<LK, 2>(st, a, grid, walk);
    case 3: return launch_score_head_t<LK, 3>(st, a, grid, walk);
    default: return launch_score_head_t<LK, 4>(st, a, grid, walk);
  }
}

int launch_score_head(hipStream_t st, const ScoreHeadArgs& a_in) {
  ScoreHeadArgs a = a_in;
  if (!score_head_supported(a.Hp, a.Gp) || a.R <= 0 || a.row_mod <= 0 || !a.A3 || !a.Wimg || !a.bias || !a.X || !a.llk_part) {
    set_error("score_head: bad shapes");
    return SMX_ERR_INVALID;
  }
  a.n_gt = a.Gp / 32;
  // Which form: the walk (score_walk_kernel; every gene tile's rows in n_split ranges, one workgroup of eight waves each, one workgroup per CU: 72 KB of
  // image + 32 KB of queues) when its rounds of workgroups x blocks per workgroup come to less than the tile form's rounds of 512 resident tiles.  Fitted to
  // tools/dev/score_form_sweep.py at three planes x four slabs (both forms scale alike with those): ~11 us per 256-row block + ~7 us per workgroup against
  // ~15 us per round of 128-row tiles -- at 128 cells the walk pays from ~15 draws (126.6 against 141.6 us per call at 25, 223 against 281 at 100); below, it
  // has too few workgroups (forced at 10 draws: 108-135 against 105).
  const int walk_knob = (int)tuning("score_walk", -1.0);   // 0: never; n > 0: n ranges
  const int n_rb256 = (a.R + 255) / 256;
  int n_split = 0;
  if (walk_knob > 0) n_split = std::min(walk_knob, n_rb256);
  else if (walk_knob < 0) {
    double best = 15.0 * std::max(0.3, (double)((a.R + 127) / 128) * a.n_gt / 512.0);
    for (int s = 1; s <= 8 && s <= n_rb256; ++s) {
      const long rounds = ((long)a.n_gt * s + 255) / 256;
      const double cost = (double)rounds * (7.0 + 10.9 * ((double)a.R / 256.0) / s);
      if (cost < best - 0.5) { best = cost; n_split = s; }
    }
  }
  dim3 grid;
  if (n_split > 0) {
    a.n_rb = n_rb256;
    a.n_split = n_split;
    a.wg_per_xcd = (a.n_gt * n_split + 7) / 8;
    grid = dim3((unsigned)(8 * a.wg_per_xcd));
  } else {
    a.n_rb = (a.R + 127) / 128;
    // (row blocks per L2 group: measured 4 / 8 / 16 / 32 / all within 3 % of each other once the loads are issued up
    // front -- the A operand's re-reads are served by the Infinity Cache at no visible cost; default: one group)
    static const int rbg = std::max(1, (int)tuning("score_rb_group", (double)(1 << 30)));
    a.rb_group = std::min(rbg, a.n_rb);
    a.gt_per_xcd = (a.n_gt + 7) / 8;
    const int n_grp = (a.n_rb + a.rb_group - 1) / a.rb_group;
    grid = dim3((unsigned)(8 * n_grp * a.rb_group * a.gt_per_xcd));
  }
  const bool walk = n_split > 0;
  switch (a.likelihood) {
    case SMX_LLK_NB: return launch_score_head_lk<SMX_LLK_NB>(st, a, grid, walk);
    case SMX_LLK_ZINB: return launch_score_head_lk<SMX_LLK_ZINB>(st, a, grid, walk);
    case SMX_LLK_NBD: return launch_score_head_lk<SMX_LLK_NBD>(st, a, grid, walk);
    case SMX_LLK_ZINBD: return launch_score_head_lk<SMX_LLK_ZINBD>(st, a, grid, walk);
    default: set_error("score_head: unknown likelihood"); return SMX_ERR_INVALID;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// scvi (scvi.py:108-171): the rate is softmax over ALL genes of a row times exp(l), so the likelihood cannot ride on a
// 32-gene tile of the product.  The k raw planes of the stacked rows are materialised (k products over S B rows), then
// one workgroup per row keeps its planes in registers: max, sum, rate / dispersion / gate, the NBD / ZINBD log-likelihood
// of the row's counts and its sum -- the activated planes and the softmax are never stored.
__device__ inline float blk_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ inline float blk_max(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

template <int NV, int LK>
__global__ __launch_bounds__(256) void scvi_score_rows_kernel(ScviScoreArgs a) {
  constexpr bool ZI = (LK == SMX_LLK_ZINBD);
  __shared__ float sh[4];
  const int b = blockIdx.x;
  const float* raw = a.raw + (long)b * a.ld;
  const int cell = b % a.row_mod;
  const long src = a.rows ? a.rows[cell] : cell;
  float4 r0[NV], r1[NV], r2[NV], xv[NV];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int g = (threadIdx.x + 256 * j) * 4;
    const bool ok = g < a.Gp;
    r0[j] = ok ? *reinterpret_cast<const float4*>(raw + g) : zero4();
    r1[j] = ok ? *reinterpret_cast<const float4*>(raw + a.plane_stride + g) : zero4();
    r2[j] = (ok && ZI) ? *reinterpret_cast<const float4*>(raw + 2 * a.plane_stride + g) : zero4();
    if (!ok) xv[j] = z4;
    else if (a.x_u16) { const ushort4 u = *reinterpret_cast<const ushort4*>(reinterpret_cast<const uint16_t*>(a.X) + src * a.ldx + g); xv[j] = make_float4(u.x, u.y, u.z, u.w); }
    else xv[j] = *reinterpret_cast<const float4*>(a.X + src * a.ldx + g);
  }
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int g = (threadIdx.x + 256 * j) * 4;
    const float v[4] = {r0[j].x, r0[j].y, r0[j].z, r0[j].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) if (g + e < a.G) mx = fmaxf(mx, v[e]);
  }
  mx = blk_max(mx, sh);
  float ex[NV][4];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int g = (threadIdx.x + 256 * j) * 4;
    const float v[4] = {r0[j].x, r0[j].y, r0[j].z, r0[j].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ex[j][e] = (g + e < a.G) ? fexp(v[e] - mx) : 0.f;
      sum += ex[j][e];
    }
  }
  sum = blk_sum(sum, sh);
  const float inv = 1.f / sum;
  const float el = expf(fminf(fmaxf(a.l[b], 0.f), a.clip_library));
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int g = (threadIdx.x + 256 * j) * 4;
    const float t[4] = {r1[j].x, r1[j].y, r1[j].z, r1[j].w};
    const float gt[4] = {r2[j].x, r2[j].y, r2[j].z, r2[j].w};
    const float xs[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
    float rate[4], th[4], gate[4], llk[4], d0[4], d1[4], d2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      rate[e] = el * fminf(fmaxf(ex[j][e] * inv, 1e-7f), 1.f - 1e-7f);
      th[e] = fexp(t[e]);
      gate[e] = gt[e];
    }
    count_elem_vec<LK, 1, 4>(xs, rate, th, gate, llk, d0, d1, d2);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc += (g + e < a.G) ? llk[e] : 0.f;
  }
  acc = blk_sum(acc, sh);
  if (threadIdx.x == 0) a.llk[b] = acc;
}

bool scvi_score_supported(int Gp) { return Gp % 4 == 0 && Gp <= 4096; }

int launch_scvi_score_rows(hipStream_t st, const ScviScoreArgs& a) {
  if (!scvi_score_supported(a.Gp) || a.R <= 0 || a.row_mod <= 0 || !a.raw || !a.l || !a.X || !a.llk || (a.likelihood != SMX_LLK_NBD && a.likelihood != SMX_LLK_ZINBD) ||
      (a.likelihood == SMX_LLK_ZINBD && a.k != 3)) {
    set_error("scvi_score_rows: bad arguments");
    return SMX_ERR_INVALID;
  }
  const dim3 grid((unsigned)a.R), block(256);
  if (a.Gp <= 2048) {
    if (a.likelihood == SMX_LLK_ZINBD) hipLaunchKernelGGL((scvi_score_rows_kernel<2, SMX_LLK_ZINBD>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((scvi_score_rows_kernel<2, SMX_LLK_NBD>), grid, block, 0, st, a);
  } else {
    if (a.likelihood == SMX_LLK_ZINBD) hipLaunchKernelGGL((scvi_score_rows_kernel<4, SMX_LLK_ZINBD>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((scvi_score_rows_kernel<4, SMX_LLK_NBD>), grid, block, 0, st, a);
  }
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// one workgroup per cell, one wave per draw in turn: lanes over the head kernel's partial sums
__global__ __launch_bounds__(512) void iw_stack_kernel(IwStackArgs a) {
  __shared__ float lws[SMX_SCORE_MAX_DRAWS], llks[SMX_SCORE_MAX_DRAWS], red[8];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const long src = a.rows ? a.rows[b] : b;
  const float lg = a.lgx1[src];
  // a wave's draws w, w + 8, ...: the partial sums (and the draws' latent weights) of up to 16 of them requested in ONE batch, then summed --
  // draw by draw (load -> wave_sum -> the weight's load -> store) it was two dependent round trips per draw, 13 draws in a row per wave.  The
  // additions of a draw keep their order (lane c takes chunks c, c + 64, ...; then wave_sum).
  constexpr int DB = 16;
  const int nc = a.n_chunks;
  for (int s0 = w; s0 < a.S; s0 += 8 * DB) {
    float v[DB], lwv[DB];
#pragma unroll
    for (int k = 0; k < DB; ++k) {
      const int s = min(s0 + 8 * k, a.S - 1);
      const long r = (long)s * a.B + b;
      const float x = a.llk_part[r * nc + min(lane, nc - 1)];
      v[k] = lane < nc ? x : 0.f;
      lwv[k] = 0.f;
      if (a.lw) lwv[k] = a.lw[r];
    }
    if (nc > 64) {   // (wide panels: further chunks of a draw, in order)
#pragma unroll
      for (int k = 0; k < DB; ++k) {
        const int s = min(s0 + 8 * k, a.S - 1);
        const long r = (long)s * a.B + b;
        for (int c = lane + 64; c < nc; c += 64) v[k] += a.llk_part[r * nc + c];
      }
    }
#pragma unroll
    for (int k = 0; k < DB; ++k) {
      const int s = s0 + 8 * k;
      if (s < a.S) {   // (wave-uniform)
        const float llk = wave_sum(v[k]) - lg;
        if (lane == 0) { llks[s] = llk; lws[s] = llk + lwv[k]; }
      }
    }
  }
  __syncthreads();
  // log-sum-exp of the chunk's weights and the sum of its likelihoods, in a fixed order
  float mx = -INFINITY;
  for (int s = threadIdx.x; s < a.S; s += 512) mx = fmaxf(mx, lws[s]);
  mx = wave_max(mx);
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int j = 1; j < 8; ++j) mx = fmaxf(mx, red[j]);
  __syncthreads();
  if (threadIdx.x >= 64) return;
  float se = 0.f, sl = 0.f;
  for (int s = lane; s < a.S; s += 64) { se += expf(lws[s] - mx); sl += llks[s]; }
  se = wave_sum(se); sl = wave_sum(sl);
  if (lane != 0) return;
  if (a.first) { a.run_max[b] = mx; a.run_sum[b] = se; if (a.llk_sum) a.llk_sum[b] = sl; }
  else {
    const float om = a.run_max[b], nm = fmaxf(om, mx);
    a.run_sum[b] = a.run_sum[b] * expf(om - nm) + se * expf(mx - nm);
    a.run_max[b] = nm;
    if (a.llk_sum) a.llk_sum[b] += sl;
  }
}

int launch_iw_stack(hipStream_t st, const IwStackArgs& a) {
  if (a.B <= 0 || a.S <= 0 || a.S > SMX_SCORE_MAX_DRAWS || a.n_chunks <= 0 || !a.llk_part || !a.lgx1 || !a.run_max || !a.run_sum) {
    set_error("iw_stack: bad arguments");
    return SMX_ERR_INVALID;
  }
  hipLaunchKernelGGL(iw_stack_kernel, dim3((unsigned)a.B), dim3(512), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

}  // namespace smx
