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
#include "smx_internal.h"
#include "smx_device.h"

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
  for (int d = lane; d < a.Dp; d += 64) {
    float z = 0.f;
    if (d < a.D) {
      const float mu = a.lat[(long)b * a.ld + d];
      const float sig = softplusf(a.lat[(long)b * a.ld + a.Dp + d] + SMX_SOFTPLUS_INV_1);
      const float4 n = normal4(philox_block(nk, cell, (uint32_t)(d >> 2)));
      const float eps = (d & 3) == 0 ? n.x : (d & 3) == 1 ? n.y : (d & 3) == 2 ? n.z : n.w;
      z = mu + sig * eps;
      lw += -0.5f * z * z + 0.5f * eps * eps + logf(sig);   // log N(z; 0, I) - log N(z; mu, sigma), constants cancel
    }
    a.z[r * a.Dp + d] = z;
  }
  lw = wave_sum(lw);
  if (lane == 0) a.lw[r] = lw;
}

int launch_score_draws(hipStream_t st, const ScoreDrawArgs& a) {
  if (a.S <= 0 || a.B <= 0 || a.Dp <= 0 || !a.lat || !a.z || !a.lw) { set_error("score_draws: bad arguments"); return SMX_ERR_INVALID; }
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

int launch_score_bn_act(hipStream_t st, const ScoreBnArgs& a) {
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

// one workgroup per cell, one wave per draw in turn: lanes over the head kernel's partial sums
__global__ __launch_bounds__(512) void iw_stack_kernel(IwStackArgs a) {
  __shared__ float lws[SMX_SCORE_MAX_DRAWS], llks[SMX_SCORE_MAX_DRAWS], red[8];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const long src = a.rows ? a.rows[b] : b;
  const float lg = a.lgx1[src];
  for (int s = w; s < a.S; s += 8) {
    const long r = (long)s * a.B + b;
    float llk = 0.f;
    for (int c = lane; c < a.n_chunks; c += 64) llk += a.llk_part[r * a.n_chunks + c];
    llk = wave_sum(llk) - lg;
    if (lane == 0) { llks[s] = llk; lws[s] = llk + a.lw[r]; }
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
  if (a.first) { a.run_max[b] = mx; a.run_sum[b] = se; a.llk_sum[b] = sl; }
  else {
    const float om = a.run_max[b], nm = fmaxf(om, mx);
    a.run_sum[b] = a.run_sum[b] * expf(om - nm) + se * expf(mx - nm);
    a.run_max[b] = nm;
    a.llk_sum[b] += sl;
  }
}

int launch_iw_stack(hipStream_t st, const IwStackArgs& a) {
  if (a.B <= 0 || a.S <= 0 || a.S > SMX_SCORE_MAX_DRAWS || a.n_chunks <= 0 || !a.llk_part || !a.lw || !a.lgx1 || !a.run_max || !a.run_sum || !a.llk_sum) {
    set_error("iw_stack: bad arguments");
    return SMX_ERR_INVALID;
  }
  hipLaunchKernelGGL(iw_stack_kernel, dim3((unsigned)a.B), dim3(512), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

}  // namespace smx
