// smx_loss.h -- elementwise NB / ZINB / NBD / ZINBD log-likelihood and its gradients wrt the parameter
// planes; shared by the standalone loss kernel (smx_kernels.hip) and the
// fused output-head kernel (smx_head.hip).
#ifndef SMX_LOSS_H_
#define SMX_LOSS_H_
#include "smx_device.h"
#include "../../include/sisua_hip.h"

namespace smx {

// ===========================================================================
// count likelihood, elementwise
// ===========================================================================
template <int LK, int DIRECT>
__device__ inline void count_elem(float x, float p0, float p1, float p2, float& llk, float& d0, float& d1,
                                  float& d2) {
  float ell;
  if (LK == SMX_LLK_NB || LK == SMX_LLK_ZINB) {
    const float r = fexp(p0);
    const SpSg s = softplus_sigmoid(p1);     // log_sigmoid(l) = l - sp, log_sigmoid(-l) = -sp
    const LgDg t = lgamma_digamma_diff(x, r);
    ell = t.lg + x * (p1 - s.sp) - r * s.sp;
    d0 = r * (t.dg - s.sp);
    d1 = x - (x + r) * s.sg;
  } else {
    float mu, th, g0 = 1.f, g1 = 1.f;
    if (DIRECT) { mu = p0; th = p1; }
    else {
      const SpSg s0 = softplus_sigmoid(p0), s1 = softplus_sigmoid(p1 + SMX_SOFTPLUS_INV_1);
      mu = s0.sp; th = s1.sp; g0 = s0.sg; g1 = s1.sg;
    }
    const float e = 1e-8f;
    // log(th+e) - log(th+mu+e) = -log1p(mu / (th+e)) and th/(th+e) - th/(th+mu+e) = th mu / ((th+e)(th+mu+e)):
    // written as differences of O(1) terms they lose everything for a large dispersion (th ~ 1e3 next to a
    // small mean: the Poisson limit), and the error is then multiplied by th again through exp() in the scvi head
    const float inv_te = frcp(th + e), inv = frcp(th + mu + e);
    const float l1p = log1p_small(mu * inv_te);
    const float lt = flog(th + mu + e);
    const LgDg t = lgamma_digamma_diff(x, th);
    ell = -th * l1p + x * (flog(mu + e) - lt) + t.lg;
    d0 = (-th * inv + x * frcp(mu + e) - x * inv) * g0;
    d1 = (-l1p + th * mu * inv_te * inv - x * inv + t.dg) * g1;
  }
  if (LK == SMX_LLK_ZINB || LK == SMX_LLK_ZINBD) {
    const SpSg sg = softplus_sigmoid(p2);
    if (x == 0.f) {
      // lse = logaddexp(g, ell); w = d lse / d ell = sigmoid(ell - g): one exponential for both
      const float dlt = ell - p2;
      const float e3 = fexp(-fabsf(dlt));
      const float inv3 = frcp(1.0f + e3);
      const float lse = fmaxf(p2, ell) + log1p_small(e3);
      const float w = dlt >= 0.f ? inv3 : e3 * inv3;
      llk = lse - sg.sp;
      d0 *= w; d1 *= w;
      d2 = (1.f - w) - sg.sg;
    } else {
      llk = ell - sg.sp;
      d2 = -sg.sg;
    }
  } else {
    llk = ell;
    d2 = 0.f;
  }
}

// ===========================================================================
// The same arithmetic for N independent elements of one lane, written stage by stage as straight-line code
// (selects instead of branches; the rare Stirling path of lgamma / digamma behind ONE wave-uniform test) so that
// the scheduler can interleave the N dependent chains: with one or two waves per SIMD a lone chain issues one
// dependent instruction per ~13 cycles, and an element-after-element epilogue ran at a third of the rate.
// Results are bit-identical to count_elem (same operations in the same order per element).
// ===========================================================================
template <int N>
__device__ inline void lgamma_digamma_diff_vec(const float (&x)[N], const float (&r_in)[N], float (&lg)[N], float (&dg)[N]) {
  float r[N];
  bool rare = false;
#pragma unroll
  for (int e = 0; e < N; ++e) {
    r[e] = fminf(fmaxf(r_in[e], 1e-30f), 1e30f);
    float P = 1.f, dP = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float t = r[e] + (float)i;
      const bool on = (float)i < x[e];
      dP = on ? fmaf(dP, t, P) : dP;
      P = on ? P * t : P;
    }
    lg[e] = flog(P);
    dg[e] = dP * frcp(P);
    rare |= !((x[e] <= 8.0f) && (x[e] == floorf(x[e])) && (r[e] < 1e4f));
  }
  if (!__any(rare)) return;   // wave-uniform: 99 % of single-cell counts are integers in 0..8
#pragma unroll
  for (int e = 0; e < N; ++e) {
    const bool small = (x[e] <= 8.0f) && (x[e] == floorf(x[e])) && (r[e] < 1e4f);
    const float nf = fmaxf(ceilf(4.0f - r[e]), 0.f);
    float num = 1.f, den = 1.f, dg_shift = 0.f;   // kept apart: see lgamma_digamma_diff
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = r[e] + (float)i, b = x[e] + a;
      const bool on = (float)i < nf;
      num = on ? num * a : num;
      den = on ? den * b : den;
      dg_shift = on ? dg_shift + x[e] * frcp(b) * frcp(a) : dg_shift;
    }
    const float lg_shift = flog(num) - flog(den);
    const float rs = r[e] + nf;
    const float zr = x[e] + rs;
    const float l1p = log1p_small(x[e] * frcp(rs));
    const float lgr = x[e] * flog(zr) + (rs - 0.5f) * l1p - x[e] + (stirling_corr(zr) - stirling_corr(rs)) + lg_shift;
    const float dgr = l1p + (digamma_corr(zr) - digamma_corr(rs)) + dg_shift;
    lg[e] = small ? lg[e] : lgr;
    dg[e] = small ? dg[e] : dgr;
  }
}

// ===========================================================================
// lgamma(x + r) - lgamma(r) and its r-derivative for the NON-ZERO counts of a wave, compacted.
// 88-93 % of the elements of a single-cell minibatch are x == 0, for which both results are exactly 0 whatever r (the recurrence
// leaves P = 1: v_log(1) = 0, dP = 0; the Stirling form differences two equal terms) -- yet the straight-line form above runs the
// 8-step recurrence for every element and the Stirling code for every element of every wave that holds ONE count above 8.  Here the
// lanes push their non-zero (x, r) pairs into the wave's own stretch of an LDS queue (ballot + mbcnt), the wave drains it 64 entries
// at a time through the scalar code (which branches per entry) and every lane reads its entries' results back.  No workgroup
// barrier: a wave's LDS accesses complete in order.  Per element the arithmetic is lgamma_digamma_diff's, operation for operation:
// bit-identical to the straight-line forms (profiles/r04_likelihood_forms.txt: the same final loss to the last bit over 200 steps at
// both widths).  What it buys is small, and measured: the whole lgamma / digamma part is 4.7 of the fused head's 35 us at 128 x 20 000
// (1.7 of the standalone kernel's 15): queue form -1.0 us (C5 width), -0.4 us (C2).  A workgroup-wide queue (one LDS atomic per wave,
// two barriers, the waves sharing the drain rounds) measured the same as the per-wave form and was not kept.
//   q : LDS, 64 N float2 per wave of the workgroup (the wave's stretch starts at wave * 64 * N)
// ===========================================================================
// (Q: anything indexed like a float2 array that is THIS wave's own stretch of 64 N entries -- a pointer, or an accessor over LDS that
// is laid out otherwise: smx_headfused.hip keeps the queue inside the rows of an operand image it owns)
template <int N, class Q>
__device__ inline void lgamma_digamma_diff_queue_at(const float (&x)[N], const float (&r_in)[N], float (&lg)[N], float (&dg)[N], Q q) {
  const int lane = threadIdx.x & 63;
  int pos[N];
  int mine = 0;
#pragma unroll
  for (int e = 0; e < N; ++e) {
    const unsigned long long m = __ballot(x[e] != 0.f);
    pos[e] = mine + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    mine += __popcll(m);
    lg[e] = 0.f; dg[e] = 0.f;
    if (x[e] != 0.f) q[pos[e]] = make_float2(x[e], r_in[e]);
  }
  __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();   // (the stretch is this wave's own)
  for (int k0 = 0; k0 < mine; k0 += 64) {
    const int k = k0 + lane;
    if (k < mine) {
      const float2 v = q[k];
      const LgDg o = lgamma_digamma_diff(v.x, v.y);
      q[k] = make_float2(o.lg, o.dg);
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int e = 0; e < N; ++e)
    if (x[e] != 0.f) { const float2 v = q[pos[e]]; lg[e] = v.x; dg[e] = v.y; }
}
template <int N>
__device__ inline void lgamma_digamma_diff_queue(const float (&x)[N], const float (&r_in)[N], float (&lg)[N], float (&dg)[N], float2* q) {
  lgamma_digamma_diff_queue_at<N>(x, r_in, lg, dg, q + (threadIdx.x >> 6) * 64 * N);
}
template <int N, class Q>
__device__ inline void lgamma_digamma_diff_queue(const float (&x)[N], const float (&r_in)[N], float (&lg)[N], float (&dg)[N], Q q) {
  lgamma_digamma_diff_queue_at<N>(x, r_in, lg, dg, q);
}

// q != nullptr: lgamma / digamma through the per-wave queue form above
template <int LK, int DIRECT, int N, class Q = float2*>
__device__ inline void count_elem_vec(const float (&x)[N], const float (&p0)[N], const float (&p1)[N], const float (&p2)[N],
                                      float (&llk)[N], float (&d0)[N], float (&d1)[N], float (&d2)[N],
                                      Q q = nullptr) {
  float ell[N];
  if (LK == SMX_LLK_NB || LK == SMX_LLK_ZINB) {
    float r[N], lg[N], dg[N];
    SpSg s[N];
#pragma unroll
    for (int e = 0; e < N; ++e) { r[e] = fexp(p0[e]); s[e] = softplus_sigmoid(p1[e]); }
    if (q) lgamma_digamma_diff_queue<N>(x, r, lg, dg, q);
    else lgamma_digamma_diff_vec<N>(x, r, lg, dg);
#pragma unroll
    for (int e = 0; e < N; ++e) {
      ell[e] = lg[e] + x[e] * (p1[e] - s[e].sp) - r[e] * s[e].sp;
      d0[e] = r[e] * (dg[e] - s[e].sp);
      d1[e] = x[e] - (x[e] + r[e]) * s[e].sg;
    }
  } else {
    float mu[N], th[N], g0[N], g1[N], lg[N], dg[N];
#pragma unroll
    for (int e = 0; e < N; ++e) {
      if (DIRECT) { mu[e] = p0[e]; th[e] = p1[e]; g0[e] = 1.f; g1[e] = 1.f; }
      else {
        const SpSg s0 = softplus_sigmoid(p0[e]), s1 = softplus_sigmoid(p1[e] + SMX_SOFTPLUS_INV_1);
        mu[e] = s0.sp; th[e] = s1.sp; g0[e] = s0.sg; g1[e] = s1.sg;
      }
    }
    if (q) lgamma_digamma_diff_queue<N>(x, th, lg, dg, q);
    else lgamma_digamma_diff_vec<N>(x, th, lg, dg);
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const float eps = 1e-8f;
      const float inv_te = frcp(th[e] + eps), inv = frcp(th[e] + mu[e] + eps);
      const float l1p = log1p_small(mu[e] * inv_te);
      const float lt = flog(th[e] + mu[e] + eps);
      ell[e] = -th[e] * l1p + x[e] * (flog(mu[e] + eps) - lt) + lg[e];
      d0[e] = (-th[e] * inv + x[e] * frcp(mu[e] + eps) - x[e] * inv) * g0[e];
      d1[e] = (-l1p + th[e] * mu[e] * inv_te * inv - x[e] * inv + dg[e]) * g1[e];
    }
  }
  if (LK == SMX_LLK_ZINB || LK == SMX_LLK_ZINBD) {
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const SpSg sg = softplus_sigmoid(p2[e]);
      const float dlt = ell[e] - p2[e];
      const float e3 = fexp(-fabsf(dlt));
      const float inv3 = frcp(1.0f + e3);
      const float lse = fmaxf(p2[e], ell[e]) + log1p_small(e3);
      const float w = dlt >= 0.f ? inv3 : e3 * inv3;
      const bool zero = x[e] == 0.f;
      llk[e] = (zero ? lse : ell[e]) - sg.sp;
      d0[e] = zero ? d0[e] * w : d0[e];
      d1[e] = zero ? d1[e] * w : d1[e];
      d2[e] = (zero ? (1.f - w) : 0.f) - sg.sg;
    }
  } else {
#pragma unroll
    for (int e = 0; e < N; ++e) { llk[e] = ell[e]; d2[e] = 0.f; }
  }
}

}  // namespace smx
#endif  // SMX_LOSS_H_
