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

}  // namespace smx
#endif  // SMX_LOSS_H_
