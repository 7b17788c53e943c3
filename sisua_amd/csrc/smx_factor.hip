// smx_factor.hip -- the two small kernels of the FactorVAE discriminator (sisua/models/fvae.py:9-18: FVAE / SemiFVAE
// are thin subclasses of odin's factorVAE / SemifactorVAE; the algorithm is Kim & Mnih 2018, Algorithm 2).  The
// discriminator's Dense layers run on the generic products (smx_gemm.hip) and the bias + leaky-ReLU launches
// (smx_kernels.hip); what is specific is
//   permute_dims_kernel   z_perm: every latent dimension permuted over the minibatch independently.  The permutation of
//                         dimension d is the rank of the cell's Philox uniform u[cell][d] within column d (ties by row),
//                         so it depends on (seed, step, cell ids) only and the oracle draws the same one;
//   disc_head_kernel      logits -> d = logsumexp (one logit: d itself), the total-correlation estimate d(z), the
//                         discriminator's loss 1/2 [softplus(-d(z)) + softplus(d(z_perm))], SemiFVAE's masked
//                         cross-entropy (summed over its label variables, each over its own logits), and the two upstream gradients (VAE objective on the rows of z, discriminator
//                         objective on all 2B rows).
// Both are HBM-trivial ([B][D] and [2B][<= 32] floats); they exist to keep the step on the device.
#include "smx_internal.h"
#include "smx_device.h"
#include "../../include/sisua_hip.h"

namespace smx {

__global__ __launch_bounds__(256) void permute_dims_kernel(PermuteArgs a) {
  extern __shared__ float us[];   // [B] uniforms of this latent dimension
  const int d = blockIdx.x;
  for (int b = threadIdx.x; b < a.B; b += 256) {
    float u;
    if (a.inj_u) u = a.inj_u[(long)b * a.inj_ld + d];
    else {
      const uint32_t cell = a.cell_base + (uint32_t)(a.rows ? a.rows[b] : b);
      const U4 w = philox_block(a.nk, cell, (uint32_t)(d >> 2));
      const int q = d & 3;
      u = u24(q == 0 ? w.x : q == 1 ? w.y : q == 2 ? w.z : w.w);
    }
    us[b] = u;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < a.B; b += 256) {
    const float u = us[b];
    int rank = 0;
    for (int o = 0; o < a.B; ++o) {
      const float v = us[o];
      rank += (v < u || (v == u && o < b)) ? 1 : 0;
    }
    const float zv = a.z[(long)b * a.ldz + d];
    a.zz[(long)b * a.ld + d] = zv;
    a.zz[(long)(a.B + rank) * a.ld + d] = zv;
  }
}

int launch_permute_dims(hipStream_t st, const PermuteArgs& a) {
  if (a.B <= 0 || a.D <= 0 || a.D > a.ld || (size_t)a.B * sizeof(float) > 64 * 1024) { set_error("permute_dims: bad shapes"); return SMX_ERR_INVALID; }
  hipLaunchKernelGGL(permute_dims_kernel, dim3(a.D), dim3(256), (size_t)a.B * sizeof(float), st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

__device__ inline float half_wave_max(float v) {
#define SMX_DPP_MAX(ctrl) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, 0xF, 0xF, false)))
  SMX_DPP_MAX(0xB1);
  SMX_DPP_MAX(0x4E);
  SMX_DPP_MAX(0x141);
  SMX_DPP_MAX(0x140);
#undef SMX_DPP_MAX
  return fmaxf(v, __shfl_xor(v, 16, 64));
}

// one LANE per (row of the stacked batch [z ; z_perm], logit): the 32 lanes of a wave half are one row -- its slabs, the label
// row and both upstream gradients move as coalesced 128-byte rows, and the row's log-sum-exp is a half-wave reduction.  (One
// thread per row walked its 32 columns and the split-K slabs of each with strided 4-byte accesses: 12.7 us for a single
// workgroup of latency.)
__global__ __launch_bounds__(256) void disc_head_kernel(DiscHeadArgs a) {
  const int r = blockIdx.x * 8 + (threadIdx.x >> 5), j = threadIdx.x & 31;
  if (r >= 2 * a.B) return;   // (whole half-waves leave together)
  const bool live = j < a.n_out;
  float l = 0.f;
  if (live) {
    l = a.bias[j];
    for (int s = 0; s < a.n_slabs; ++s) l += a.logits[(long)s * a.slab_stride + (long)r * a.ld + j];
  }
  const float mx = half_wave_max(live ? l : -3.0e38f);
  const float se = half_wave_sum(live ? expf(l - mx) : 0.f);
  const float d = mx + logf(se);
  const bool real = r < a.B;
  if (j == 0) {
    if (real) a.tc_cell[r] = d;
    a.dl_cell[r] = 0.5f * softplusf(real ? -d : d);
  }
  // the supervised term: every label variable's masked cross-entropy under the softmax of ITS logits (one variable: that softmax is the
  // one of the TC logit, d its log-sum-exp -- the same arithmetic as before there were several)
  float mk = 0.f, ysum = 0.f, yj = 0.f, gl = d;   // (gl: the log-sum-exp of this lane's variable)
  const bool sup = real && a.n_groups > 0;
  if (sup) {
    const long row = a.rows ? a.rows[r] : r;
    mk = (a.mask && a.mask[row]) ? 1.f : 0.f;
    float ce = 0.f;
    for (int g = 0; g < a.n_groups; ++g) {   // (launch-uniform; every reduction by the whole half-wave)
      const bool mine = j >= a.gstart[g] && j < a.gstart[g + 1];
      const float yv = mine ? a.Y[g][row * a.ldy[g] + (j - a.gstart[g])] : 0.f;
      float lse = d;
      if (a.n_groups > 1) {
        const float gm = half_wave_max(mine ? l : -3.0e38f);
        lse = gm + logf(half_wave_sum(mine ? expf(l - gm) : 0.f));
      }
      const float ys = half_wave_sum(yv);
      ce -= half_wave_sum(mine ? yv * (l - lse) : 0.f);
      if (mine) { yj = yv; ysum = ys; gl = lse; }
    }
    if (j == 0) a.llk_y[r] = -mk * ce;
  } else if (real && a.llk_y && j == 0) a.llk_y[r] = 0.f;
  if (!a.backward) return;
  // d J_d / d d(row): -1/2 sigmoid(-d) on the rows of z, +1/2 sigmoid(d) on the permuted rows (means over the global batch)
  const float gd = (real ? -0.5f / (1.f + expf(d)) : 0.5f / (1.f + expf(-d))) * a.inv_gb;
  float ut = 0.f, ud = 0.f;
  if (live) {
    const float sm = expf(l - d);                                        // softmax = d logsumexp / d logit (1 for one logit)
    const float su = sup ? a.alpha * a.inv_gb * mk * ((a.n_groups > 1 ? expf(l - gl) : sm) * ysum - yj) : 0.f;
    ut = a.gamma * a.inv_gb * sm + su;
    ud = gd * sm + su;
  }
  if (real) a.u_tc[(long)r * 32 + j] = ut;
  a.u_d[(long)r * 32 + j] = ud;
}

int launch_disc_head(hipStream_t st, const DiscHeadArgs& a) {
  if (a.B <= 0 || a.n_out < 1 || a.n_out > 32 || a.ld < a.n_out || a.n_groups < 0 || a.n_groups > SMX_DISC_MAX_GROUPS) { set_error("disc_head: bad shapes"); return SMX_ERR_INVALID; }
  for (int g = 0; g < a.n_groups; ++g)
    if (!a.Y[g] || a.gstart[g + 1] <= a.gstart[g] || a.gstart[0] != 0 || a.gstart[a.n_groups] != a.n_out || a.ldy[g] < a.gstart[g + 1] - a.gstart[g] || !a.llk_y) {
      set_error("disc_head: the label variables' logit ranges must tile the logit layer");
      return SMX_ERR_INVALID;
    }
  hipLaunchKernelGGL(disc_head_kernel, dim3((2 * a.B + 7) / 8), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

}  // namespace smx
