// smx_scvi.hip -- the scVI output head of a TRAINING step as one row-local launch (SURVEY.md 8 row a-12;
// sisua/models/scvi.py:88-171):
//
//   library latent  (mu_l, s_l) = h_l W_l + b_l  ->  l = mu_l + sigma_l eps,  KL_l vs N(local_mean, sqrt(local_var))
//   head            rho = clip(softmax_G(raw_0)),  rate = exp(clip(l, 0, 1e3)) rho,  theta = exp(raw_1),  gate = raw_2
//   likelihood      NBD / ZINBD log-likelihood of the cell's counts and its gradient wrt (rate, theta, gate)
//   backward        through exp / the softmax (row sums) -> d raw planes;  d l -> d (mu_l, s_l)
//
// Every step of that chain is local to one cell's row of G genes, so one 256-thread workgroup per cell keeps the row
// in registers from the raw head outputs to their gradients: the activated planes, their gradients and the saved
// softmax never exist in memory.  It replaces five launches of the separate form (library-latent product,
// lib_latent_fwd, scvi_head_fwd, count_loss, scvi_head_bwd) and lib_latent_bwd: 6 launches, ~33 us of a 180 us step
// at batch 256.  Evaluation / prediction / scoring keep the separate kernels (they hand the planes back).
#include <stdlib.h>

#include "smx_internal.h"
#include "smx_loss.h"
#include "../../include/sisua_hip.h"

namespace smx {

template <int NT = 256>
__device__ inline float scvi_block_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float lo = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  return NT > 256 ? lo + ((sh[4] + sh[5]) + (sh[6] + sh[7])) : lo;
}
template <int NT = 256>
__device__ inline float scvi_block_max(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float lo = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  return NT > 256 ? fmaxf(lo, fmaxf(fmaxf(sh[4], sh[5]), fmaxf(sh[6], sh[7]))) : lo;
}

// NT threads hold 4 NT NV genes of the row: 256 threads up to 8192 genes (NV = 8: one wave per SIMD, its 512 registers), 512 threads up to
// 20 480 (NV = 10; round 6 -- before it, panels beyond 4096 genes took the separate launches, whose one-workgroup-per-cell sweeps with 4-byte
// accesses made the scVI step at 20 000 genes three times the VAE's).  What is held across the barriers is kept small for that: d raw of the
// dispersion and gate planes leave as soon as they are known, the softmax terms become rho in place.
template <int NV, int LK, int U16, int NT = 256>
__global__ __launch_bounds__(NT) void scvi_head_train_kernel(ScviTrainArgs a) {
  constexpr int K3 = (LK == SMX_LLK_ZINBD) ? 1 : 0;
  __shared__ float sh[NT / 64];
  __shared__ float shl[2];
  const int b = blockIdx.x;
  const long src = a.rows ? a.rows[b] : b;
  const long xsrc = a.x_identity ? b : src;
  const float* raw = a.raw + (long)b * a.ld;
  // ---- every load of the row first: raw planes and the cell's counts; what thread 0 needs for the library latent beside them (the prior's
  // moments, the head's biases, injected noise: read where they were used -- behind the wave's dot products -- each was a round trip of its own on
  // the path every wave then waits for at the row maximum's barrier) -----------------------------------------------------------------
  const float bl0 = a.bl[0], bl1 = a.bl[1];
  const float mp_in = a.library[src * 2], vp_in = a.library[src * 2 + 1];
  float eps_in = 0.f;
  if (a.inj_eps) eps_in = a.inj_eps[(long)b * a.inj_ld];
  float4 r0[NV], r1[NV], r2[NV], xv[NV];
  ushort4 xh[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int g = (threadIdx.x + NT * j) * 4;
    const bool ok = g < a.Gp;
    r0[j] = ok ? *reinterpret_cast<const float4*>(raw + g) : zero4();
    r1[j] = ok ? *reinterpret_cast<const float4*>(raw + a.plane_stride + g) : zero4();
    r2[j] = (ok && K3) ? *reinterpret_cast<const float4*>(raw + 2 * a.plane_stride + g) : zero4();
    if (U16) xh[j] = ok ? *reinterpret_cast<const ushort4*>(reinterpret_cast<const uint16_t*>(a.X) + xsrc * a.ldx + g) : make_ushort4(0, 0, 0, 0);
    else xv[j] = ok ? *reinterpret_cast<const float4*>(a.X + xsrc * a.ldx + g) : zero4();
  }
  // ---- library latent of this cell (wave 0; its loads overlap the row's) ----------------------------------------------
  float mu_l = 0.f, sraw_l = 0.f, sig_l = 1.f, eps_l = 0.f, mp = 0.f, vp = 1.f;
  if (threadIdx.x < 64) {
    float p0 = 0.f, p1 = 0.f;
    // (a lane's elements k = lane, lane + 64, ... four at a time in flight; the multiply-adds keep their order)
    for (int k0 = threadIdx.x; k0 < a.Kl; k0 += 256) {
      float hv[4]; float2 wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = min(k0 + 64 * u, a.Kl - 1);
        hv[u] = a.hl[(long)b * a.ldh + k];
        wv[u] = *reinterpret_cast<const float2*>(a.Wl + (long)k * a.ldwl);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (k0 + 64 * u < a.Kl) { p0 = fmaf(hv[u], wv[u].x, p0); p1 = fmaf(hv[u], wv[u].y, p1); }
    }
    p0 = wave_sum(p0); p1 = wave_sum(p1);
    if (threadIdx.x == 0) {
      mu_l = p0 + bl0; sraw_l = p1 + bl1;
      sig_l = softplusf(sraw_l + SMX_SOFTPLUS_INV_1);
      if (a.inj_eps) eps_l = eps_in;
      else eps_l = normal4(philox_block(a.nk, a.cell_base + (uint32_t)src, 0u)).x;
      mp = mp_in; vp = vp_in;
      const float sp = sqrtf(vp);
      const float l = mu_l + sig_l * eps_l;
      a.l[b] = l; a.sig[b] = sig_l; a.eps[b] = eps_l;
      a.kl[b] = logf(sp / sig_l) + (sig_l * sig_l + (mu_l - mp) * (mu_l - mp)) / (2.f * vp) - 0.5f;
      a.latl[(long)b * a.ldl] = mu_l; a.latl[(long)b * a.ldl + 1] = sraw_l;
      shl[0] = l;
    }
  }
  if (U16) {
#pragma unroll
    for (int j = 0; j < NV; ++j) xv[j] = make_float4((float)xh[j].x, (float)xh[j].y, (float)xh[j].z, (float)xh[j].w);
  }
  // ---- softmax over the genes (row max, row sum) ----------------------------------------------------------------------
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int g = (threadIdx.x + NT * j) * 4;
    const float v[4] = {r0[j].x, r0[j].y, r0[j].z, r0[j].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) if (g + e < a.G) mx = fmaxf(mx, v[e]);
  }
  mx = scvi_block_max<NT>(mx, sh);   // (its barriers also publish shl[0])
  float ex[NV][4];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int g = (threadIdx.x + NT * j) * 4;
    const float v[4] = {r0[j].x, r0[j].y, r0[j].z, r0[j].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ex[j][e] = (g + e < a.G) ? fexp(v[e] - mx) : 0.f;
      sum += ex[j][e];
    }
  }
  sum = scvi_block_sum<NT>(sum, sh);
  const float inv = 1.f / sum;
  const float lraw = shl[0];
  const float el = expf(fminf(fmaxf(lraw, 0.f), a.clip_library));
  // ---- parameters, likelihood and its gradient; the two row sums of the backward pass ---------------------------------
  float d0s[NV][4];
  float (&rho)[NV][4] = ex;   // (in place)
  float* dr = a.draw + (long)b * a.ld;
  float llk_sum = 0.f, s = 0.f, dlh = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int g = (threadIdx.x + NT * j) * 4;
    const float t[4] = {r1[j].x, r1[j].y, r1[j].z, r1[j].w};
    const float gt[4] = {r2[j].x, r2[j].y, r2[j].z, r2[j].w};
    const float xs[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
    float rate[4], th[4], gate[4], llk[4], d0[4], d1[4], d2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool live = g + e < a.G;
      rho[j][e] = live ? ex[j][e] * inv : 0.f;
      rate[e] = live ? el * fminf(fmaxf(rho[j][e], 1e-7f), 1.f - 1e-7f) : 1.f;   // (dead lanes: any finite parameters)
      th[e] = live ? fexp(t[e]) : 1.f;
      gate[e] = live ? gt[e] : 0.f;
    }
    count_elem_vec<LK, 1, 4>(xs, rate, th, gate, llk, d0, d1, d2);
    float o1[4], o2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool live = g + e < a.G;
      const float dd = live ? d0[e] * a.grad_scale : 0.f;   // d loss / d rate
      d0s[j][e] = dd;
      o1[e] = live ? d1[e] * a.grad_scale * th[e] : 0.f;   // d loss / d raw_1 (theta = exp(raw_1))
      o2[e] = live ? d2[e] * a.grad_scale : 0.f;
      if (live) {
        llk_sum += llk[e];
        const float inside = (rho[j][e] > 1e-7f && rho[j][e] < 1.f - 1e-7f) ? 1.f : 0.f;
        s += dd * el * inside * rho[j][e];
        dlh += dd * rate[e];
      }
    }
    if (g < a.Gp) {   // (these two planes' gradients are final: out now, not held across the row sums' barriers)
      *reinterpret_cast<float4*>(dr + a.plane_stride + g) = make_float4(o1[0], o1[1], o1[2], o1[3]);
      if (K3) *reinterpret_cast<float4*>(dr + 2 * a.plane_stride + g) = make_float4(o2[0], o2[1], o2[2], o2[3]);
    }
  }
  llk_sum = scvi_block_sum<NT>(llk_sum, sh);
  s = scvi_block_sum<NT>(s, sh);
  dlh = scvi_block_sum<NT>(dlh, sh);
  // ---- gradient wrt the raw head outputs ------------------------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int g = (threadIdx.x + NT * j) * 4;
    if (g >= a.Gp) continue;
    float o0[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool live = g + e < a.G;
      const float inside = (rho[j][e] > 1e-7f && rho[j][e] < 1.f - 1e-7f) ? 1.f : 0.f;
      o0[e] = live ? rho[j][e] * (d0s[j][e] * el * inside - s) : 0.f;
    }
    *reinterpret_cast<float4*>(dr + g) = make_float4(o0[0], o0[1], o0[2], o0[3]);
  }
  // ---- the cell's scalars: likelihood partial, d l and the library latent's backward ----------------------------------
  if (threadIdx.x == 0) {
    a.llk_part[b] = llk_sum;
    const float dl = (lraw > 0.f && lraw < a.clip_library) ? dlh : 0.f;
    a.dl[b] = dl;
    float* o = a.dlatl + (long)b * a.ldl;
    o[0] = dl + a.kl_scale * (mu_l - mp) / vp;
    o[1] = (dl * eps_l + a.kl_scale * (sig_l / vp - 1.f / sig_l)) * sigmoidf(sraw_l + SMX_SOFTPLUS_INV_1);
  }
  if (threadIdx.x >= 2 && (int)threadIdx.x < a.ldl) a.dlatl[(long)b * a.ldl + threadIdx.x] = 0.f;
}

bool scvi_head_train_supported(const ScviTrainArgs& a) {
  const bool off = false;   // (the launch form is a model flag: smx_set_flag("scvi_fused"))
  return !off && (a.likelihood == SMX_LLK_NBD || a.likelihood == SMX_LLK_ZINBD) && (a.ld % 4) == 0 && (a.plane_stride % 4) == 0 &&
         (a.Gp % 4) == 0 && a.Gp <= 20480 && (a.ldx % 4) == 0 && (a.ldwl % 2) == 0 && a.ldl >= 2 && a.ldl <= 256 && a.Kl > 0;
}

template <int NV, int LK, int NT = 256>
static void launch_sht(hipStream_t st, const ScviTrainArgs& a) {
  if (a.x_u16) hipLaunchKernelGGL((scvi_head_train_kernel<NV, LK, 1, NT>), dim3(a.B), dim3(NT), 0, st, a);
  else hipLaunchKernelGGL((scvi_head_train_kernel<NV, LK, 0, NT>), dim3(a.B), dim3(NT), 0, st, a);
}

int launch_scvi_head_train(hipStream_t st, const ScviTrainArgs& a) {
  if (!scvi_head_train_supported(a) || !a.raw || !a.X || !a.draw || !a.llk_part || !a.hl || !a.Wl || !a.bl || !a.library ||
      !a.latl || !a.l || !a.sig || !a.eps || !a.kl || !a.dlatl || !a.dl) {
    set_error("scvi_head_train: bad arguments");
    return SMX_ERR_INVALID;
  }
  const bool zi = a.likelihood == SMX_LLK_ZINBD;
  if (a.Gp <= 2048) { if (zi) launch_sht<2, SMX_LLK_ZINBD>(st, a); else launch_sht<2, SMX_LLK_NBD>(st, a); }
  else if (a.Gp <= 4096) { if (zi) launch_sht<4, SMX_LLK_ZINBD>(st, a); else launch_sht<4, SMX_LLK_NBD>(st, a); }
  else if (a.Gp <= 8192) { if (zi) launch_sht<8, SMX_LLK_ZINBD>(st, a); else launch_sht<8, SMX_LLK_NBD>(st, a); }   // (one wave per SIMD: 512 registers hold 8192 genes of a row)
  else if (a.Gp <= 16384) { if (zi) launch_sht<8, SMX_LLK_ZINBD, 512>(st, a); else launch_sht<8, SMX_LLK_NBD, 512>(st, a); }
  else { if (zi) launch_sht<10, SMX_LLK_ZINBD, 512>(st, a); else launch_sht<10, SMX_LLK_NBD, 512>(st, a); }
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// ---- scvi.py:55-56,66-86: a plane of the gene output WITHOUT a Dense head (dispersion / inflation = 'share') ----------------
// The plane's raw value is one trainable per-gene vector v[g] shared by every cell: forward = the vector copied into every row of the
// raw plane (what the head's product + bias leaves for a 'full' plane, so every consumer downstream is unchanged), backward = the
// column sum of the plane's d raw in row order (deterministic); no kernel, no d d contribution.
__global__ __launch_bounds__(256) void plane_fill_kernel(float* dst, long ld, const float* v, int B, int Np, int single) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= Np) return;
  const float x = v[single ? 0 : g];   // ('single': one scalar for every gene)
  for (int b = blockIdx.y; b < B; b += gridDim.y) dst[(long)b * ld + g] = x;
}
__global__ __launch_bounds__(256) void plane_colsum_kernel(const float* src, long ld, float* dst, int B, int Np) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= Np) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += src[(long)b * ld + g];
  dst[g] = s;
}
// 'single': the one scalar's gradient = the sum of the whole plane's d raw over the G live genes -- one workgroup, fixed order
// (thread t: genes t, t + 256, ... each summed over the rows; then the threads' sums in LDS order): deterministic
__global__ __launch_bounds__(256) void plane_sum_kernel(const float* src, long ld, float* dst, int B, int G) {
  __shared__ float sh[256];
  float s = 0.f;
  for (int g = threadIdx.x; g < G; g += 256) {
    float c = 0.f;
    for (int b = 0; b < B; ++b) c += src[(long)b * ld + g];
    s += c;
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { float t = 0.f; for (int i = 0; i < 256; ++i) t += sh[i]; dst[0] = t; }
}
int launch_plane_fill(hipStream_t st, float* dst, long ld, const float* v, int B, int Np, int single) {
  hipLaunchKernelGGL(plane_fill_kernel, dim3((Np + 255) / 256, std::min(B, 16)), dim3(256), 0, st, dst, ld, v, B, Np, single);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}
int launch_plane_colsum(hipStream_t st, const float* src, long ld, float* dst, int B, int Np, int single_G) {
  if (single_G > 0) hipLaunchKernelGGL(plane_sum_kernel, dim3(1), dim3(256), 0, st, src, ld, dst, B, single_G);
  else hipLaunchKernelGGL(plane_colsum_kernel, dim3((Np + 255) / 256), dim3(256), 0, st, src, ld, dst, B, Np);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

}  // namespace smx
