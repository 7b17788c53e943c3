// smx_mid.hip -- the "middle" of the network as ONE workgroup per kernel.
//
// Between the two wide products of a step (genes -> hidden, hidden -> k*genes) the
// network is a chain of tiny dependent operators on [B x <=128] activations:
// (further encoder layers) -> latent head -> reparameterised sample + KL -> decoder
// layers, and its mirror image in the backward pass.  As separate launches each of
// them costs ~5 us of launch + dependent-HBM-round-trip latency for ~0.1 us of
// work.  Here the whole chain runs inside one 1024-thread workgroup: activations
// stay in LDS ([rows][width+1] so MFMA A-operand column reads are conflict-free),
// weights (<= 64 KB each, L2 resident) stream straight from global memory into the
// MFMA B operand, and only what the backward pass needs is written back to HBM.
//
// Eligibility (checked on the host, otherwise the per-operator path runs):
// B <= 128, every hidden width and 2*latent padded <= 128.
#include "smx_internal.h"
#include "../../include/sisua_hip.h"

namespace smx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MID_THREADS = 1024;
constexpr int MID_WAVES = MID_THREADS / 64;
constexpr int MID_MAXW = 128;            // widest activation held in LDS
constexpr int MID_LD = MID_MAXW + 1;     // odd stride: lanes i -> rows i hit distinct banks
constexpr int MID_ROWS = 128;

// C[M x N] = A[M x K] * B[K x N] with all waves of the workgroup; tiles of 32x32 dealt
// round-robin to waves.  k index of MFMA step s for lane half h is h*K/2 + s (any bijection
// works as long as A and B agree).  The wave's whole-K B fragment (<= 64 values per lane) is
// fetched from global memory up front so every load is in flight before the first MFMA:
//   B_NMAJOR = 0: B stored [K][ldb] (weights as they are): coalesced 4-byte loads
//   B_NMAJOR = 1: B stored [N][ldb] (W^T products): each lane walks its own row in 16-byte loads
//   AF(row, k) -> float (LDS), EPI(row, col, value)
template <int KH, int B_NMAJOR, class AF, class EPI>
__device__ inline void block_gemm_k(int M, int N, const float* Bp, int ldb, AF af, EPI epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int ntm = (M + 31) / 32, ntn = N / 32;
  const int kb = lh * KH;
  for (int t = wave; t < ntm * ntn; t += MID_WAVES) {
    const int m0 = (t / ntn) * 32, n0 = (t % ntn) * 32;
    float bfrag[KH];
    if (B_NMAJOR) {
      const float4* src = reinterpret_cast<const float4*>(Bp + (long)(n0 + li) * ldb + kb);
#pragma unroll
      for (int q = 0; q < KH / 4; ++q) {
        const float4 v = src[q];
        bfrag[4 * q] = v.x; bfrag[4 * q + 1] = v.y; bfrag[4 * q + 2] = v.z; bfrag[4 * q + 3] = v.w;
      }
    } else {
      const float* src = Bp + (long)kb * ldb + n0 + li;
#pragma unroll
      for (int s = 0; s < KH; ++s) bfrag[s] = src[(long)s * ldb];
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < KH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af(m0 + li, kb + s), bfrag[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) epi(m0 + (r & 3) + 8 * (r >> 2) + 4 * lh, n0 + li, acc[r]);
  }
}

template <int B_NMAJOR, class AF, class EPI>
__device__ inline void block_gemm(int M, int N, int K, const float* Bp, int ldb, AF af, EPI epi) {
  switch (K) {   // K is a padded feature width: 32, 64, 96 or 128
    case 32: block_gemm_k<16, B_NMAJOR>(M, N, Bp, ldb, af, epi); break;
    case 64: block_gemm_k<32, B_NMAJOR>(M, N, Bp, ldb, af, epi); break;
    case 96: block_gemm_k<48, B_NMAJOR>(M, N, Bp, ldb, af, epi); break;
    default: block_gemm_k<64, B_NMAJOR>(M, N, Bp, ldb, af, epi); break;
  }
}

__device__ inline NoiseKey mid_key(const MidArgs& a, uint32_t stream) {
  NoiseKey nk;
  nk.k0 = a.k0; nk.k1 = a.k1; nk.step = a.step; nk.stream = (stream & 0xFF) | ((a.sample & 0xFFFFFF) << 8);
  nk.step_ptr = a.step_ptr;
  return nk;
}

// column statistics of act[B][w] (LDS, stride MID_LD) by all threads; red: [MID_THREADS] scratch.
// Returns the column sum of f(row, col) in every thread of that column (threads c, c+w, ...).
template <class F>
__device__ inline float col_sum(int B, int w, float* red, F f) {
  const int rg_n = MID_THREADS / w;           // row groups
  const int c = threadIdx.x % w, rg = threadIdx.x / w;
  float s = 0.f;
  if (rg < rg_n)
    for (int r = rg; r < B; r += rg_n) s += f(r, c);
  __syncthreads();
  red[threadIdx.x] = (rg < rg_n) ? s : 0.f;
  __syncthreads();
  float tot = 0.f;
  for (int g = 0; g < rg_n; ++g) tot += red[g * w + c];   // fixed order
  return tot;
}

// BatchNorm + ReLU + Dropout on act (LDS, in place), saving xhat / out for the backward pass.
__device__ inline void mid_bn_act(const MidArgs& a, const MidLayer& L, float* act, float* red) {
  const int w = L.out_p, B = a.B;
  const int c = threadIdx.x % w, rg = threadIdx.x / w;
  const bool live = c < L.out;
  float mean = 0.f, inv = 1.f, gamma = 1.f, beta = 0.f;
  if (a.batchnorm) {
    gamma = live ? L.gamma[c] : 0.f;
    beta = live ? L.beta[c] : 0.f;
    float var;
    if (a.training) {
      const float s1 = col_sum(B, w, red, [&](int r, int cc) { return act[r * MID_LD + cc]; });
      mean = s1 / (float)B;
      const float s2 = col_sum(B, w, red, [&](int r, int cc) { const float d = act[r * MID_LD + cc] - mean; return d * d; });
      var = s2 / (float)B;
      if (rg == 0) {
        if (L.batch_mean) { L.batch_mean[c] = mean; L.batch_var[c] = var; }
        if (a.update_moving && live) {
          L.moving_mean[c] = L.moving_mean[c] * a.momentum + mean * (1.f - a.momentum);
          L.moving_var[c] = L.moving_var[c] * a.momentum + var * (1.f - a.momentum);
        }
      }
    } else {
      mean = live ? L.moving_mean[c] : 0.f;
      var = live ? L.moving_var[c] : 1.f;
    }
    inv = rsqrtf(var + a.eps);
    if (rg == 0) L.inv_std[c] = inv;
  }
  // per-column affine form of BN for the second pass: v = x * sc + sh, y = gamma * v + beta
  float* col_sc = red;            // [w]
  float* col_sh = red + MID_MAXW; // [w]
  float* col_g = red + 2 * MID_MAXW;
  float* col_b = red + 3 * MID_MAXW;
  __syncthreads();
  if (rg == 0) {
    col_sc[c] = a.batchnorm ? inv : 1.f;
    col_sh[c] = a.batchnorm ? -mean * inv : 0.f;
    col_g[c] = a.batchnorm ? gamma : 1.f;
    col_b[c] = a.batchnorm ? beta : 0.f;
  }
  __syncthreads();
  const bool drop = a.training && L.drop_p > 0.f;
  const float scale = drop ? 1.f / (1.f - L.drop_p) : 1.f;
  const NoiseKey nk = mid_key(a, L.stream);
  // one thread = 4 consecutive columns of one row: one Philox block serves all four
  const int wq = w >> 2;
  for (int idx = threadIdx.x; idx < B * wq; idx += MID_THREADS) {
    const int r = idx / wq, c0 = (idx % wq) * 4;
    float4 mult = make_float4(1.f, 1.f, 1.f, 1.f);
    if (drop) {
      if (L.inj_mask) mult = *reinterpret_cast<const float4*>(L.inj_mask + (long)r * L.inj_ld + c0);
      else {
        const uint32_t cell = a.cell_base + (uint32_t)(a.rows ? a.rows[r] : r);
        mult = dropout_mult4(philox_block(nk, cell, (uint32_t)(c0 >> 2)), L.drop_p, scale);
      }
    }
    const float mm[4] = {mult.x, mult.y, mult.z, mult.w};
    float xv[4], hv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int cc = c0 + e;
      const float v = act[r * MID_LD + cc] * col_sc[cc] + col_sh[cc];
      const float y = a.batchnorm ? col_g[cc] * v + col_b[cc] : v;
      float h = fmaxf(y, 0.f) * mm[e];
      h = cc < L.out ? h : 0.f;
      xv[e] = v; hv[e] = h;
      act[r * MID_LD + cc] = h;
    }
    *reinterpret_cast<float4*>(L.xhat + (long)r * w + c0) = make_float4(xv[0], xv[1], xv[2], xv[3]);
    *reinterpret_cast<float4*>(L.outb + (long)r * w + c0) = make_float4(hv[0], hv[1], hv[2], hv[3]);
  }
  __syncthreads();
}

__device__ inline void mid_dense(const MidArgs& a, const MidLayer& L, const float* cur, float* nxt) {
  const float* W = L.W; const int ldw = L.ldw; const int B = a.B;
  const float* bias = a.batchnorm ? nullptr : L.bias;
  block_gemm<0>(B, L.out_p, L.in_p, W, ldw,
                [&](int r, int k) { return r < B ? cur[r * MID_LD + k] : 0.f; },
                [&](int r, int n, float v) { if (r < B) nxt[r * MID_LD + n] = v + (bias ? bias[n] : 0.f); });
  __syncthreads();
}

#define MID_STAMP(i) do { if (a.dbg && threadIdx.x == 0) a.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)

__global__ __launch_bounds__(MID_THREADS) void mid_fwd_kernel(MidArgs a_by_value) {
  // kernarg segment pointer: run-time indexing of a.enc[i] / a.dec[i] must not spill the struct to scratch
  const MidArgs& a = *(const MidArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* bufA = lds;
  float* bufB = lds + MID_ROWS * MID_LD;
  float* red = lds + 2 * MID_ROWS * MID_LD;   // [MID_THREADS]
  const int B = a.B;
  MID_STAMP(0);
  // ---- load the activated output of encoder layer 0 ----
  for (int idx = threadIdx.x; idx < B * a.h0_w; idx += MID_THREADS) {
    const int r = idx / a.h0_w, c = idx % a.h0_w;
    bufA[r * MID_LD + c] = a.h0[idx];
  }
  __syncthreads();
  float* cur = bufA; float* nxt = bufB;
  MID_STAMP(1);
  for (int i = 0; i < a.n_enc; ++i) {
    mid_dense(a, a.enc[i], cur, nxt);
    mid_bn_act(a, a.enc[i], nxt, red);
    float* t = cur; cur = nxt; nxt = t;
  }
  // ---- latent head: lat = h W + b ----
  {
    const float* W = a.Wlat; const int ldw = a.ld_wlat; const float* bias = a.blat; const int ld = a.lat_ld;
    float* latbuf = a.latbuf;
    block_gemm<0>(B, a.lat_ld, a.lat_in_p, W, ldw,
                  [&](int r, int k) { return r < B ? cur[r * MID_LD + k] : 0.f; },
                  [&](int r, int n, float v) {
                    if (r < B) { v += bias[n]; nxt[r * MID_LD + n] = v; latbuf[(long)r * ld + n] = v; }
                  });
    __syncthreads();
  }
  MID_STAMP(2);
  // ---- sample / KL: one thread = 4 consecutive latent dims of one cell (one Philox block);
  //      the Dp/4 threads of a cell are adjacent lanes, KL is a short shuffle reduction ----
  {
    const NoiseKey nk = mid_key(a, 64u /* STREAM_EPS_Z */);
    const int dq = a.Dp >> 2;   // 8 or 16 quads per cell
    const int total = ((B * dq + 63) / 64) * 64;
    for (int idx = threadIdx.x; idx < total; idx += MID_THREADS) {
      const int b = idx / dq, d0 = (idx % dq) * 4;
      float kl = 0.f;
      if (b < B) {
        float zz[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {1.f, 1.f, 1.f, 1.f}, ee[4] = {0.f, 0.f, 0.f, 0.f};
        float4 n4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.stochastic) {
          if (a.inj_eps) n4 = *reinterpret_cast<const float4*>(a.inj_eps + (long)b * a.inj_eps_ld + d0);
          else n4 = normal4(philox_block(nk, a.cell_base + (uint32_t)(a.rows ? a.rows[b] : b), (uint32_t)(d0 >> 2)));
        }
        const float nn[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int d = d0 + e;
          if (d < a.D) {
            const float mu = nxt[b * MID_LD + d];
            if (a.stochastic) {
              const float sg = softplusf(nxt[b * MID_LD + a.Dp + d] + SMX_SOFTPLUS_INV_1);
              ss[e] = sg; ee[e] = nn[e];
              zz[e] = mu + sg * nn[e];
              kl += 0.5f * (sg * sg + mu * mu - 1.f - 2.f * flog(sg));
            } else {
              zz[e] = a.relu ? fmaxf(mu, 0.f) : mu;
            }
          }
          cur[b * MID_LD + d] = zz[e];
        }
        *reinterpret_cast<float4*>(a.z + (long)b * a.Dp + d0) = make_float4(zz[0], zz[1], zz[2], zz[3]);
        if (a.sig) {
          *reinterpret_cast<float4*>(a.sig + (long)b * a.Dp + d0) = make_float4(ss[0], ss[1], ss[2], ss[3]);
          *reinterpret_cast<float4*>(a.eps_out + (long)b * a.Dp + d0) = make_float4(ee[0], ee[1], ee[2], ee[3]);
        }
      }
      for (int off = 1; off < dq; off <<= 1) kl += __shfl_xor(kl, off, 64);
      if (b < B && (idx % dq) == 0 && a.kl) a.kl[b] = kl;
    }
    __syncthreads();
  }
  MID_STAMP(3);
  // ---- decoder ----
  for (int i = 0; i < a.n_dec; ++i) {
    mid_dense(a, a.dec[i], cur, nxt);
    MID_STAMP(4 + 2 * i);
    mid_bn_act(a, a.dec[i], nxt, red);
    MID_STAMP(5 + 2 * i);
    float* t = cur; cur = nxt; nxt = t;
  }
}

// ===========================================================================
// backward middle: from d loss / d pre-activation of the LAST decoder layer (global, produced by
// the slab-consuming bn_act_bwd kernel) back to d loss / d pre-activation of encoder layer 0.
// Weight gradients of the layers in between are plain GEMMs on the saved dpre buffers and run
// on the side stream (host), off the critical path.
// ===========================================================================
__device__ inline void mid_bn_act_bwd(const MidArgs& a, const float* out, const float* xhat, const float* inv_std,
                                      const float* gamma_p, int w_log, int w, float drop_p, float* dgamma, float* dbeta,
                                      float* dbias, float* g /* LDS [B][MID_LD]: d out in, d pre out */, float* dpre_glb,
                                      float* red) {
  const int B = a.B;
  const int rg_n = MID_THREADS / w;
  const int c = threadIdx.x % w, rg = threadIdx.x / w;
  const bool live = c < w_log;
  const float dscale = (a.training && drop_p > 0.f) ? 1.f / (1.f - drop_p) : 1.f;
  if (rg < rg_n)
    for (int r = rg; r < B; r += rg_n) {
      const float v = g[r * MID_LD + c];
      g[r * MID_LD + c] = (live && out[(long)r * w + c] > 0.f) ? v * dscale : 0.f;
    }
  const float s1 = col_sum(B, w, red, [&](int r, int cc) { return g[r * MID_LD + cc]; });
  if (!a.batchnorm) {
    if (rg == 0 && dbias && live) dbias[c] = s1;
    if (rg < rg_n)
      for (int r = rg; r < B; r += rg_n) dpre_glb[(long)r * w + c] = g[r * MID_LD + c];
    __syncthreads();
    return;
  }
  const float s2 = col_sum(B, w, red, [&](int r, int cc) { return g[r * MID_LD + cc] * xhat[(long)r * w + cc]; });
  const float gamma = live ? gamma_p[c] : 0.f;
  const float inv = inv_std[c];
  if (rg == 0) { dgamma[c] = live ? s2 : 0.f; dbeta[c] = live ? s1 : 0.f; }
  const float invB = 1.f / (float)B;
  if (rg < rg_n)
    for (int r = rg; r < B; r += rg_n) {
      const float dy = g[r * MID_LD + c];
      float d;
      if (a.training) d = gamma * inv * (dy - invB * (s1 + xhat[(long)r * w + c] * s2));
      else d = dy * gamma * inv;
      g[r * MID_LD + c] = d;
      dpre_glb[(long)r * w + c] = d;
    }
  __syncthreads();
}

// d in = d pre * W^T : [B x in_p] from cur [B x out_p] (LDS) and W [in_p][ldw] (global)
__device__ inline void mid_dx(const MidArgs& a, const float* W, int ldw, int in_p, int out_p, const float* cur, float* nxt) {
  const int B = a.B;
  block_gemm<1>(B, in_p, out_p, W, ldw,
                [&](int r, int k) { return r < B ? cur[r * MID_LD + k] : 0.f; },
                [&](int r, int n, float v) { if (r < B) nxt[r * MID_LD + n] = v; });
  __syncthreads();
}

__global__ __launch_bounds__(MID_THREADS) void mid_bwd_kernel(MidArgs a_by_value) {
  const MidArgs& a = *(const MidArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* bufA = lds;
  float* bufB = lds + MID_ROWS * MID_LD;
  float* red = lds + 2 * MID_ROWS * MID_LD;
  const int B = a.B;
  const MidLayer& last = a.dec[a.n_dec - 1];
  for (int idx = threadIdx.x; idx < B * last.out_p; idx += MID_THREADS) {
    const int r = idx / last.out_p, c = idx % last.out_p;
    bufA[r * MID_LD + c] = last.dpre[idx];
  }
  __syncthreads();
  float* cur = bufA; float* nxt = bufB;
  for (int i = a.n_dec - 1; i >= 0; --i) {
    const MidLayer& L = a.dec[i];
    mid_dx(a, L.W, L.ldw, L.in_p, L.out_p, cur, nxt);
    { float* t = cur; cur = nxt; nxt = t; }
    if (i > 0) {
      const MidLayer& P = a.dec[i - 1];
      mid_bn_act_bwd(a, P.outb, P.xhat, P.inv_std, P.gamma, P.out, P.out_p, P.drop_p, P.dgamma, P.dbeta, P.dbias, cur, P.dpre, red);
    }
  }
  // cur = d z [B][Dp]; latent backward -> d lat in nxt (+ global)
  {
    const int ld = a.lat_ld;
    for (int idx = threadIdx.x; idx < B * a.Dp; idx += MID_THREADS) {
      const int b = idx / a.Dp, d = idx % a.Dp;
      const float dz = cur[b * MID_LD + d];
      if (a.stochastic) {
        float dmu = 0.f, ds = 0.f;
        if (d < a.D) {
          const float mu = a.latbuf[(long)b * ld + d], sraw = a.latbuf[(long)b * ld + a.Dp + d];
          const float sig = a.sig[idx], eps = a.eps_out[idx];
          dmu = dz + a.kl_scale * mu;
          ds = (dz * eps + a.kl_scale * (sig - 1.f / sig)) * sigmoidf(sraw + SMX_SOFTPLUS_INV_1);
        }
        nxt[b * MID_LD + d] = dmu; nxt[b * MID_LD + a.Dp + d] = ds;
        a.dlat[(long)b * ld + d] = dmu; a.dlat[(long)b * ld + a.Dp + d] = ds;
      } else {
        float g = 0.f;
        if (d < a.D) g = (a.relu && !(a.latbuf[(long)b * ld + d] > 0.f)) ? 0.f : dz;
        nxt[b * MID_LD + d] = g;
        a.dlat[(long)b * ld + d] = g;
      }
    }
    __syncthreads();
    float* t = cur; cur = nxt; nxt = t;
  }
  // d h = d lat * Wlat^T, then back through the encoder layers held in the middle
  mid_dx(a, a.Wlat, a.ld_wlat, a.lat_in_p, a.lat_ld, cur, nxt);
  { float* t = cur; cur = nxt; nxt = t; }
  for (int i = a.n_enc - 1; i >= 0; --i) {
    const MidLayer& L = a.enc[i];
    mid_bn_act_bwd(a, L.outb, L.xhat, L.inv_std, L.gamma, L.out, L.out_p, L.drop_p, L.dgamma, L.dbeta, L.dbias, cur, L.dpre, red);
    mid_dx(a, L.W, L.ldw, L.in_p, L.out_p, cur, nxt);
    { float* t = cur; cur = nxt; nxt = t; }
  }
  mid_bn_act_bwd(a, a.enc0_out, a.enc0_xhat, a.enc0_inv_std, a.enc0_gamma, a.enc0_out_w, a.enc0_out_p, a.enc0_drop_p,
                 a.enc0_dgamma, a.enc0_dbeta, a.enc0_dbias, cur, a.dpre_enc0, red);
}

constexpr size_t MID_LDS_BYTES = (size_t)(2 * MID_ROWS * MID_LD + MID_THREADS) * sizeof(float);

int launch_mid_fwd(hipStream_t st, const MidArgs& a) {
  static bool attr_set = false;
  if (!attr_set) {
    SMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mid_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)MID_LDS_BYTES));
    attr_set = true;
  }
  hipLaunchKernelGGL(mid_fwd_kernel, dim3(1), dim3(MID_THREADS), MID_LDS_BYTES, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

int launch_mid_bwd(hipStream_t st, const MidArgs& a) {
  static bool attr_set = false;
  if (!attr_set) {
    SMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mid_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)MID_LDS_BYTES));
    attr_set = true;
  }
  hipLaunchKernelGGL(mid_bwd_kernel, dim3(1), dim3(MID_THREADS), MID_LDS_BYTES, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

}  // namespace smx
