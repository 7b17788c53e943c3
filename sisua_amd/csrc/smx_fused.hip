// smx_fused.hip -- small-layer fusions that remove launches from the forward critical path.
//
//  latent_head_fwd   lat = h W + b, sigma = softplus1(s), z = mu + sigma eps (Philox), KL -- one kernel.
//                    A workgroup owns 32 cells and ALL 2*Dp columns, so mu and s of a cell meet in LDS
//                    and the sample / KL run in the epilogue (was: GEMM launch + elementwise launch).
//  dense_bn_act_fwd  out = dropout(relu(BN(in W))) for layers whose input width fits one K pass
//                    (decoder layers, encoder layers >= 1).  A workgroup owns 32 output columns and ALL
//                    cells (B <= 128), so the batch statistics of its columns are complete inside the
//                    workgroup: GEMM, BatchNorm, ReLU, Dropout in one launch (was: GEMM + BN launches).
//
// Both stage the activation operand through LDS (K-major, conflict-free MFMA operand reads), stream the
// weight operand straight from global memory into registers (coalesced, whole K in flight), and use
// the exact-fp32 MFMA v_mfma_f32_32x32x2_f32.
#include "smx_internal.h"
#include "../../include/sisua_hip.h"

namespace smx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FZ_MAXK = 128;        // widest input handled by the fused small-layer kernels
constexpr int FZ_ROWS = 128;        // cells per workgroup in dense_bn_act_fwd

// ---------------------------------------------------------------------------------------------
// latent head
// ---------------------------------------------------------------------------------------------
struct LatentHeadArgs {
  const float* h; int ldh; int K;          // [B][ldh], K = padded input width (<= 128)
  const float* W; int ldw; const float* bias;  // [K][ldw = lat_ld]
  int B, D, Dp, lat_ld, stochastic, relu;
  NoiseKey nk; const int32_t* rows; uint32_t cell_base;
  const float* inj_eps; int inj_ld;
  float* latbuf; float* z; float* sig; float* eps; float* kl;
};

__global__ __launch_bounds__(256) void latent_head_fwd_kernel(LatentHeadArgs a) {
  __shared__ float As[FZ_MAXK * 33];          // [k][32 rows + 1]
  __shared__ float Ls[32 * (FZ_MAXK + 1)];    // [32 rows][lat_ld + 1] (also the K-split reduction scratch)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * 32;
  const int K = a.K, ntn = a.lat_ld >> 5;     // column tiles: 1, 2 or 4
  const int wk_n = 4 / ntn;                   // waves along K per column tile
  const int wn = wave % ntn, wk = wave / ntn;
  // stage h tile transposed: thread = (row, 4 consecutive k)
  for (int f = tid; f < 32 * (K >> 2); f += 256) {
    const int r = f / (K >> 2), kq = f % (K >> 2);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m0 + r < a.B) v = *reinterpret_cast<const float4*>(a.h + (long)(m0 + r) * a.ldh + kq * 4);
    As[(kq * 4 + 0) * 33 + r] = v.x; As[(kq * 4 + 1) * 33 + r] = v.y;
    As[(kq * 4 + 2) * 33 + r] = v.z; As[(kq * 4 + 3) * 33 + r] = v.w;
  }
  // this wave's slice of K: [k0, k0 + ks); lane half h takes the upper / lower half of the slice
  const int ks = K / wk_n, k0 = wk * ks, kh = ks >> 1;
  float bfrag[FZ_MAXK / 2];
  const float* wsrc = a.W + (long)(k0 + lh * kh) * a.ldw + wn * 32 + li;
#pragma unroll
  for (int s = 0; s < FZ_MAXK / 2; ++s) bfrag[s] = (s < kh) ? wsrc[(long)s * a.ldw] : 0.f;
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* as = As + (k0 + lh * kh) * 33 + li;
#pragma unroll
  for (int s = 0; s < FZ_MAXK / 2; ++s)
    if (s < kh) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[s * 33], bfrag[s], acc, 0, 0, 0);
  // K-split partial tiles meet in LDS (fixed order: deterministic)
  const int ldl = a.lat_ld + 1;
  for (int q = wk_n - 1; q >= 1; --q) {
    if (wk == q) {
#pragma unroll
      for (int r = 0; r < 16; ++r) Ls[((r & 3) + 8 * (r >> 2) + 4 * lh) * ldl + wn * 32 + li] = acc[r];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += Ls[((r & 3) + 8 * (r >> 2) + 4 * lh) * ldl + wn * 32 + li];
    }
    __syncthreads();
  }
  if (wk == 0) {
    const float bias = a.bias[wn * 32 + li];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float v = acc[r] + bias;
      Ls[row * ldl + wn * 32 + li] = v;
      if (m0 + row < a.B) a.latbuf[(long)(m0 + row) * a.lat_ld + wn * 32 + li] = v;
    }
  }
  __syncthreads();
  // sample / KL: one thread = 4 consecutive latent dims of one cell (one Philox block)
  const int dq = a.Dp >> 2;
  for (int idx = tid; idx < 32 * dq; idx += 256) {   // 32*dq is a multiple of 64: whole waves iterate together
    const int r = idx / dq, d0 = (idx % dq) * 4;
    const int b = m0 + r;
    float kl = 0.f;
    if (b < a.B) {
      float zz[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {1.f, 1.f, 1.f, 1.f}, ee[4] = {0.f, 0.f, 0.f, 0.f};
      float4 n4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.stochastic) {
        if (a.inj_eps) n4 = *reinterpret_cast<const float4*>(a.inj_eps + (long)b * a.inj_ld + d0);
        else n4 = normal4(philox_block(a.nk, a.cell_base + (uint32_t)(a.rows ? a.rows[b] : b), (uint32_t)(d0 >> 2)));
      }
      const float nn[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = d0 + e;
        if (d < a.D) {
          const float mu = Ls[r * ldl + d];
          if (a.stochastic) {
            const float sg = softplusf(Ls[r * ldl + a.Dp + d] + SMX_SOFTPLUS_INV_1);
            ss[e] = sg; ee[e] = nn[e];
            zz[e] = mu + sg * nn[e];
            kl += 0.5f * (sg * sg + mu * mu - 1.f - 2.f * flog(sg));
          } else {
            zz[e] = a.relu ? fmaxf(mu, 0.f) : mu;
          }
        }
      }
      *reinterpret_cast<float4*>(a.z + (long)b * a.Dp + d0) = make_float4(zz[0], zz[1], zz[2], zz[3]);
      if (a.sig) {
        *reinterpret_cast<float4*>(a.sig + (long)b * a.Dp + d0) = make_float4(ss[0], ss[1], ss[2], ss[3]);
        *reinterpret_cast<float4*>(a.eps + (long)b * a.Dp + d0) = make_float4(ee[0], ee[1], ee[2], ee[3]);
      }
    }
    for (int off = 1; off < dq; off <<= 1) kl += __shfl_xor(kl, off, 64);
    if (b < a.B && (idx % dq) == 0 && a.kl) a.kl[b] = kl;
  }
}

int launch_latent_head_fwd(hipStream_t st, const LatentArgs& la, const float* h, int ldh, int K, const float* W, int ldw,
                           const float* bias, float* latbuf) {
  LatentHeadArgs a;
  a.h = h; a.ldh = ldh; a.K = K; a.W = W; a.ldw = ldw; a.bias = bias;
  a.B = la.B; a.D = la.D; a.Dp = la.Dp; a.lat_ld = la.ld; a.stochastic = la.stochastic; a.relu = la.relu;
  a.nk = la.nk; a.rows = la.rows; a.cell_base = la.cell_base; a.inj_eps = la.inj_eps; a.inj_ld = la.inj_ld;
  a.latbuf = latbuf; a.z = la.z; a.sig = la.sig; a.eps = la.eps; a.kl = la.kl;
  hipLaunchKernelGGL(latent_head_fwd_kernel, dim3((la.B + 31) / 32), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

bool latent_head_fusable(int K, int lat_ld, int Dp) {
  const int ntn = lat_ld / 32;
  const int dq = Dp / 4;
  return K % 32 == 0 && K <= FZ_MAXK && (ntn == 1 || ntn == 2 || ntn == 4) && (K / (4 / ntn)) % 2 == 0 &&
         (dq & (dq - 1)) == 0;
}

// ---------------------------------------------------------------------------------------------
// Dense + BatchNorm + ReLU + Dropout, one launch
// ---------------------------------------------------------------------------------------------
struct DenseBnArgs {
  const float* in; int ldi; int K;      // [B][ldi]
  const float* W; int ldw;              // [K][ldw]
  BnFwdArgs bn;                         // B, H, Hp, statistics, outputs, dropout (pre / slabs unused)
};

__device__ inline float col_reduce_4waves(float v, float* sh /*[4][32]*/, int wave, int li, int lh) {
  v += __shfl_xor(v, 32, 64);           // the two row halves of the wave's 32 x 32 tile
  __syncthreads();
  if (lh == 0) sh[wave * 32 + li] = v;
  __syncthreads();
  return (sh[li] + sh[32 + li]) + (sh[64 + li] + sh[96 + li]);
}

__global__ __launch_bounds__(256) void dense_bn_act_fwd_kernel(DenseBnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float fz_lds[];   // [K][128 rows + 1] + [4][32]
  float* As = fz_lds;
  float* sh = fz_lds + a.K * (FZ_ROWS + 1);
  const BnFwdArgs& bn = a.bn;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * 32, col = n0 + li;
  const int K = a.K, B = bn.B;
  constexpr int LDA = FZ_ROWS + 1;
  for (int f = tid; f < FZ_ROWS * (K >> 2); f += 256) {
    const int r = f / (K >> 2), kq = f % (K >> 2);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < B) v = *reinterpret_cast<const float4*>(a.in + (long)r * a.ldi + kq * 4);
    As[(kq * 4 + 0) * LDA + r] = v.x; As[(kq * 4 + 1) * LDA + r] = v.y;
    As[(kq * 4 + 2) * LDA + r] = v.z; As[(kq * 4 + 3) * LDA + r] = v.w;
  }
  const int kh = K >> 1;
  float bfrag[FZ_MAXK / 2];
  const float* wsrc = a.W + (long)(lh * kh) * a.ldw + col;
#pragma unroll
  for (int s = 0; s < FZ_MAXK / 2; ++s) bfrag[s] = (s < kh) ? wsrc[(long)s * a.ldw] : 0.f;
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* as = As + (lh * kh) * LDA + wave * 32 + li;
#pragma unroll
  for (int s = 0; s < FZ_MAXK / 2; ++s)
    if (s < kh) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[s * LDA], bfrag[s], acc, 0, 0, 0);
  // lane holds column `col`, rows wave*32 + (r&3) + 8(r>>2) + 4 lh
  const bool live = col < bn.H;
  const float bias = (!bn.batchnorm && bn.bias && live) ? bn.bias[col] : 0.f;
  float v[16], mult[16];
  bool ok[16];
  float s1 = 0.f;
  const bool drop = bn.training && bn.drop_p > 0.f;
  const float scale = drop ? 1.f / (1.f - bn.drop_p) : 1.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    ok[r] = row < B;
    v[r] = acc[r] + bias;
    if (ok[r]) s1 += v[r];
    // dropout multipliers: issue every load now (they complete under the two column reductions)
    mult[r] = 1.f;
    if (drop && ok[r]) {
      if (bn.inj_mask) mult[r] = bn.inj_mask[(long)row * bn.inj_ld + col];
      else {
        const uint32_t cell = bn.cell_base + (uint32_t)(bn.rows ? bn.rows[row] : row);
        mult[r] = dropout_mult1(philox_block(bn.nk, cell, (uint32_t)(col >> 2)), col & 3, bn.drop_p, scale);
      }
    }
  }
  float mean = 0.f, inv = 1.f, gamma = 1.f, beta = 0.f;
  if (bn.batchnorm) {
    gamma = live ? bn.gamma[col] : 0.f;
    beta = live ? bn.beta[col] : 0.f;
    float var;
    if (bn.training) {
      s1 = col_reduce_4waves(s1, sh, wave, li, lh);
      mean = s1 / (float)B;
      float s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (ok[r]) { const float d = v[r] - mean; s2 += d * d; }
      s2 = col_reduce_4waves(s2, sh, wave, li, lh);
      var = s2 / (float)B;
      if (wave == 0 && lh == 0) {
        if (bn.batch_mean) { bn.batch_mean[col] = mean; bn.batch_var[col] = var; }
        if (bn.update_moving && live) {
          bn.moving_mean[col] = bn.moving_mean[col] * bn.momentum + mean * (1.f - bn.momentum);
          bn.moving_var[col] = bn.moving_var[col] * bn.momentum + var * (1.f - bn.momentum);
        }
      }
    } else {
      mean = live ? bn.moving_mean[col] : 0.f;
      var = live ? bn.moving_var[col] : 1.f;
    }
    inv = rsqrtf(var + bn.eps);
    if (wave == 0 && lh == 0 && bn.inv_std) bn.inv_std[col] = inv;
  }
  float xo[16], ho[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float x = v[r], y = x;
    if (bn.batchnorm) { x = (x - mean) * inv; y = gamma * x + beta; }
    xo[r] = x;
    ho[r] = live ? fmaxf(y, 0.f) * mult[r] : 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (ok[r]) {
      const long o = (long)row * bn.Hp + col;
      bn.xhat[o] = xo[r];
      bn.out[o] = ho[r];
    }
  }
}

int launch_dense_bn_act_fwd(hipStream_t st, const float* in, int ldi, int K, const float* W, int ldw, const BnFwdArgs& bn) {
  DenseBnArgs a;
  a.in = in; a.ldi = ldi; a.K = K; a.W = W; a.ldw = ldw; a.bn = bn;
  static bool attr_set = false;
  if (!attr_set) {
    SMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dense_bn_act_fwd_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)((FZ_MAXK * (FZ_ROWS + 1) + 128) * sizeof(float))));
    attr_set = true;
  }
  const size_t lds = (size_t)(K * (FZ_ROWS + 1) + 128) * sizeof(float);
  hipLaunchKernelGGL(dense_bn_act_fwd_kernel, dim3(bn.Hp / 32), dim3(256), lds, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

bool dense_bn_fusable(int B, int K) { return B <= FZ_ROWS && K % 32 == 0 && K <= FZ_MAXK; }

}  // namespace smx
