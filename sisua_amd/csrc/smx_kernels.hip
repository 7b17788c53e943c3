// smx_kernels.hip -- the fused non-GEMM kernels of the SISUA step (gfx950).
//
//  count_loss     NB / ZINB / NBD / ZINBD log-likelihood, forward + gradient wrt the
//                 parameter planes, one pass over B x G (the bandwidth-bound kernel the
//                 roofline is quoted on: SURVEY.md 8d, rows a-10/a-11)
//  bn_act_fwd/bwd (split-K slab sum) -> BatchNorm -> ReLU -> Dropout and its backward
//  latent_fwd/bwd diagonal-Gaussian head: softplus1, reparameterised sample (Philox),
//                 analytic KL (a-7); deterministic DCA latent (a-14)
//  label_loss     masked NB / one-hot label heads of SISUA (a-13)
//  scvi_head      softmax-rate / exp-dispersion head of SCVI and its backward (a-12)
//  metrics        ELBO scalars (a-15);  adam: per-tensor clipnorm + Adam (a-16)
#include <stdlib.h>

#include "smx_internal.h"
#include "smx_loss.h"
#include "smx_adam.h"
#include "../../include/sisua_hip.h"

namespace smx {
SMX_STAMP_TABLE


// grid (n_chunks, B); thread = VEC consecutive genes of one cell (VEC*4-byte accesses).
template <int VEC> struct VecT;
template <> struct VecT<4> { typedef float4 T; };
template <> struct VecT<2> { typedef float2 T; };
template <> struct VecT<1> { typedef float T; };

template <int VEC>
__device__ inline void vload(const float* p, float (&v)[VEC]) {
  const typename VecT<VEC>::T t = *reinterpret_cast<const typename VecT<VEC>::T*>(p);
  const float* f = reinterpret_cast<const float*>(&t);
#pragma unroll
  for (int i = 0; i < VEC; ++i) v[i] = f[i];
}
template <int VEC>
__device__ inline void vstore(float* p, const float (&v)[VEC]) {
  typename VecT<VEC>::T t;
  float* f = reinterpret_cast<float*>(&t);
#pragma unroll
  for (int i = 0; i < VEC; ++i) f[i] = v[i];
  *reinterpret_cast<typename VecT<VEC>::T*>(p) = t;
}

template <int LK, int DIRECT, int BWD, int VEC, int BLOCK = 256, int U16 = 0>   // U16: counts from the compact uint16 store
__global__ __launch_bounds__(BLOCK) void count_loss_kernel(LossArgs a) {
  constexpr int K = LK == SMX_LLK_MSE ? 1 : (LK == SMX_LLK_ZINB || LK == SMX_LLK_ZINBD) ? 3 : 2;
  constexpr int LKC = LK == SMX_LLK_MSE ? SMX_LLK_NB : LK;
  // every wave's non-zero counts go through ONE compacted pass of the lgamma / digamma code (smx_loss.h: lgamma_digamma_diff_queue)
  __shared__ float2 lq[LK == SMX_LLK_MSE ? 1 : BLOCK * VEC];
  const float inv_g = 1.f / (float)a.G;   // SMX_LLK_MSE: -log p = mean over the genes of (x - mean)^2
  const int b = blockIdx.y;
  const int g0 = (blockIdx.x * BLOCK + threadIdx.x) * VEC;
  const bool in = g0 < a.Gp;   // (lanes beyond the row carry zero counts)
  float acc = 0.f;
  float xs[VEC], a0[VEC], a1[VEC], a2[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { xs[e] = 0.f; a0[e] = 0.f; a1[e] = 0.f; a2[e] = 0.f; }
  if (in) {
    const long src = a.rows ? a.rows[b] : b;
    const float* pb = a.P + (long)b * a.ldp + g0;
    if (U16) {
      const uint16_t* xh = reinterpret_cast<const uint16_t*>(a.X) + src * a.ldx + g0;
      if (VEC == 4) { const ushort4 h = *reinterpret_cast<const ushort4*>(xh); xs[0] = h.x; xs[1 % VEC] = h.y; xs[2 % VEC] = h.z; xs[3 % VEC] = h.w; }
      else if (VEC == 2) { const ushort2 h = *reinterpret_cast<const ushort2*>(xh); xs[0] = h.x; xs[1 % VEC] = h.y; }
      else xs[0] = (float)xh[0];
    } else {
      vload<VEC>(a.X + src * a.ldx + g0, xs);
    }
    vload<VEC>(pb, a0);
    if (K >= 2) vload<VEC>(pb + a.plane_stride, a1);
    if (K == 3) vload<VEC>(pb + 2 * a.plane_stride, a2);
  }
  float r0[VEC], r1[VEC], r2[VEC];
  if (a.likelihood < 0 || LK == SMX_LLK_MSE) {   // (launch-uniform) the diagnostics and the deterministic output: element by element
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float llk = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f;
      if (a.likelihood == -1) {  // diagnostic: same traffic, no arithmetic
        d0 = a0[e] + xs[e]; d1 = K >= 2 ? a1[e] : 0.f; d2 = K == 3 ? a2[e] : 0.f; acc += d0;
      } else if (a.likelihood == -2) {  // diagnostic: every count treated as 0 (no lgamma work)
        count_elem<LKC, DIRECT>(0.f, a0[e], K >= 2 ? a1[e] : 0.f, K == 3 ? a2[e] : 0.f, llk, d0, d1, d2);
        acc += llk + xs[e];
      } else if (in && g0 + e < a.G) {
        const float df = xs[e] - a0[e];
        llk = -(df * df) * inv_g; d0 = 2.f * df * inv_g;
        acc += llk;
      }
      r0[e] = d0 * a.grad_scale; r1[e] = d1 * a.grad_scale; r2[e] = d2 * a.grad_scale;
    }
  } else {
    float llk[VEC], d0[VEC], d1[VEC], d2[VEC], xq[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) xq[e] = (in && g0 + e < a.G) ? xs[e] : 0.f;
    count_elem_vec<LKC, DIRECT, VEC>(xq, a0, a1, a2, llk, d0, d1, d2, lq);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const bool ok = in && g0 + e < a.G;
      acc += ok ? llk[e] : 0.f;
      r0[e] = ok ? d0[e] * a.grad_scale : 0.f; r1[e] = ok ? d1[e] * a.grad_scale : 0.f; r2[e] = ok ? d2[e] * a.grad_scale : 0.f;
    }
  }
  if (BWD && in) {
    float* db = a.dP + (long)b * a.ldp + g0;
    vstore<VEC>(db, r0);
    if (K >= 2) vstore<VEC>(db + a.plane_stride, r1);
    if (K == 3) vstore<VEC>(db + 2 * a.plane_stride, r2);
  }
  // one partial per wave, no workgroup barrier: the consumer sums [n_chunks * waves] values per cell
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0)
    a.llk_part[((long)b * gridDim.x + blockIdx.x) * (BLOCK / 64) + (threadIdx.x >> 6)] = acc;
}

static int loss_block() { return 256; }
// Elements per lane: 4-byte accesses win while the launch is latency-bound (one wave of workgroups), 8-byte
// from ~0.4 M elements (47 % of the HBM peak at 128 x 20 000 against 38 %), 16-byte from ~8 M (64 % at 1024 x
// 20 000) -- tools/loss_roofline.py.  SMX_LOSS_VEC forces a width.
static int loss_vec(int B, int Gp) {
  static const int forced = (int)tuning("loss_vec", 0);
  if (forced == 1 || forced == 2 || forced == 4) return forced;
  const long n = (long)B * Gp;
  return n < 400000 ? 1 : (n < 8000000 ? 2 : 4);
}
static int loss_grid_x(int Gp, int vec) { return (Gp + loss_block() * vec - 1) / (loss_block() * vec); }
// number of partial sums per cell the loss kernel writes (one per wave)
int loss_chunks(int Gp, int B) { return loss_grid_x(Gp, loss_vec(B, Gp)) * (loss_block() / 64); }
int loss_chunks_max(int Gp) { return loss_grid_x(Gp, 1) * (loss_block() / 64); }

template <int LK, int DIRECT>
static void launch_loss_t(hipStream_t st, const LossArgs& a, dim3 grid) {
  const int v = loss_vec(a.B, a.Gp);
#define SMX_LOSS_LAUNCH(B_, V_) do { \
    if (a.x_u16) hipLaunchKernelGGL((count_loss_kernel<LK, DIRECT, B_, V_, 256, 1>), grid, dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((count_loss_kernel<LK, DIRECT, B_, V_, 256>), grid, dim3(256), 0, st, a); } while (0)
  if (a.backward) { if (v == 4) SMX_LOSS_LAUNCH(1, 4); else if (v == 2) SMX_LOSS_LAUNCH(1, 2); else SMX_LOSS_LAUNCH(1, 1); }
  else { if (v == 4) SMX_LOSS_LAUNCH(0, 4); else if (v == 2) SMX_LOSS_LAUNCH(0, 2); else SMX_LOSS_LAUNCH(0, 1); }
#undef SMX_LOSS_LAUNCH
}

int launch_count_loss(hipStream_t st, const LossArgs& a) {
  if (a.B <= 0 || a.Gp % 4 || a.ldx % 4 || a.ldp % 4 || a.plane_stride % 4) {
    set_error("count_loss: bad shapes");
    return SMX_ERR_INVALID;
  }
  dim3 grid(loss_grid_x(a.Gp, loss_vec(a.B, a.Gp)), a.B);
  switch (a.likelihood) {
    case SMX_LLK_NB: launch_loss_t<SMX_LLK_NB, 0>(st, a, grid); break;
    case SMX_LLK_ZINB: launch_loss_t<SMX_LLK_ZINB, 0>(st, a, grid); break;
    case SMX_LLK_NBD:
      if (a.direct) launch_loss_t<SMX_LLK_NBD, 1>(st, a, grid); else launch_loss_t<SMX_LLK_NBD, 0>(st, a, grid);
      break;
    case SMX_LLK_ZINBD:
      if (a.direct) launch_loss_t<SMX_LLK_ZINBD, 1>(st, a, grid); else launch_loss_t<SMX_LLK_ZINBD, 0>(st, a, grid);
      break;
    case SMX_LLK_MSE: launch_loss_t<SMX_LLK_MSE, 0>(st, a, grid); break;
    default: set_error("count_loss: unknown likelihood"); return SMX_ERR_INVALID;
  }
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// ===========================================================================
// BatchNorm + ReLU + Dropout.  A workgroup owns BN_COLS columns and all rows:
// thread = (column c = tid % BN_COLS, row lane rl = tid / BN_COLS); rows are walked
// in chunks of BN_RL * BN_RPT with every slab load of a chunk in flight at once
// (the kernel is a latency chain, not a bandwidth problem).
// ===========================================================================
#ifndef SMX_BN_COLS
#define SMX_BN_COLS 8
#endif
constexpr int BN_COLS = SMX_BN_COLS;
constexpr int BN_RL = 64;
constexpr int BN_RPT_DEFAULT = 2;   // rows per thread; the register-resident kernels exist for 2, 4, 8, 16 (B <= 1024)
constexpr int BN_THREADS = BN_COLS * BN_RL;   // one wave per BN_COLS... waves = BN_THREADS / 64
constexpr int BN_WAVES = BN_THREADS / 64;

// column sum over the workgroup: lanes of a wave that share a column are 4 apart (xor 4..32),
// then the 4 waves meet in LDS; fixed order -> deterministic
__device__ inline float bn_col_reduce(float v, float* sh /*[BN_WAVES][BN_COLS]*/) {
  const int c = threadIdx.x % BN_COLS, w = threadIdx.x >> 6;
  static_assert(BN_COLS == 8, "bn_col_reduce: the lanes of a column are 8 apart");
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));   // row_ror:8 = lane ^ 8 (were three ds_bpermute round trips)
  v = xor32_add(xor16_add(v));
  __syncthreads();
  if ((threadIdx.x & 63) < BN_COLS) sh[w * BN_COLS + c] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int q = 0; q < BN_WAVES; q += 4)
    t += (sh[q * BN_COLS + c] + sh[(q + 1) * BN_COLS + c]) + (sh[(q + 2) * BN_COLS + c] + sh[(q + 3) * BN_COLS + c]);
  return t;
}

// sum of the split-K slabs for BN_RPT rows of one column, loads issued together.
// Up to SLAB_FLIGHT slabs' values of a thread (RPT rows each) are requested in ONE batch -- unconditional loads from a clamped slab index,
// left out at the add -- so that a column's sum costs one memory round trip, not one per batch of 8 plus one per remaining slab (the
// decoder's 12 slabs at BASELINE configs[1]: 5 dependent round trips, 4.6 of the launch's 9 us; tools/c2_stamps.sh).  The adds keep
// their order (slab 0, 1, ...): the same bits.
constexpr int SLAB_FLIGHT = 16;
template <int NR>
struct SlabBatch { float t[SLAB_FLIGHT][NR]; };
template <int NR>
__device__ inline void slab_issue(const float* base, int s0, int n_slabs, long slab_stride, int ld, int col, int r0, int rl, int B, SlabBatch<NR>& sb) {
#pragma unroll
  for (int q = 0; q < SLAB_FLIGHT; ++q) {
    const int s = min(s0 + q, n_slabs - 1);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int r = min(r0 + rl + BN_RL * i, B - 1);
      sb.t[q][i] = base[(long)s * slab_stride + (long)r * ld + col];
    }
  }
}
template <int NR>
__device__ inline void slab_accumulate(const SlabBatch<NR>& sb, int s0, int n_slabs, int r0, int rl, int B, float (&acc)[NR]) {
#pragma unroll
  for (int q = 0; q < SLAB_FLIGHT; ++q)
    if (s0 + q < n_slabs) {   // (block-uniform)
#pragma unroll
      for (int i = 0; i < NR; ++i) acc[i] += (r0 + rl + BN_RL * i < B) ? sb.t[q][i] : 0.f;
    }
}
template <int BN_RPT>
__device__ inline void slab_sum(const float* base, int n_slabs, long slab_stride, int ld, int col, int r0, int rl,
                                int B, float (&acc)[BN_RPT]) {
#pragma unroll
  for (int i = 0; i < BN_RPT; ++i) acc[i] = 0.f;
  if (BN_RPT <= 2) {   // (32 values in flight per lane)
    for (int s = 0; s < n_slabs; s += SLAB_FLIGHT) {
      SlabBatch<BN_RPT> sb;
      slab_issue<BN_RPT>(base, s, n_slabs, slab_stride, ld, col, r0, rl, B, sb);
      slab_accumulate<BN_RPT>(sb, s, n_slabs, r0, rl, B, acc);
    }
    return;
  }
  constexpr int SU = BN_RPT <= 4 ? 4 : 2;   // slabs per batch of loads: 16 values in flight per lane
  int s = 0;
  for (; s + SU <= n_slabs; s += SU) {
    float t[SU][BN_RPT];
#pragma unroll
    for (int q = 0; q < SU; ++q)
#pragma unroll
      for (int i = 0; i < BN_RPT; ++i) {
        const int r = r0 + rl + BN_RL * i;
        t[q][i] = r < B ? base[(long)(s + q) * slab_stride + (long)r * ld + col] : 0.f;
      }
#pragma unroll
    for (int q = 0; q < SU; ++q)
#pragma unroll
      for (int i = 0; i < BN_RPT; ++i) acc[i] += t[q][i];
  }
  for (; s < n_slabs; ++s)
#pragma unroll
    for (int i = 0; i < BN_RPT; ++i) {
      const int r = r0 + rl + BN_RL * i;
      if (r < B) acc[i] += base[(long)s * slab_stride + (long)r * ld + col];
    }
}

// extra workgroups of the BN launch: Philox multipliers / normals for later layers, 4 columns per thread
__device__ inline void noise_fill(const BnFwdArgs& a, int job_block) {
  const NoiseJob& j = a.jobs[job_block / SMX_NOISE_BLOCKS_PER_JOB];
  const int sub = job_block % SMX_NOISE_BLOCKS_PER_JOB;
  const int wq = (j.width + 3) >> 2;
  NoiseKey nk = a.nk;
  nk.stream = j.stream;
  const float scale = j.p > 0.f ? 1.f / (1.f - j.p) : 1.f;
  for (int idx = sub * BN_THREADS + threadIdx.x; idx < a.B * wq; idx += SMX_NOISE_BLOCKS_PER_JOB * BN_THREADS) {
    const int r = idx / wq, c0 = (idx % wq) * 4;
    const uint32_t cell = a.cell_base + (uint32_t)(a.rows ? a.rows[r] : r);
    const U4 w = philox_block(nk, cell, (uint32_t)(c0 >> 2));
    const float4 v = j.normal ? normal4(w) : dropout_mult4(w, j.p, scale);
    *reinterpret_cast<float4*>(j.dst + (long)r * j.ld + c0) = v;
  }
}

// RPT > 0: B <= BN_RL * RPT, every value of the column stays in registers (RPT rows per thread);
// RPT == 0: any B, the normalised values make a round trip through xhat
// the latent tile of the whole minibatch into LDS (row stride Dp + 4), one thread per 4 latent dims of a cell; the
// first workgroup also leaves z / sigma / eps / KL in memory for the backward pass (same arithmetic, same Philox
// blocks as latent_fwd_quad_kernel)
template <int MAXIT>
__device__ inline void latent_tile_to_lds(const LatentArgs& a, float* zs, bool store) {
  const int dq = a.Dp >> 2, ldz = a.Dp + 4;   // (rows 16-byte aligned: the dot products read the tile four k at a time)
  const int dsh = __builtin_ctz((unsigned)dq), dmask = dq - 1;   // dq is a power of two (bn_front_supported): shifts, not the ~35-instruction integer division per index
  const int total = a.B * dq;
  // every load of every iteration first (left as a loop the compiler waits for each iteration's loads in turn:
  // MAXIT serial round trips to data the previous launch has just written)
  float4 m4[MAXIT], s4[MAXIT], n4[MAXIT];
  uint32_t cell[MAXIT];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int total_r = (total + 63) & ~63;
  // Unconditional loads from a clamped index under block-uniform branches only: a lane-predicated load sits in a block of its own, and the
  // row id's `cell_base + rows[b]` inside such a block made the compiler wait for EVERY outstanding load of the iteration before the next
  // iteration's loads were issued -- two serial memory round trips ahead of the first dot product (tools/c2_stamps.sh).  The row id is only
  // read when the normals are drawn here (not when an earlier launch drew them: inj_eps).
  const bool need_cell = a.stochastic && !a.inj_eps;
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    m4[it] = z4; s4[it] = z4; n4[it] = z4; cell[it] = 0;
    if (it * BN_THREADS >= total_r) continue;   // block-uniform
    const int idx = min((int)threadIdx.x + it * BN_THREADS, total - 1);
    const int b = idx >> dsh, d0 = (idx & dmask) * 4;
    m4[it] = *reinterpret_cast<const float4*>(a.lat + (long)b * a.ld + d0);
    if (a.stochastic) s4[it] = *reinterpret_cast<const float4*>(a.lat + (long)b * a.ld + a.Dp + d0);
    if (a.stochastic && a.inj_eps) n4[it] = *reinterpret_cast<const float4*>(a.inj_eps + (long)b * a.inj_ld + d0);
    cell[it] = (uint32_t)b;
    if (need_cell && a.rows) cell[it] = (uint32_t)a.rows[b];
  }
  SMX_STAMP(1, 7);   // the tile's loads issued
#ifdef SMX_STAMPS
  if (m4[0].x == 12345.678f && s4[0].x == 1.f) zs[0] = n4[0].x;   // (the first iteration's operands have arrived)
  SMX_STAMP(1, 8);
#endif
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int idx = threadIdx.x + it * BN_THREADS;
    if (it * BN_THREADS >= ((total + 63) & ~63)) break;   // block-uniform
    const int b = idx >> dsh, d0 = (idx & dmask) * 4;
    float kl = 0.f;
    if (idx < total) {
      float4 zq = z4, sq = make_float4(1.f, 1.f, 1.f, 1.f), eq = z4;
      const float4 mq = m4[it];
      if (a.stochastic) {
        float4 nq = n4[it];
        if (!a.inj_eps) nq = normal4(philox_block(a.nk, a.cell_base + cell[it], (uint32_t)(d0 >> 2)));
        const float4 sr = s4[it];
        auto one = [&](int e, float mu, float s_raw, float nn, float& z, float& s, float& en) {
          if (d0 + e < a.D) {
            const float sg = softplusf(s_raw + SMX_SOFTPLUS_INV_1);
            s = sg; en = nn;
            z = mu + sg * nn;
            if (store) kl += 0.5f * (sg * sg + mu * mu - 1.f - 2.f * flog(sg));   // (block-uniform: the KL term belongs to the workgroup that stores)
          }
        };
        one(0, mq.x, sr.x, nq.x, zq.x, sq.x, eq.x);
        one(1, mq.y, sr.y, nq.y, zq.y, sq.y, eq.y);
        one(2, mq.z, sr.z, nq.z, zq.z, sq.z, eq.z);
        one(3, mq.w, sr.w, nq.w, zq.w, sq.w, eq.w);
      } else {
        auto one = [&](int e, float mu, float& z) { if (d0 + e < a.D) z = a.relu ? fmaxf(mu, 0.f) : mu; };
        one(0, mq.x, zq.x); one(1, mq.y, zq.y); one(2, mq.z, zq.z); one(3, mq.w, zq.w);
      }
      *reinterpret_cast<float4*>(zs + b * ldz + d0) = zq;
      if (store) {
        const long o = (long)b * a.Dp + d0;
        *reinterpret_cast<float4*>(a.z + o) = zq;
        if (a.sig) {
          *reinterpret_cast<float4*>(a.sig + o) = sq;
          *reinterpret_cast<float4*>(a.eps + o) = eq;
        }
      }
    }
    if (store && a.kl && idx < ((total + 63) & ~63)) {   // the dq lanes of a cell are adjacent (dq a power of two <= 16)
      for (int off = 1; off < dq; off <<= 1) kl += __shfl_xor(kl, off, 64);
      if (idx < total && (idx & dmask) == 0) a.kl[b] = kl;
    }
  }
  SMX_STAMP(1, 9);   // sample + KL computed, LDS / global stores issued
}

template <int RPT, int FRONT>
__device__ inline void bn_act_fwd_body(const BnFwdArgs& a, const int bid) {
  constexpr bool SMALL = RPT > 0;
  constexpr int BN_RPT = SMALL ? RPT : BN_RPT_DEFAULT;
  extern __shared__ __attribute__((aligned(16))) float zs[];   // FRONT: [B][Dp + 4] | this workgroup's columns of W [8][Dp + 4]
  if (bid >= a.Hp / BN_COLS) {
    // FRONT = 1: ONE extra workgroup leaves z / sigma / eps / KL in memory for the backward pass and does nothing else -- as a duty of column
    // block 0 the stores and the KL arithmetic made that workgroup the launch's longest
    if constexpr (FRONT == 1) latent_tile_to_lds<BN_RPT * 2>(a.lat, zs, true);
    else noise_fill(a, bid - a.Hp / BN_COLS);
    return;
  }
  __shared__ float sh[BN_WAVES * BN_COLS];
  SMX_STAMP(FRONT ? 1 : 0, 0);   // entry
  if (FRONT) preload(a.lat.lat, a.lat.ld, a.lat.Dp, a.lat.D, a.lat.B, a.lat.stochastic, a.lat.inj_eps, a.lat.inj_ld, a.lat.rows, a.lat.z, a.lat.sig, a.lat.eps,
                     a.lat.kl, a.W, a.ldw, a.B, a.H, a.Hp, a.gamma, a.beta, a.inj_mask, a.inj_ld, a.batchnorm, a.training);
  else preload(a.pre, a.n_slabs, a.slab_stride, a.ld, a.B, a.H, a.Hp, a.gamma, a.beta, a.bias, a.inj_mask, a.inj_ld, a.xhat, a.out, a.batchnorm, a.training);
  const int c = threadIdx.x % BN_COLS, rl = threadIdx.x / BN_COLS;
  const int col = bid * BN_COLS + c;
  const bool live = col < a.H;  // padded columns produce zeros
  const float bias = (!a.batchnorm && a.bias && live) ? a.bias[col] : 0.f;
  const float gamma_pre = (a.batchnorm && live) ? a.gamma[col] : 0.f, beta_pre = (a.batchnorm && live) ? a.beta[col] : 0.f;   // (requested ahead of pass 1)
  // ... and the moving statistics the column's first thread updates at the end: read there, they were a memory round trip of their own between
  // the last reduction and the thread's exit -- the workgroup's life
  float mm_pre = 0.f, mv_pre = 0.f;
  if (a.batchnorm && a.training && a.update_moving && live && rl == 0) { mm_pre = a.moving_mean[col]; mv_pre = a.moving_var[col]; }
  constexpr int CH = BN_RL * BN_RPT;
  float vreg[BN_RPT];
  // FRONT = 1: input tile up to 64 wide; FRONT = 2: exactly 128 wide (hidden -> hidden layers of 128-unit networks: the
  // column of W costs 128 registers, fine at one workgroup per CU)
  constexpr int FK = FRONT == 2 ? 128 : 64;
  float wcol[FRONT == 1 ? 64 : 1];   // FRONT = 2 reads its column of W from LDS inside the dot products (128 registers spilled)
  const float* wcol_lds = nullptr;
  if (FRONT) {
    // this workgroup's [Dp][8] tile of W: one coalesced pass, left in LDS TRANSPOSED ([8 columns][Dp + 4]: a thread's column is a run of
    // 16-byte reads; as [Dp][8] it was Dp 4-byte reads, and the input tile's rows -- stride Dp + 1 -- one 4-byte read per multiply-add)
    const int lds_ld = a.lat.Dp + 4;
    float* ws = zs + a.B * lds_ld;
    float wl[FRONT == 2 ? 2 : 1];
#pragma unroll
    for (int u = 0; u < (FRONT == 2 ? 2 : 1); ++u) {
      const int t = (int)threadIdx.x + u * BN_THREADS;
      wl[u] = (t < a.lat.Dp * BN_COLS) ? a.W[(long)(t / BN_COLS) * a.ldw + bid * BN_COLS + (t % BN_COLS)] : 0.f;
    }
    if constexpr (FRONT == 2) {   // a plain input tile (hidden layers): every load first, then LDS
      constexpr int MAXIT = BN_RPT * 4;
      const int kq = a.lat.Dp >> 2, ldz = lds_ld, total = a.B * kq;
      const int ksh = __builtin_ctz((unsigned)kq), kmask = kq - 1;   // (Dp = 128)
      float4 t4[MAXIT];
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int idx = threadIdx.x + it * BN_THREADS;
        t4[it] = idx < total ? *reinterpret_cast<const float4*>(a.lat.lat + (long)(idx >> ksh) * a.lat.ld + (idx & kmask) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int idx = threadIdx.x + it * BN_THREADS;
        if (idx < total) {
          *reinterpret_cast<float4*>(zs + (idx >> ksh) * ldz + (idx & kmask) * 4) = t4[it];
        }
      }
    } else {
      latent_tile_to_lds<BN_RPT * 2>(a.lat, zs, false);   // B Dp / 4 quads over 512 threads: <= 2 RPT iterations (the extra workgroup stores)
    }
#pragma unroll
    for (int u = 0; u < (FRONT == 2 ? 2 : 1); ++u) {
      const int t = (int)threadIdx.x + u * BN_THREADS;
      if (t < a.lat.Dp * BN_COLS) ws[(t % BN_COLS) * lds_ld + t / BN_COLS] = wl[u];
    }
    __syncthreads();
    SMX_STAMP(1, 1);   // latent tile (sample + KL) and the W tile are in LDS
    if constexpr (FRONT == 1) {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const float4 w4 = (4 * v < a.lat.Dp) ? *reinterpret_cast<const float4*>(ws + c * lds_ld + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
        wcol[4 * v] = w4.x; wcol[4 * v + 1] = w4.y; wcol[4 * v + 2] = w4.z; wcol[4 * v + 3] = w4.w;
      }
    } else {
      wcol_lds = ws + c * lds_ld;
    }
    SMX_STAMP(1, 2);   // the thread's column of W in registers
  }

  // dropout multipliers drawn ahead by an earlier launch (or injected): loaded now, used after the reductions
  const bool drop = a.training && a.drop_p > 0.f;
  float mpre[BN_RPT];
  if (SMALL && drop && a.inj_mask) {
#pragma unroll
    for (int i = 0; i < BN_RPT; ++i) {
      const int r = rl + BN_RL * i;
      mpre[i] = r < a.B ? a.inj_mask[(long)r * a.inj_ld + col] : 0.f;
    }
  }
  // pass 1: slab sum (+ bias), column sum
  float s1 = 0.f;
  for (int r0 = 0; r0 < a.B; r0 += CH) {
    float acc[BN_RPT];
    if (FRONT) {
      const int ldz = a.lat.Dp + 4;
#pragma unroll
      for (int i = 0; i < BN_RPT; ++i) {
        const int r = min(r0 + rl + BN_RL * i, a.B - 1);
        const float4* z4 = reinterpret_cast<const float4*>(zs + r * ldz);
        float t = 0.f;
        if (FRONT == 2) {
#pragma unroll 4
          for (int v = 0; v < FK / 4; ++v) {
            const float4 z = z4[v], w = *reinterpret_cast<const float4*>(wcol_lds + 4 * v);
            t = fmaf(z.x, w.x, t); t = fmaf(z.y, w.y, t); t = fmaf(z.z, w.z, t); t = fmaf(z.w, w.w, t);
          }
        } else if (a.lat.Dp <= 32) {
#pragma unroll
          for (int v = 0; v < 8; ++v) {
            const float4 z = z4[v];
            t = fmaf(z.x, wcol[4 * v], t); t = fmaf(z.y, wcol[4 * v + 1], t); t = fmaf(z.z, wcol[4 * v + 2], t); t = fmaf(z.w, wcol[4 * v + 3], t);
          }
        } else {
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const float4 z = z4[v];
            t = fmaf(z.x, wcol[4 * v], t); t = fmaf(z.y, wcol[4 * v + 1], t); t = fmaf(z.z, wcol[4 * v + 2], t); t = fmaf(z.w, wcol[4 * v + 3], t);
          }
        }
        acc[i] = t;
      }
    } else
    slab_sum(a.pre, a.n_slabs, a.slab_stride, a.ld, col, r0, rl, a.B, acc);
#pragma unroll
    for (int i = 0; i < BN_RPT; ++i) {
      const int r = r0 + rl + BN_RL * i;
      const float v = acc[i] + bias;
      if (SMALL) vreg[i] = v;
      if (r < a.B) {
        if (!SMALL) a.xhat[(long)r * a.Hp + col] = v;
        s1 += v;
      }
    }
  }
  float mean = 0.f, inv = 1.f, gamma = 1.f, beta = 0.f;
  SMX_STAMP(FRONT ? 1 : 0, 3);   // pass 1: slab sum / dot products
  if (a.batchnorm) {
    gamma = gamma_pre;
    beta = beta_pre;
    float var;
    if (a.training) {
      s1 = bn_col_reduce(s1, sh);
      SMX_STAMP(FRONT ? 1 : 0, 4);   // column sums
      mean = s1 / (float)a.B;
      float s2 = 0.f;
      if (SMALL) {
#pragma unroll
        for (int i = 0; i < BN_RPT; ++i)
          if (rl + BN_RL * i < a.B) { const float d = vreg[i] - mean; s2 = __builtin_fmaf(d, d, s2); }
      } else {
        for (int r = rl; r < a.B; r += BN_RL) {
          const float d = a.xhat[(long)r * a.Hp + col] - mean;
          s2 = __builtin_fmaf(d, d, s2);
        }
      }
      s2 = bn_col_reduce(s2, sh);
      SMX_STAMP(FRONT ? 1 : 0, 5);   // column variances
      var = s2 / (float)a.B;
      if (rl == 0) {
        if (a.batch_mean) { a.batch_mean[col] = mean; a.batch_var[col] = var; }
        if (a.update_moving && live) {
          // (two products and a sum each, NOT fused: the arithmetic of every build so far -- round 5's compiler paired the two updates
          // into packed multiplies and a packed add; spelled out since the library is built without that pairing, sisua_amd/build.py)
#pragma clang fp contract(off)
          a.moving_mean[col] = mm_pre * a.momentum + mean * (1.f - a.momentum);
          a.moving_var[col] = mv_pre * a.momentum + var * (1.f - a.momentum);
        }
      }
    } else {
      mean = live ? a.moving_mean[col] : 0.f;
      var = live ? a.moving_var[col] : 1.f;
    }
    inv = rsqrtf(var + a.eps);
    if (rl == 0 && a.inv_std) a.inv_std[col] = inv;
  }
  const float scale = drop ? 1.f / (1.f - a.drop_p) : 1.f;
  auto finish = [&](int r, float v, float mahead) {
#pragma clang fp contract(off)
    const long o = (long)r * a.Hp + col;
    float y = v;
    if (a.batchnorm) {
      v = (v - mean) * inv;
      y = __builtin_fmaf(gamma, v, beta);
    }
    if (a.batchnorm || SMALL) a.xhat[o] = v;
    float h = fmaxf(y, 0.f);
    if (a.leak != 0.f) h += a.leak * fminf(y, 0.f);
    if (drop) {
      float mult;
      if (a.inj_mask) mult = SMALL ? mahead : a.inj_mask[(long)r * a.inj_ld + col];
      else {
        const uint32_t cell = a.cell_base + (uint32_t)(a.rows ? a.rows[r] : r);
        const U4 w = philox_block(a.nk, cell, (uint32_t)(col >> 2));
        mult = dropout_mult1(w, col & 3, a.drop_p, scale);
      }
      h *= mult;
    }
    a.out[o] = live ? h : 0.f;
  };
  if (SMALL) {
#pragma unroll
    for (int i = 0; i < BN_RPT; ++i)
      if (rl + BN_RL * i < a.B) finish(rl + BN_RL * i, vreg[i], mpre[i]);
  } else {
    for (int r = rl; r < a.B; r += BN_RL) finish(r, a.xhat[(long)r * a.Hp + col], 0.f);
  }
  SMX_STAMP(FRONT ? 1 : 0, 6);   // normalise, ReLU, dropout, stores issued
}

template <int RPT, int FRONT = 0>
__global__ __launch_bounds__(BN_THREADS) void bn_act_fwd_kernel(BnFwdArgs a) { bn_act_fwd_body<RPT, FRONT>(a, (int)blockIdx.x); }
// two independent layers over the same minibatch in ONE launch (scvi: first layers of the encoder and of the library
// encoder): blocks [0, na) belong to a (its column blocks, then its noise jobs), the rest to b
template <int RPT>
__global__ __launch_bounds__(BN_THREADS) void bn_act_fwd_dual_kernel(BnFwdArgs a, BnFwdArgs b, int na) {
  if ((int)blockIdx.x < na) bn_act_fwd_body<RPT, 0>(a, (int)blockIdx.x);
  else bn_act_fwd_body<RPT, 0>(b, (int)blockIdx.x - na);
}


__global__ void bn_wide_fwd_kernel(BnFwdArgs a);   // (below: the forms that sum a wide panel's column-major slabs themselves)
__global__ void bn_wide_bwd_kernel(BnBwdArgs a);

bool bn_front_supported(int B, int Dp) {
  const int dq = Dp >> 2;
  return B > 0 && B <= BN_RL * 4 && Dp >= 4 && (Dp <= 64 || Dp == 128) && (Dp % 4) == 0 && (dq & (dq - 1)) == 0 && ((size_t)B * (Dp + 4) + (size_t)8 * (Dp + 4)) * sizeof(float) <= 96 * 1024;
}

int launch_bn_act_fwd(hipStream_t st, const BnFwdArgs& a_in) {
  BnFwdArgs a = a_in;
  if (a.front) {
    if (!bn_front_supported(a.B, a.lat.Dp) || a.n_jobs || !a.W || (a.lat.ld % 4) || (a.lat.inj_eps && (a.lat.inj_ld % 4)) ||
        (a.lat.Dp > 32 && a.lat.Dp != 64 && a.lat.Dp != 128)) {
      set_error("bn_act_fwd: latent front not applicable");
      return SMX_ERR_INVALID;
    }
    if (a.Hp % BN_COLS) { set_error("bn_act_fwd: bad shapes"); return SMX_ERR_INVALID; }
    const size_t lds = ((size_t)a.B * (a.lat.Dp + 4) + (size_t)BN_COLS * (a.lat.Dp + 4)) * sizeof(float);
    static const bool big_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_act_fwd_kernel<4, 1>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
    if (lds > 64 * 1024 && !big_ok) { set_error("bn_act_fwd: cannot reserve the dynamic LDS of the latent front"); return SMX_ERR_HIP; }
    if (a.lat.Dp == 128) {   // (B <= 128 by the LDS bound of bn_front_supported)
      static const bool big2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_act_fwd_kernel<2, 2>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
      if (a.B > BN_RL * 2 || a.lat.stochastic || a.lat.relu || a.lat.z || (lds > 64 * 1024 && !big2)) {   // (plain input tiles only)
        set_error("bn_act_fwd: 128-wide front not applicable");
        return SMX_ERR_INVALID;
      }
      hipLaunchKernelGGL((bn_act_fwd_kernel<2, 2>), dim3(a.Hp / BN_COLS), dim3(BN_THREADS), lds, st, a);
    } else if (a.B <= BN_RL * 2) hipLaunchKernelGGL((bn_act_fwd_kernel<2, 1>), dim3(a.Hp / BN_COLS + (a.lat.z ? 1 : 0)), dim3(BN_THREADS), lds, st, a);
    else hipLaunchKernelGGL((bn_act_fwd_kernel<4, 1>), dim3(a.Hp / BN_COLS + (a.lat.z ? 1 : 0)), dim3(BN_THREADS), lds, st, a);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
  }
  if (a.Hp % BN_COLS || a.B <= 0) { set_error("bn_act_fwd: bad shapes"); return SMX_ERR_INVALID; }
  if (a.wide) {
    if (a.B > 128 || a.Hp > 128 || !a.pre || !a.xhat || a.slab_stride < (long)a.Hp * 128 || (a.slab_stride % 4)) { set_error("bn_act_fwd: wide slabs take at most 128 x 128"); return SMX_ERR_INVALID; }
    hipLaunchKernelGGL(bn_wide_fwd_kernel, dim3(a.Hp + a.n_jobs * SMX_NOISE_BLOCKS_PER_JOB), dim3(BN_THREADS), 0, st, a);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
  }
  const int grid = a.Hp / BN_COLS + a.n_jobs * SMX_NOISE_BLOCKS_PER_JOB;
  if (a.B <= BN_RL * 2) hipLaunchKernelGGL(bn_act_fwd_kernel<2>, dim3(grid), dim3(BN_THREADS), 0, st, a);
  else if (a.B <= BN_RL * 4) hipLaunchKernelGGL(bn_act_fwd_kernel<4>, dim3(grid), dim3(BN_THREADS), 0, st, a);
  else if (a.B <= BN_RL * 8) hipLaunchKernelGGL(bn_act_fwd_kernel<8>, dim3(grid), dim3(BN_THREADS), 0, st, a);
  else if (a.B <= BN_RL * 16) hipLaunchKernelGGL(bn_act_fwd_kernel<16>, dim3(grid), dim3(BN_THREADS), 0, st, a);
  else hipLaunchKernelGGL(bn_act_fwd_kernel<0>, dim3(grid), dim3(BN_THREADS), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// two layers over the same minibatch, one launch (no latent front, no SyncBatchNorm; register-resident forms only)
bool bn_dual_supported(int B) { return B > 0 && B <= BN_RL * 4; }
int launch_bn_act_fwd_dual(hipStream_t st, const BnFwdArgs& a, const BnFwdArgs& b) {
  if (a.front || b.front || b.n_jobs || a.B != b.B || !bn_dual_supported(a.B) || a.Hp % BN_COLS || b.Hp % BN_COLS) {
    set_error("bn_act_fwd_dual: bad shapes");
    return SMX_ERR_INVALID;
  }
  const int na = a.Hp / BN_COLS + a.n_jobs * SMX_NOISE_BLOCKS_PER_JOB;
  const int grid = na + b.Hp / BN_COLS;
  if (a.B <= BN_RL * 2) hipLaunchKernelGGL(bn_act_fwd_dual_kernel<2>, dim3(grid), dim3(BN_THREADS), 0, st, a, b, na);
  else hipLaunchKernelGGL(bn_act_fwd_dual_kernel<4>, dim3(grid), dim3(BN_THREADS), 0, st, a, b, na);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

__device__ inline void metrics_body(const MetricsArgs& a);
__device__ inline void sq_reduce_body(const float* sl, int cnt, float* dst);

// ---- fold_dz: d lat [B][64 + 4] into `tile` (LDS), computed by the workgroup itself (BnBwdArgs::fold_dz) ------------------------------------
// 8 waves; wave w owns the cells 16 w .. 16 w + 15.  d z [128 x 32] = zD [128 x 128] zW^T as v_mfma_f32_16x16x32_bf16 on three-way split
// operands (six of the nine cross products: f32 accuracy, smx_device.h): A = the wave's rows of zD straight from global memory in the
// operand's layout (lane: row lane & 15, eight consecutive k from 8 (lane >> 4)), B = zW's bf16 x 3 image in LDS (split once per
// workgroup: 32 x 128 values over 512 threads).  The latent head's backward runs on the accumulators in place; its
// operand loads (mu, s_raw, sigma, eps) were requested at entry.  `wimg`: 3 x 32 rows of 136 bf16.
typedef float bnf_f32x4 __attribute__((ext_vector_type(4)));
__device__ inline bnf_f32x4 bnf_mfma16x3(const Split8& a, const Split8& b, bnf_f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t2, b.t0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t0, b.t2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t1, b.t1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t1, b.t0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t0, b.t1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t0, b.t0, acc, 0, 0, 0);
  return acc;
}
#define SMX_FOLD_WROW 136   /* bf16 per row of the W image: 128 + 8 (rows 272 bytes apart: 16-byte reads of 16 rows spread over the banks) */
#define SMX_FOLD_WIMG_BYTES (3 * 32 * SMX_FOLD_WROW * 2)
__device__ inline void fold_dz_tile(const BnBwdArgs& a, const int bid, float* tile, const int ldd, unsigned char* wimg) {
  const EpiLatentBwd& e = a.zlb;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, kg = lane >> 4;
  // (1) what the latent backward reads at the accumulators' positions -- cell 16 w + 4 kg + r, latent dim 16 nb + li -- requested first (64-byte runs per
  // 16 lanes; the same unconditional loads from clamped rows as gemm_body's EPI = 2)
  float e_mu[2][4], e_sr[2][4], e_sg[2][4], e_ep[2][4];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long b = min(16 * w + 4 * kg + r, a.B - 1);
      const int d = 16 * nb + li;
      e_mu[nb][r] = e.lat[b * e.ld + d];
      e_sr[nb][r] = e.lat[b * e.ld + e.Dp + d];
      e_sg[nb][r] = e.sig[b * e.Dp + d];
      e_ep[nb][r] = e.eps[b * e.Dp + d];
    }
  // (2) zW -> bf16 x 3 image: thread -> row d = tid >> 4, k = 8 (tid & 15) .. + 7
  {
    const int d = tid >> 4, k0 = (tid & 15) * 8;
    const float4 lo = *reinterpret_cast<const float4*>(a.zW + (long)d * a.zldw + k0), hi = *reinterpret_cast<const float4*>(a.zW + (long)d * a.zldw + k0 + 4);
    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    const Split8 sp = split3x8(x);
    smx_bf16x8* row = reinterpret_cast<smx_bf16x8*>(wimg + ((long)d * SMX_FOLD_WROW + k0) * 2);
    row[0] = sp.t0;
    row[(32 * SMX_FOLD_WROW * 2) / 16] = sp.t1;
    row[(2 * 32 * SMX_FOLD_WROW * 2) / 16] = sp.t2;
  }
  // (3) the wave's rows of zD in the A operand's layout
  Split8 A[4];
  {
    const float* zr = a.zD + (long)min(16 * w + li, a.B - 1) * a.zld + 8 * kg;
    float4 v[8];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { v[2 * ks] = *reinterpret_cast<const float4*>(zr + 32 * ks); v[2 * ks + 1] = *reinterpret_cast<const float4*>(zr + 32 * ks + 4); }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float x[8] = {v[2 * ks].x, v[2 * ks].y, v[2 * ks].z, v[2 * ks].w, v[2 * ks + 1].x, v[2 * ks + 1].y, v[2 * ks + 1].z, v[2 * ks + 1].w};
      A[ks] = split3x8(x);
    }
  }
  __syncthreads();
  // (4) d z: two 16 x 16 tiles per wave (latent dims 0-15, 16-31), K = 128 in four steps
  bnf_f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const smx_bf16x8* row = reinterpret_cast<const smx_bf16x8*>(wimg + ((long)(16 * nb + li) * SMX_FOLD_WROW + 32 * ks + 8 * kg) * 2);
      Split8 Bq;
      Bq.t0 = row[0]; Bq.t1 = row[(32 * SMX_FOLD_WROW * 2) / 16]; Bq.t2 = row[(2 * 32 * SMX_FOLD_WROW * 2) / 16];
      acc[nb] = bnf_mfma16x3(A[ks], Bq, acc[nb]);
    }
  // (5) the latent head's backward on the accumulators where they are (gemm_body's EPI = 2, the plain stochastic form): d mu | d s_raw into the tile
  // (rows of 16 w .. 16 w + 15: this wave's own) and, from workgroup 0, once to memory for the weight-gradient launch
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = 16 * w + 4 * kg + r, d = 16 * nb + li;
      const bool live = d < e.D;
      const float dz = acc[nb][r];
      const float o0 = live ? dz + e.kl_scale * e_mu[nb][r] : 0.f;
      const float o1 = live ? (dz * e_ep[nb][r] + e.kl_scale * (e_sg[nb][r] - frcp(e_sg[nb][r]))) * sigmoidf(e_sr[nb][r] + SMX_SOFTPLUS_INV_1) : 0.f;
      if (b < a.B) {
        tile[b * ldd + d] = o0;
        tile[b * ldd + e.Dp + d] = o1;
        if (bid == 0) { e.dlat[(long)b * e.ld + d] = o0; e.dlat[(long)b * e.ld + e.Dp + d] = o1; }
      }
    }
}

template <int RPT, int FRONT>
__device__ inline void bn_act_bwd_body(const BnBwdArgs& a, const int bid) {
  constexpr bool SMALL = RPT > 0;
  constexpr int BN_RPT = SMALL ? RPT : BN_RPT_DEFAULT;
  extern __shared__ __attribute__((aligned(16))) float ds[];   // FRONT: d lat tile [B][fK + 4] | this workgroup's rows of W [8][fK + 4]
  {
    const int nb = a.Hp / BN_COLS, extra = bid - nb;
    if (extra >= 0) {
      const int e = extra - (a.with_metrics ? 1 : 0);
      if (e >= 0 && e < a.adam_count) { adam_chunk_body<BN_THREADS>(a.adam, a.adam_first + e); return; }   // optimiser chunks of the heads: every thread of the workgroup
      if (threadIdx.x >= 256) return;                   // the other riders are 256-thread bodies
      if (e < 0) metrics_body(a.metrics);                                                 // ELBO scalars
      else {                                                                              // or only their gradient norms
        const int i = e - a.adam_count;
        sq_reduce_body(a.adam.sq_slots + a.sqr_first[i], a.sqr_n[i], a.sq_total + a.sqr_dst[i]);
      }
      return;
    }
  }
  __shared__ float sh[BN_WAVES * BN_COLS];
  SMX_STAMP(FRONT ? 3 : 2, 0);   // entry
  if (FRONT) preload(a.fD, a.fld, a.fW, a.fldw, a.fK, a.diag, a.out, a.xhat, a.inv_std, a.gamma, a.B, a.H, a.Hp, a.batchnorm, a.training, a.drop_scale);
  else preload(a.dout, a.n_slabs, a.slab_stride, a.ld, a.out, a.xhat, a.inv_std, a.gamma, a.B, a.H, a.Hp, a.batchnorm, a.training, a.drop_scale, a.dpre, a.leak);
  const int c = threadIdx.x % BN_COLS, rl = threadIdx.x / BN_COLS;
  const int col = bid * BN_COLS + c;
  const bool live = col < a.H;
  constexpr int CH = BN_RL * BN_RPT;
  float dyreg[BN_RPT], xhreg[BN_RPT];
  float s1 = 0.f, s2 = 0.f;
  // (register-resident forms) what the activation mask and the BatchNorm formula need of the forward pass, requested FIRST: these loads
  // used to follow the dot products / slab sums -- a memory round trip of their own behind them (tools/c2_stamps.sh)
  float outpre[BN_RPT], xhpre[BN_RPT];
  float gamma_pre = 0.f, inv_pre = 0.f;
  if (SMALL) {
#pragma unroll
    for (int i = 0; i < BN_RPT; ++i) {
      const long o = (long)min(rl + BN_RL * i, a.B - 1) * a.Hp + col;
      outpre[i] = a.out[o];
      xhpre[i] = a.batchnorm ? a.xhat[o] : 0.f;
    }
    if (a.batchnorm) { gamma_pre = live ? a.gamma[col] : 0.f; inv_pre = a.inv_std[col]; }
  }
  constexpr int FK = FRONT == 2 ? 128 : 64;   // FRONT = 2: K = 128 exactly (see bn_act_fwd_body)
  float wrow[FRONT ? FK : 1];
  if (FRONT) {
    const int ldd = a.fK + 4, kq = a.fK >> 2;   // (rows 16-byte aligned: the dot products read the tile four k at a time)
    const int ksh = __builtin_ctz((unsigned)kq), kmask = kq - 1;   // fK is 32, 64 or 128 (bn_bwd_front_supported)
    // this workgroup's 8 rows of W_lat: ONE coalesced pass into LDS (every thread loading its own row from global
    // memory is 8 different cache lines per quarter-wave: 5 us), then each thread copies its row to registers
    float* ws = ds + a.B * ldd;                  // [BN_COLS][fK + 4]
    const int ldw_s = a.fK + 4;
    float4 wl = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool wl_on = (int)threadIdx.x < BN_COLS * kq && !(a.diag & 64);   // (8 rows x fK / 4 <= 256 float4: one per thread)
    if (wl_on) wl = *reinterpret_cast<const float4*>(a.fW + (long)(bid * BN_COLS + ((int)threadIdx.x >> ksh)) * a.fldw + ((int)threadIdx.x & kmask) * 4);
    if (FRONT == 1 && a.fold_dz) {   // (block-uniform) the tile is computed here: d z product + latent backward (fold_dz_tile)
      fold_dz_tile(a, bid, ds, ldd, reinterpret_cast<unsigned char*>(ws + BN_COLS * ldw_s));
      if ((int)threadIdx.x < BN_COLS * kq) *reinterpret_cast<float4*>(&ws[((int)threadIdx.x >> ksh) * ldw_s + ((int)threadIdx.x & kmask) * 4]) = wl;
    } else
    {   // all loads of the tile in flight at once (B fK / 4 float4 over 512 threads), then LDS
      constexpr int MAXIT = BN_RPT * (FK / 32);
      float4 tl[MAXIT];
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int idx = threadIdx.x + it * BN_THREADS;
        tl[it] = (idx < a.B * kq && !(a.diag & 32)) ? *reinterpret_cast<const float4*>(a.fD + (long)(idx >> ksh) * a.fld + (idx & kmask) * 4)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int idx = threadIdx.x + it * BN_THREADS;
        if (idx < a.B * kq) {
          *reinterpret_cast<float4*>(ds + (idx >> ksh) * ldd + (idx & kmask) * 4) = tl[it];
        }
      }
      if ((int)threadIdx.x < BN_COLS * kq) *reinterpret_cast<float4*>(&ws[((int)threadIdx.x >> ksh) * ldw_s + ((int)threadIdx.x & kmask) * 4]) = wl;
    }
    __syncthreads();
    SMX_STAMP(3, 1);   // the gradient tile and the rows of W are in LDS
#pragma unroll
    for (int v = 0; v < FK / 4; ++v) {
      const float4 t = (4 * v < a.fK) ? *reinterpret_cast<const float4*>(&ws[c * ldw_s + 4 * v]) : make_float4(0.f, 0.f, 0.f, 0.f);
      wrow[4 * v] = t.x; wrow[4 * v + 1] = t.y; wrow[4 * v + 2] = t.z; wrow[4 * v + 3] = t.w;
    }
    SMX_STAMP(3, 2);   // the thread's row of W in registers
  }
  for (int r0 = 0; r0 < a.B; r0 += CH) {
    float acc[BN_RPT];
    if (FRONT) {
      const int ldd = a.fK + 4;
#pragma unroll
      for (int i = 0; i < BN_RPT; ++i) {
        const int r = min(r0 + rl + BN_RL * i, a.B - 1);
        const float4* d4 = reinterpret_cast<const float4*>(ds + r * ldd);
        float t = 0.f;
        auto dots = [&](auto nv) {
#pragma unroll
          for (int v = 0; v < decltype(nv)::value; ++v) {
            const float4 d = d4[v];
            t = fmaf(d.x, wrow[4 * v], t); t = fmaf(d.y, wrow[4 * v + 1], t); t = fmaf(d.z, wrow[4 * v + 2], t); t = fmaf(d.w, wrow[4 * v + 3], t);
          }
        };
        if (a.diag & 16) t = ds[r * ldd] + wrow[0] + wrow[63];
        else if (FRONT == 2) dots(std::integral_constant<int, FK / 4>());
        else if (a.fK <= 32) dots(std::integral_constant<int, 8>());
        else dots(std::integral_constant<int, 16>());
        acc[i] = t;
      }
    } else
    slab_sum(a.dout, a.n_slabs, a.slab_stride, a.ld, col, r0, rl, a.B, acc);
#pragma unroll
    for (int i = 0; i < BN_RPT; ++i) {
      const int r = r0 + rl + BN_RL * i;
      float dy = 0.f, xh = 0.f;
      if (r < a.B) {
        const long o = (long)r * a.Hp + col;
        const float ov = SMALL ? outpre[i] : a.out[o];
        dy = (live && ov > 0.f) ? acc[i] * a.drop_scale : 0.f;
        if (a.leak != 0.f && live && !(ov > 0.f)) dy = acc[i] * a.leak;
        if (a.batchnorm) xh = SMALL ? xhpre[i] : a.xhat[o];
        if (!SMALL) a.dpre[o] = dy;
        s1 += dy;
        { // (product, then sum: not fused -- see the moving statistics of bn_act_fwd_body)
#pragma clang fp contract(off)
          s2 += dy * xh;
        }
      }
      if (SMALL) { dyreg[i] = dy; xhreg[i] = xh; }
    }
  }
  SMX_STAMP(FRONT ? 3 : 2, 3);   // slab sum / dot products, activation mask, the loads of out and xhat
  s1 = bn_col_reduce(s1, sh);
  SMX_STAMP(FRONT ? 3 : 2, 4);
  if (!a.batchnorm) {
    if (rl == 0 && a.dbias && live) a.dbias[col] = s1;
    if (SMALL) {
#pragma unroll
      for (int i = 0; i < BN_RPT; ++i)
        if (rl + BN_RL * i < a.B) a.dpre[(long)(rl + BN_RL * i) * a.Hp + col] = dyreg[i];
    }
    return;
  }
  s2 = bn_col_reduce(s2, sh);
  SMX_STAMP(FRONT ? 3 : 2, 5);
  const float gamma = SMALL ? gamma_pre : (live ? a.gamma[col] : 0.f);
  const float inv = SMALL ? inv_pre : a.inv_std[col];
  if (rl == 0) { a.dgamma[col] = live ? s2 : 0.f; a.dbeta[col] = live ? s1 : 0.f; }
  const float invB = 1.f / (float)a.B;
  auto finish = [&](int r, float dy, float xh) {
    float d;
    if (a.training) d = (gamma * inv) * __builtin_fmaf(-__builtin_fmaf(xh, s2, s1), invB, dy);
    else d = dy * gamma * inv;
    a.dpre[(long)r * a.Hp + col] = d;
  };
  if (SMALL) {
#pragma unroll
    for (int i = 0; i < BN_RPT; ++i)
      if (rl + BN_RL * i < a.B) finish(rl + BN_RL * i, dyreg[i], xhreg[i]);
  } else {
    for (int r = rl; r < a.B; r += BN_RL) finish(r, a.dpre[(long)r * a.Hp + col], a.xhat[(long)r * a.Hp + col]);
  }
  SMX_STAMP(FRONT ? 3 : 2, 6);   // stores issued
}

template <int RPT, int FRONT = 0>
__global__ __launch_bounds__(BN_THREADS) void bn_act_bwd_kernel(BnBwdArgs a) { bn_act_bwd_body<RPT, FRONT>(a, (int)blockIdx.x); }
// two independent layers in ONE launch, both with the gradient front (scvi: last layers of the encoder and of the
// library encoder): blocks [0, na) belong to a (column blocks, then its riders), the rest to b (no riders)
template <int RPT>
__global__ __launch_bounds__(BN_THREADS) void bn_act_bwd_dual_kernel(BnBwdArgs a, BnBwdArgs b, int na) {
  if ((int)blockIdx.x < na) bn_act_bwd_body<RPT, 1>(a, (int)blockIdx.x);
  else bn_act_bwd_body<RPT, 1>(b, (int)blockIdx.x - na);
}

// fold_dz: the latent tile is [B][64] = mu | s_raw halves of 32; the d z tile [B][36] and the rows it overwrites live in the tile's own space
bool bn_bwd_fold_supported(int B, int fK, int Dp) { return B > 0 && B <= 128 && fK == 64 && Dp == 32 && !tuning_on("no_fold_dz"); }
bool bn_bwd_front_supported(int B, int K) {
  return B > 0 && B <= BN_RL * 4 && (K == 32 || K == 64 || (K == 128 && B <= BN_RL * 2)) && ((size_t)B * (K + 4) + 8 * (K + 4)) * sizeof(float) <= 96 * 1024;
}

int launch_bn_act_bwd(hipStream_t st, const BnBwdArgs& a_in) {
  BnBwdArgs a = a_in;
  if (a.front) {
    if (!bn_bwd_front_supported(a.B, a.fK) || !a.fD || !a.fW || (a.fld % 4) || (a.fldw % 4) || a.Hp % BN_COLS) {
      set_error("bn_act_bwd: gradient front not applicable");
      return SMX_ERR_INVALID;
    }
    const int grid = a.Hp / BN_COLS + (a.with_metrics ? 1 : 0) + a.adam_count + a.sqr_count;
    size_t lds = ((size_t)a.B * (a.fK + 4) + (size_t)BN_COLS * (a.fK + 4)) * sizeof(float);
    if (a.fold_dz) {
      const EpiLatentBwd& e = a.zlb;
      if (!bn_bwd_fold_supported(a.B, a.fK, e.Dp) || !a.zD || !a.zW || (a.zld % 4) || (a.zldw % 4) || a.zldw < 128 || a.zld < 128 || !e.lat || !e.sig || !e.eps || !e.dlat ||
          !e.stochastic || e.dklz || e.dz_add || (e.ld % 4) || e.ld < 2 * e.Dp || a.B > BN_RL * 2) {
        set_error("bn_act_bwd: fold_dz not applicable");
        return SMX_ERR_INVALID;
      }
      lds += SMX_FOLD_WIMG_BYTES;
      static const bool fold_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_act_bwd_kernel<2, 1>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
      if (lds > 64 * 1024 && !fold_ok) { set_error("bn_act_bwd: cannot reserve the dynamic LDS of fold_dz"); return SMX_ERR_HIP; }
    }
    static const bool big_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_act_bwd_kernel<4, 1>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
    if (lds > 64 * 1024 && !big_ok) { set_error("bn_act_bwd: cannot reserve the dynamic LDS of the gradient front"); return SMX_ERR_HIP; }
    if (a.fK == 128) {
      static const bool big2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_act_bwd_kernel<2, 2>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
      if (lds > 64 * 1024 && !big2) { set_error("bn_act_bwd: cannot reserve the dynamic LDS of the gradient front"); return SMX_ERR_HIP; }
      hipLaunchKernelGGL((bn_act_bwd_kernel<2, 2>), dim3(grid), dim3(BN_THREADS), lds, st, a);
    } else if (a.B <= BN_RL * 2) hipLaunchKernelGGL((bn_act_bwd_kernel<2, 1>), dim3(grid), dim3(BN_THREADS), lds, st, a);
    else hipLaunchKernelGGL((bn_act_bwd_kernel<4, 1>), dim3(grid), dim3(BN_THREADS), lds, st, a);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
  }
  if (a.Hp % BN_COLS || a.B <= 0) { set_error("bn_act_bwd: bad shapes"); return SMX_ERR_INVALID; }
  if (a.wide) {
    if (a.B > 128 || a.Hp > 128 || !a.dout || a.slab_stride < (long)a.Hp * 128 || (a.slab_stride % 4)) { set_error("bn_act_bwd: wide slabs take at most 128 x 128"); return SMX_ERR_INVALID; }
    hipLaunchKernelGGL(bn_wide_bwd_kernel, dim3(a.Hp + (a.with_metrics ? 1 : 0) + a.adam_count + a.sqr_count), dim3(BN_THREADS), 0, st, a);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
  }
  const int grid = a.Hp / BN_COLS + (a.with_metrics ? 1 : 0) + a.adam_count + a.sqr_count;
  if (a.B <= BN_RL * 2) hipLaunchKernelGGL(bn_act_bwd_kernel<2>, dim3(grid), dim3(BN_THREADS), 0, st, a);
  else if (a.B <= BN_RL * 4) hipLaunchKernelGGL(bn_act_bwd_kernel<4>, dim3(grid), dim3(BN_THREADS), 0, st, a);
  else if (a.B <= BN_RL * 8) hipLaunchKernelGGL(bn_act_bwd_kernel<8>, dim3(grid), dim3(BN_THREADS), 0, st, a);
  else if (a.B <= BN_RL * 16) hipLaunchKernelGGL(bn_act_bwd_kernel<16>, dim3(grid), dim3(BN_THREADS), 0, st, a);
  else hipLaunchKernelGGL(bn_act_bwd_kernel<0>, dim3(grid), dim3(BN_THREADS), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// both layers with the gradient front (their incoming gradients as dot products of an LDS tile)
int launch_bn_act_bwd_dual(hipStream_t st, const BnBwdArgs& a_in, const BnBwdArgs& b_in) {
  BnBwdArgs a = a_in, b = b_in;
  a.diag = b.diag = 0;
  auto ok = [](const BnBwdArgs& x) {
    return x.front && x.fK <= 64 && bn_bwd_front_supported(x.B, x.fK) && x.fD && x.fW && !(x.fld % 4) && !(x.fldw % 4) && !(x.Hp % BN_COLS);
  };
  if (!ok(a) || !ok(b) || a.B != b.B || !bn_dual_supported(a.B) || b.with_metrics || b.adam_count || b.sqr_count) {
    set_error("bn_act_bwd_dual: gradient fronts not applicable");
    return SMX_ERR_INVALID;
  }
  const int na = a.Hp / BN_COLS + (a.with_metrics ? 1 : 0) + a.adam_count + a.sqr_count;
  const int grid = na + b.Hp / BN_COLS;
  auto need = [](const BnBwdArgs& x) { return ((size_t)x.B * (x.fK + 4) + (size_t)BN_COLS * (x.fK + 4)) * sizeof(float); };
  const size_t lds = need(a) > need(b) ? need(a) : need(b);
  static const bool big_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_act_bwd_dual_kernel<4>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
  if (lds > 64 * 1024 && !big_ok) { set_error("bn_act_bwd_dual: cannot reserve the dynamic LDS of the gradient fronts"); return SMX_ERR_HIP; }
  if (a.B <= BN_RL * 2) hipLaunchKernelGGL(bn_act_bwd_dual_kernel<2>, dim3(grid), dim3(BN_THREADS), lds, st, a, b, na);
  else hipLaunchKernelGGL(bn_act_bwd_dual_kernel<4>, dim3(grid), dim3(BN_THREADS), lds, st, a, b, na);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// ===========================================================================
// BatchNorm launches that sum the HUNDREDS of slabs of a wide-panel product themselves (BASELINE.json configs[4]: the encoder front's
// 209 K slices, the fused output head's 250 workgroups, each with a [128][128] partial sum).  Until round 5 a reduce launch
// (bigk_reduce_kernel) stood between the product and the 16-workgroup BatchNorm launch: two latency-bound launches (5-6 us + 6-7 us) for
// 14-16 MB of slabs and 64 KB of result.  Here the producers leave their slabs COLUMN-major ([slab][column][128 rows]: a column of a slab
// is 512 contiguous bytes) and ONE workgroup per column sums its column over the slabs -- 16-byte loads, every slab of a thread in flight
// at once -- and finishes the BatchNorm pass on the 128 sums.  The additions keep the order of the launches they replace (thread sg of 16
// sums the slabs sg, sg + 16, ...; the 16 partial sums in sg order; the column statistics as the balanced tree over rows (r, r + 64) of
// bn_col_reduce): bit for bit the same results (tests/test_gpu_configs.py).  Minibatches of at most 128 cells.
// ===========================================================================
__device__ inline void wide_slab_column(const float* part, long slab_stride, int n_slabs, int col, float* sh /*[16][128]*/) {
  const int rq = threadIdx.x & 31, sg = threadIdx.x >> 5;   // rows 4 rq .. + 3; slabs sg, sg + 16, ...
  const float4* p = reinterpret_cast<const float4*>(part) + (long)col * 32 + rq;
  const long s4 = slab_stride >> 2;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z0 = sg; z0 < n_slabs; z0 += 256) {
    float4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[(long)min(z0 + 16 * u, n_slabs - 1) * s4];
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (z0 + 16 * u < n_slabs) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  *reinterpret_cast<float4*>(sh + sg * 128 + 4 * rq) = acc;
}
// the column's value of row r (threads 0 .. 127 after the barrier)
__device__ inline float wide_row_value(const float* sh, int r) {
  float v = sh[r];
#pragma unroll
  for (int u = 1; u < 16; ++u) v += sh[u * 128 + r];
  return v;
}
// sum over the 64 pairs (r, r + 64) held by the lanes of wave 0: the balanced tree of bn_col_reduce
__device__ inline float wide_tree64(float p) { return wave_sum(p); }   // (lane ^ 1, 2, 4, 8 by DPP, 16 and 32 by row / half exchanges: that tree)

// (Which multiply-adds the compiler fuses in bn_act_fwd_body / bn_act_bwd_body<2, 0> was read off their ISA; the kernels below spell the
// same operations out with contraction switched off, so that the two forms stay equal bit for bit.)
__global__ __launch_bounds__(BN_THREADS) void bn_wide_fwd_kernel(BnFwdArgs a) {
#pragma clang fp contract(off)
  const int bid = (int)blockIdx.x;
  if (bid >= a.Hp) { noise_fill(a, bid - a.Hp); return; }
  __shared__ __attribute__((aligned(16))) float sh[16 * 128];
  __shared__ float vs[128], st[2];
  preload(a.pre, a.n_slabs, a.slab_stride, a.B, a.H, a.Hp, a.gamma, a.beta, a.bias, a.inj_mask, a.inj_ld, a.xhat, a.out, a.batchnorm, a.training, a.drop_p,
          a.batch_mean, a.batch_var, a.moving_mean, a.moving_var, a.inv_std, a.update_moving, a.rows, a.leak);   // (one batch: smx_device.h)
  const int col = bid, r = (int)threadIdx.x;
  const bool live = col < a.H, rowt = r < 128, on = rowt && r < a.B;
  const float bias = (!a.batchnorm && a.bias && live) ? a.bias[col] : 0.f;
  const float gamma = (a.batchnorm && live) ? a.gamma[col] : 0.f, beta = (a.batchnorm && live) ? a.beta[col] : 0.f;
  const bool drop = a.training && a.drop_p > 0.f;
  float mpre = 0.f;
  if (drop && a.inj_mask && on) mpre = a.inj_mask[(long)r * a.inj_ld + col];
  float mm_pre = 0.f, mv_pre = 0.f;   // (the moving statistics thread 0 updates at the end, requested now)
  if (a.batchnorm && a.training && a.update_moving && live && r == 0) { mm_pre = a.moving_mean[col]; mv_pre = a.moving_var[col]; }
  SMX_STAMP(0, 0);
  wide_slab_column(a.pre, a.slab_stride, a.n_slabs, col, sh);
  SMX_STAMP(0, 3);   // the thread's slabs summed, partial sums in LDS
  __syncthreads();
  float v = 0.f;
  if (rowt) { v = wide_row_value(sh, r) + bias; vs[r] = v; }
  float mean = 0.f, inv = 1.f;
  if (a.batchnorm) {
    float var;
    if (a.training) {
      __syncthreads();
      if (r < 64) {   // wave 0: rows r and r + 64, summed and squared in the form of bn_act_fwd_body (the same contractions: the same bits)
        const float v0 = vs[r], v1 = vs[r + 64];
        float s1 = 0.f;
        if (r < a.B) s1 += v0;
        if (r + 64 < a.B) s1 += v1;
        s1 = wide_tree64(s1);
        const float mu = s1 / (float)a.B;
        float s2 = 0.f;
        if (r < a.B) { const float d = v0 - mu; s2 = d * d; }
        if (r + 64 < a.B) { const float d = v1 - mu; s2 = __builtin_fmaf(d, d, s2); }
        s2 = wide_tree64(s2);
        if (r == 0) { st[0] = mu; st[1] = s2 / (float)a.B; }
      }
      __syncthreads();
      mean = st[0];
      var = st[1];
      if (r == 0) {
        if (a.batch_mean) { a.batch_mean[col] = mean; a.batch_var[col] = var; }
        if (a.update_moving && live) {
          a.moving_mean[col] = mm_pre * a.momentum + mean * (1.f - a.momentum);
          a.moving_var[col] = mv_pre * a.momentum + var * (1.f - a.momentum);
        }
      }
    } else {
      mean = live ? a.moving_mean[col] : 0.f;
      var = live ? a.moving_var[col] : 1.f;
    }
    inv = rsqrtf(var + a.eps);
    if (r == 0 && a.inv_std) a.inv_std[col] = inv;
  }
  SMX_STAMP(0, 5);   // column statistics
  if (!on) return;
  const float scale = drop ? 1.f / (1.f - a.drop_p) : 1.f;
  const long o = (long)r * a.Hp + col;
  float y = v;
  if (a.batchnorm) {
    v = (v - mean) * inv;
    y = __builtin_fmaf(gamma, v, beta);
  }
  a.xhat[o] = v;
  float h = fmaxf(y, 0.f);
  if (a.leak != 0.f) h += a.leak * fminf(y, 0.f);
  if (drop) {
    float mult = mpre;
    if (!a.inj_mask) {
      const uint32_t cell = a.cell_base + (uint32_t)(a.rows ? a.rows[r] : r);
      const U4 w = philox_block(a.nk, cell, (uint32_t)(col >> 2));
      mult = dropout_mult1(w, col & 3, a.drop_p, scale);
    }
    h *= mult;
  }
  a.out[o] = live ? h : 0.f;
  SMX_STAMP(0, 6);   // stores issued
}

__global__ __launch_bounds__(BN_THREADS) void bn_wide_bwd_kernel(BnBwdArgs a) {
#pragma clang fp contract(off)
  const int bid = (int)blockIdx.x;
  {
    const int extra = bid - a.Hp;
    if (extra >= 0) {   // the riders of bn_act_bwd_body
      const int e = extra - (a.with_metrics ? 1 : 0);
      if (e >= 0 && e < a.adam_count) { adam_chunk_body<BN_THREADS>(a.adam, a.adam_first + e); return; }
      if (threadIdx.x >= 256) return;
      if (e < 0) metrics_body(a.metrics);
      else {
        const int i = e - a.adam_count;
        sq_reduce_body(a.adam.sq_slots + a.sqr_first[i], a.sqr_n[i], a.sq_total + a.sqr_dst[i]);
      }
      return;
    }
  }
  __shared__ __attribute__((aligned(16))) float sh[16 * 128];
  __shared__ float vs[128], xs[128], st[2];
  preload(a.dout, a.n_slabs, a.slab_stride, a.out, a.xhat, a.inv_std, a.gamma, a.B, a.H, a.Hp, a.batchnorm, a.training, a.drop_scale, a.dpre, a.dgamma, a.dbeta);
  const int col = bid, r = (int)threadIdx.x;
  const bool live = col < a.H, rowt = r < 128, on = rowt && r < a.B;
  // what the activation mask and the BatchNorm formula need of the forward pass, requested ahead of the slabs
  float ov = 0.f, xh = 0.f, gamma = 0.f, inv = 0.f;
  if (on) {
    const long o = (long)r * a.Hp + col;
    ov = a.out[o];
    if (a.batchnorm) xh = a.xhat[o];
  }
  if (a.batchnorm) { gamma = live ? a.gamma[col] : 0.f; inv = a.inv_std[col]; }
  SMX_STAMP(2, 0);
  wide_slab_column(a.dout, a.slab_stride, a.n_slabs, col, sh);
  SMX_STAMP(2, 3);
  __syncthreads();
  float dy = 0.f;
  if (on) {
    const float acc = wide_row_value(sh, r);
    dy = (live && ov > 0.f) ? acc * a.drop_scale : 0.f;
    if (a.leak != 0.f && live && !(ov > 0.f)) dy = acc * a.leak;
  }
  if (rowt) { vs[r] = dy; xs[r] = xh; }
  __syncthreads();
  if (r < 64) {   // wave 0: rows r and r + 64 in the form of bn_act_bwd_body
    float s1 = 0.f, s2 = 0.f;
    if (r < a.B) { const float d0 = vs[r]; s1 += d0; s2 += d0 * xs[r]; }
    if (r + 64 < a.B) { const float d1 = vs[r + 64]; s1 += d1; s2 += d1 * xs[r + 64]; }
    s1 = wide_tree64(s1);
    if (a.batchnorm) s2 = wide_tree64(s2);
    if (r == 0) { st[0] = s1; st[1] = s2; }
  }
  __syncthreads();
  const float s1 = st[0], s2 = st[1];
  SMX_STAMP(2, 5);
  if (!a.batchnorm) {
    if (r == 0 && a.dbias && live) a.dbias[col] = s1;
    if (on) a.dpre[(long)r * a.Hp + col] = dy;
    return;
  }
  if (r == 0) { a.dgamma[col] = live ? s2 : 0.f; a.dbeta[col] = live ? s1 : 0.f; }
  if (!on) return;
  const float invB = 1.f / (float)a.B;
  float d;
  if (a.training) d = (gamma * inv) * __builtin_fmaf(-__builtin_fmaf(xh, s2, s1), invB, dy);
  else d = dy * gamma * inv;
  a.dpre[(long)r * a.Hp + col] = d;
  SMX_STAMP(2, 6);
}

bool bn_wide_supported(int B, int Hp, int n_slabs) { return B > 0 && B <= 128 && Hp > 0 && Hp <= 128 && n_slabs > 0 && !tuning_on("no_bn_wide"); }

// ===========================================================================
// SyncBatchNorm (opt-in under data parallelism, SURVEY.md 8e caveat i): statistics over the GLOBAL minibatch.
// Each BatchNorm pass is cut in two around one small all-reduce: the first launch leaves this rank's column
// statistics in its own slot of a [world][2][Hp] buffer (zeros in the other slots, so that the sum all-reduce
// is an all-gather), the second combines the slots in rank order -- Chan's pairwise form, no E[x^2] - E[x]^2
// cancellation -- and finishes the pass.  Equal batch sizes on every rank.  Plain row loops: the values make the
// round trip through xhat / dpre (the register-resident forms are for the single collective-free launch).
// ===========================================================================
__global__ __launch_bounds__(BN_THREADS) void bn_sync_stats_fwd_kernel(BnFwdArgs a, BnSyncArgs y) {
  __shared__ float sh[BN_WAVES * BN_COLS];
  const int c = threadIdx.x % BN_COLS, rl = threadIdx.x / BN_COLS;
  const int col = blockIdx.x * BN_COLS + c;
  float s1 = 0.f;
  for (int r0 = 0; r0 < a.B; r0 += BN_RL * 2) {
    float acc[2];
    slab_sum<2>(a.pre, a.n_slabs, a.slab_stride, a.ld, col, r0, rl, a.B, acc);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = r0 + rl + BN_RL * i;
      if (r < a.B) { a.xhat[(long)r * a.Hp + col] = acc[i]; s1 += acc[i]; }
    }
  }
  s1 = bn_col_reduce(s1, sh);
  const float mean = s1 / (float)a.B;
  float s2 = 0.f;
  for (int r = rl; r < a.B; r += BN_RL) { const float d = a.xhat[(long)r * a.Hp + col] - mean; s2 += d * d; }
  s2 = bn_col_reduce(s2, sh);
  if (rl == 0)
    for (int r = 0; r < y.world; ++r) {
      y.gather[((long)r * 2 + 0) * a.Hp + col] = r == y.rank ? mean : 0.f;
      y.gather[((long)r * 2 + 1) * a.Hp + col] = r == y.rank ? s2 : 0.f;
    }
}

__global__ __launch_bounds__(BN_THREADS) void bn_sync_apply_fwd_kernel(BnFwdArgs a, BnSyncArgs y) {
  const int c = threadIdx.x % BN_COLS, rl = threadIdx.x / BN_COLS;
  const int col = blockIdx.x * BN_COLS + c;
  const bool live = col < a.H;
  float mean = 0.f;
  for (int r = 0; r < y.world; ++r) mean += y.gather[((long)r * 2) * a.Hp + col];
  mean /= (float)y.world;
  float m2 = 0.f;
  for (int r = 0; r < y.world; ++r) {
    const float d = y.gather[((long)r * 2) * a.Hp + col] - mean;
    m2 += y.gather[((long)r * 2 + 1) * a.Hp + col] + (float)a.B * d * d;
  }
  const float var = m2 / ((float)a.B * (float)y.world);
  const float inv = rsqrtf(var + a.eps);
  const float gamma = live ? a.gamma[col] : 0.f, beta = live ? a.beta[col] : 0.f;
  if (rl == 0) {
    if (a.batch_mean) { a.batch_mean[col] = mean; a.batch_var[col] = var; }
    if (a.inv_std) a.inv_std[col] = inv;
  }
  const bool drop = a.training && a.drop_p > 0.f;
  const float scale = drop ? 1.f / (1.f - a.drop_p) : 1.f;
  for (int r = rl; r < a.B; r += BN_RL) {
    const long o = (long)r * a.Hp + col;
    const float v = (a.xhat[o] - mean) * inv;
    a.xhat[o] = v;
    float h = fmaxf(gamma * v + beta, 0.f);
    if (drop) {
      float mult;
      if (a.inj_mask) mult = a.inj_mask[(long)r * a.inj_ld + col];
      else {
        const uint32_t cell = a.cell_base + (uint32_t)(a.rows ? a.rows[r] : r);
        mult = dropout_mult1(philox_block(a.nk, cell, (uint32_t)(col >> 2)), col & 3, a.drop_p, scale);
      }
      h *= mult;
    }
    a.out[o] = live ? h : 0.f;
  }
}

__global__ __launch_bounds__(BN_THREADS) void bn_sync_stats_bwd_kernel(BnBwdArgs a, BnSyncArgs y) {
  __shared__ float sh[BN_WAVES * BN_COLS];
  const int c = threadIdx.x % BN_COLS, rl = threadIdx.x / BN_COLS;
  const int col = blockIdx.x * BN_COLS + c;
  const bool live = col < a.H;
  float s1 = 0.f, s2 = 0.f;
  for (int r0 = 0; r0 < a.B; r0 += BN_RL * 2) {
    float acc[2];
    slab_sum<2>(a.dout, a.n_slabs, a.slab_stride, a.ld, col, r0, rl, a.B, acc);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = r0 + rl + BN_RL * i;
      if (r < a.B) {
        const long o = (long)r * a.Hp + col;
        const float dy = (live && a.out[o] > 0.f) ? acc[i] * a.drop_scale : 0.f;
        a.dpre[o] = dy;
        s1 += dy;
        s2 += dy * a.xhat[o];
      }
    }
  }
  s1 = bn_col_reduce(s1, sh);
  s2 = bn_col_reduce(s2, sh);
  if (rl == 0) {
    // this rank's share of the parameter gradients: the gradient all-reduce sums the shares
    a.dgamma[col] = live ? s2 : 0.f;
    a.dbeta[col] = live ? s1 : 0.f;
    for (int r = 0; r < y.world; ++r) {
      y.gather[((long)r * 2 + 0) * a.Hp + col] = r == y.rank ? s1 : 0.f;
      y.gather[((long)r * 2 + 1) * a.Hp + col] = r == y.rank ? s2 : 0.f;
    }
  }
}

__global__ __launch_bounds__(BN_THREADS) void bn_sync_apply_bwd_kernel(BnBwdArgs a, BnSyncArgs y) {
  const int c = threadIdx.x % BN_COLS, rl = threadIdx.x / BN_COLS;
  const int col = blockIdx.x * BN_COLS + c;
  const bool live = col < a.H;
  float s1 = 0.f, s2 = 0.f;
  for (int r = 0; r < y.world; ++r) {
    s1 += y.gather[((long)r * 2 + 0) * a.Hp + col];
    s2 += y.gather[((long)r * 2 + 1) * a.Hp + col];
  }
  const float gamma = live ? a.gamma[col] : 0.f;
  const float inv = a.inv_std[col];
  const float invN = 1.f / ((float)a.B * (float)y.world);
  for (int r = rl; r < a.B; r += BN_RL) {
    const long o = (long)r * a.Hp + col;
    a.dpre[o] = gamma * inv * (a.dpre[o] - invN * (s1 + a.xhat[o] * s2));
  }
}

int launch_bn_sync_fwd(hipStream_t st, const BnFwdArgs& a, const BnSyncArgs& y, int phase) {
  if (a.Hp % BN_COLS || a.B <= 0 || !a.batchnorm || !a.training || y.world < 1 || !y.gather) { set_error("bn_sync_fwd: bad arguments"); return SMX_ERR_INVALID; }
  if (phase == 0) hipLaunchKernelGGL(bn_sync_stats_fwd_kernel, dim3(a.Hp / BN_COLS), dim3(BN_THREADS), 0, st, a, y);
  else hipLaunchKernelGGL(bn_sync_apply_fwd_kernel, dim3(a.Hp / BN_COLS), dim3(BN_THREADS), 0, st, a, y);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}
int launch_bn_sync_bwd(hipStream_t st, const BnBwdArgs& a, const BnSyncArgs& y, int phase) {
  if (a.Hp % BN_COLS || a.B <= 0 || !a.batchnorm || !a.training || y.world < 1 || !y.gather) { set_error("bn_sync_bwd: bad arguments"); return SMX_ERR_INVALID; }
  if (phase == 0) hipLaunchKernelGGL(bn_sync_stats_bwd_kernel, dim3(a.Hp / BN_COLS), dim3(BN_THREADS), 0, st, a, y);
  else hipLaunchKernelGGL(bn_sync_apply_bwd_kernel, dim3(a.Hp / BN_COLS), dim3(BN_THREADS), 0, st, a, y);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// ===========================================================================
// Latent head.  One wave per cell, lanes over latent dims.
// ===========================================================================
// one lane = 4 consecutive latent dims of one cell (one Philox block, 16-byte accesses); the Dp/4 lanes of a
// cell are adjacent, so the KL sum is a short shuffle reduction.  Dp/4 is a power of two <= 64 (Dp in {32, 64,
// 128, 256}); other widths take the scalar kernel below.
__global__ __launch_bounds__(64) void latent_fwd_quad_kernel(LatentArgs a) {
  const int dq = a.Dp >> 2;
  const int idx = blockIdx.x * 64 + threadIdx.x;
  const int b = idx / dq, d0 = (idx % dq) * 4;
  float kl = 0.f;
  if (b < a.B) {
    float zz[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {1.f, 1.f, 1.f, 1.f}, ee[4] = {0.f, 0.f, 0.f, 0.f};
    const float4 m4 = *reinterpret_cast<const float4*>(a.lat + (long)b * a.ld + d0);
    const float mu[4] = {m4.x, m4.y, m4.z, m4.w};
    if (a.stochastic) {
      const float4 s4 = *reinterpret_cast<const float4*>(a.lat + (long)b * a.ld + a.Dp + d0);
      const float sr[4] = {s4.x, s4.y, s4.z, s4.w};
      float4 n4;
      if (a.inj_eps) n4 = *reinterpret_cast<const float4*>(a.inj_eps + (long)b * a.inj_ld + d0);
      else n4 = normal4(philox_block(a.nk, a.cell_base + (uint32_t)(a.rows ? a.rows[b] : b), (uint32_t)(d0 >> 2)));
      const float nn[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (d0 + e < a.D) {
          const float sg = softplusf(sr[e] + SMX_SOFTPLUS_INV_1);
          ss[e] = sg; ee[e] = nn[e];
          zz[e] = mu[e] + sg * nn[e];
          kl += 0.5f * (sg * sg + mu[e] * mu[e] - 1.f - 2.f * flog(sg));
        }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (d0 + e < a.D) zz[e] = a.relu ? fmaxf(mu[e], 0.f) : mu[e];
    }
    const long o = (long)b * a.Dp + d0;
    *reinterpret_cast<float4*>(a.z + o) = make_float4(zz[0], zz[1], zz[2], zz[3]);
    if (a.sig) {
      *reinterpret_cast<float4*>(a.sig + o) = make_float4(ss[0], ss[1], ss[2], ss[3]);
      *reinterpret_cast<float4*>(a.eps + o) = make_float4(ee[0], ee[1], ee[2], ee[3]);
    }
  }
  for (int off = 1; off < dq; off <<= 1) kl += __shfl_xor(kl, off, 64);
  if (b < a.B && (idx % dq) == 0 && a.kl) a.kl[b] = kl;
}

__global__ __launch_bounds__(256) void latent_fwd_kernel(LatentArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  float kl = 0.f;
  const uint32_t cell = a.cell_base + (uint32_t)(a.rows ? a.rows[b] : b);
  for (int d = lane; d < a.Dp; d += 64) {
    const long o = (long)b * a.Dp + d;
    float z = 0.f, sig = 1.f, eps = 0.f;
    if (d < a.D) {
      const float mu = a.lat[(long)b * a.ld + d];
      if (a.stochastic) {
        sig = softplusf(a.lat[(long)b * a.ld + a.Dp + d] + SMX_SOFTPLUS_INV_1);
        if (a.inj_eps) eps = a.inj_eps[(long)b * a.inj_ld + d];
        else {
          const float4 n = normal4(philox_block(a.nk, cell, (uint32_t)(d >> 2)));
          eps = (d & 3) == 0 ? n.x : (d & 3) == 1 ? n.y : (d & 3) == 2 ? n.z : n.w;
        }
        z = mu + sig * eps;
        kl += 0.5f * (sig * sig + mu * mu - 1.f - 2.f * flog(sig));
      } else {
        z = a.relu ? fmaxf(mu, 0.f) : mu;
      }
    }
    a.z[o] = z;
    if (a.sig) { a.sig[o] = sig; a.eps[o] = eps; }
  }
  kl = wave_sum(kl);
  if (lane == 0 && a.kl) a.kl[b] = kl;
}

int launch_latent_fwd(hipStream_t st, const LatentArgs& a) {
  const int dq = a.Dp >> 2;
  if (dq >= 1 && dq <= 64 && (dq & (dq - 1)) == 0 && (a.ld % 4) == 0 && (!a.inj_eps || (a.inj_ld % 4) == 0)) {
    hipLaunchKernelGGL(latent_fwd_quad_kernel, dim3((a.B * dq + 63) / 64), dim3(64), 0, st, a);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
  }
  hipLaunchKernelGGL(latent_fwd_kernel, dim3((a.B + 3) / 4), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// ---- SCALE with a mixture-density POSTERIOR (MixLatArgs): one wave per cell, lane d = latent dimension d ----------------------
__device__ inline void mixlat_softmax(const MixLatArgs& a, const float* lat, int lane, float& logpi, float& pi) {
  const float lg = lane < a.C ? lat[lane] : -3.0e38f;
  const float mx = wave_max(lg);
  const float ex = lane < a.C ? fexp(lg - mx) : 0.f;
  const float se = wave_sum(ex);
  logpi = lg - mx - flog(se);
  pi = ex * frcp(se);
}
__global__ __launch_bounds__(256) void mixlat_fwd_kernel(MixLatArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const float HALF_LOG_2PI = 0.9189385332046727f;
  const float* lat = a.lat + (long)b * a.ld;
  const uint32_t cell = a.cell_base + (uint32_t)(a.rows ? a.rows[b] : b);
  float logpi, pi;
  mixlat_softmax(a, lat, lane, logpi, pi);
  // the component: the first c whose running sum of pi reaches the cell's uniform (sums in component order)
  const float u = u24(philox_block(a.nk_pick, cell, 0u).x);
  int k = 0;
  float run = 0.f;
  for (int c = 0; c < a.C; ++c) {
    run += lane_bcast(pi, c);
    k += (run < u) ? 1 : 0;
  }
  k = min(k, a.C - 1);
  const bool live = lane < a.D;
  float eps = 0.f;
  if (live) {
    if (a.inj_eps) eps = a.inj_eps[(long)b * a.inj_ld + lane];
    else {
      const float4 n = normal4(philox_block(a.nk, cell, (uint32_t)(lane >> 2)));
      eps = (lane & 3) == 0 ? n.x : (lane & 3) == 1 ? n.y : (lane & 3) == 2 ? n.z : n.w;
    }
  }
  float mu[8], sg[8];
  float z = 0.f, mean = 0.f, second = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    mu[c] = 0.f; sg[c] = 1.f;
    if (c < a.C) {   // (uniform)
      if (live) { mu[c] = lat[(1 + c) * a.Dp + lane]; sg[c] = softplusf(lat[(1 + a.C + c) * a.Dp + lane] + SMX_SOFTPLUS_INV_1); }
      const float pc = lane_bcast(pi, c);
      mean += pc * mu[c];
      second += pc * (sg[c] * sg[c] + mu[c] * mu[c]);
      if (c == k) z = mu[c] + sg[c] * eps;
    }
  }
  float comp_mine = -3.0e38f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c < a.C) {
      const float dzm = (z - mu[c]) * frcp(sg[c]);
      const float t = wave_sum(live ? -0.5f * dzm * dzm - flog(sg[c]) - HALF_LOG_2PI : 0.f) + lane_bcast(logpi, c);
      if (lane == c) comp_mine = t;
    }
  }
  const float cmx = wave_max(comp_mine);
  const float log_q = cmx + flog(wave_sum(lane < a.C ? fexp(comp_mine - cmx) : 0.f));
  const float log_p = wave_sum(live ? -0.5f * z * z - HALF_LOG_2PI : 0.f);
  if (lane < 32) a.resp[(long)b * 32 + lane] = lane < a.C ? fexp(comp_mine - log_q) : 0.f;
  if (lane == 0) { a.kl[b] = log_q - log_p; a.pick[b] = k; }
  for (int d = lane; d < a.Dp; d += 64) {   // (d == lane for d < D <= 64)
    const long o = (long)b * a.Dp + d;
    a.z[o] = d < a.D ? z : 0.f;
    a.eps[o] = d < a.D ? eps : 0.f;
    a.zmean[o] = d < a.D ? mean : 0.f;
    a.zstd[o] = d < a.D ? fsqrt(fmaxf(second - mean * mean, 0.f)) : 1.f;
  }
}
// d lat from d z: log q depends on z and on every component's parameters, z on the picked component's (mu_k, sigma_k) only
__global__ __launch_bounds__(256) void mixlat_bwd_kernel(MixLatArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const float* lat = a.lat + (long)b * a.ld;
  float* dl = a.dlat + (long)b * a.ld;
  float logpi, pi;
  mixlat_softmax(a, lat, lane, logpi, pi);
  const float rme = lane < a.C ? a.resp[(long)b * 32 + lane] : 0.f;
  const int k = a.pick[b];
  const bool live = lane < a.D;
  float dz = 0.f;
  if (live) for (int s = 0; s < a.dz_slabs; ++s) dz += a.dz[(long)s * a.dz_slab_stride + (long)b * a.ldz + lane];
  const float z = live ? a.z[(long)b * a.Dp + lane] : 0.f, eps = live ? a.eps[(long)b * a.Dp + lane] : 0.f;
  float mu[8], sg[8], sraw[8];
  float dqz = 0.f;   // d log q / d z_d = -sum_c r_c (z - mu_c) / sigma_c^2
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    mu[c] = 0.f; sg[c] = 1.f; sraw[c] = 0.f;
    if (c < a.C) {
      if (live) { mu[c] = lat[(1 + c) * a.Dp + lane]; sraw[c] = lat[(1 + a.C + c) * a.Dp + lane]; sg[c] = softplusf(sraw[c] + SMX_SOFTPLUS_INV_1); }
      const float is = frcp(sg[c]);
      dqz -= lane_bcast(rme, c) * (z - mu[c]) * is * is;
    }
  }
  const float g = dz + a.kl_scale * (z + dqz);
  for (int d = lane; d < a.Dp; d += 64) dl[d] = d < a.C ? a.kl_scale * (rme - pi) : 0.f;   // plane 0: logits (d == lane)
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c < a.C) {
      const float rc = lane_bcast(rme, c), is = frcp(sg[c]);
      const float dzm = (z - mu[c]) * is;
      float dmu = a.kl_scale * rc * dzm * is, dsg = a.kl_scale * rc * (dzm * dzm - 1.f) * is;
      if (c == k) { dmu += g; dsg += g * eps; }
      for (int d = lane; d < a.Dp; d += 64) {
        dl[(1 + c) * a.Dp + d] = d < a.D ? dmu : 0.f;
        dl[(1 + a.C + c) * a.Dp + d] = d < a.D ? dsg * sigmoidf(sraw[c] + SMX_SOFTPLUS_INV_1) : 0.f;
      }
    }
  }
}
static bool mixlat_ok(const MixLatArgs& a) {
  return a.B > 0 && a.C >= 2 && a.C <= 8 && a.D >= a.C && a.D <= 64 && a.Dp <= 64 && a.ld == (1 + 2 * a.C) * a.Dp && a.lat && a.z && a.eps && a.resp && a.pick;
}
int launch_mixlat_fwd(hipStream_t st, const MixLatArgs& a) {
  if (!mixlat_ok(a) || !a.kl || !a.zmean || !a.zstd) { set_error("mixlat_fwd: bad arguments (2..8 components <= latent_dim <= 64)"); return SMX_ERR_INVALID; }
  hipLaunchKernelGGL(mixlat_fwd_kernel, dim3((a.B + 3) / 4), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}
int launch_mixlat_bwd(hipStream_t st, const MixLatArgs& a) {
  if (!mixlat_ok(a) || !a.dz || !a.dlat || a.dz_slabs < 1) { set_error("mixlat_bwd: bad arguments"); return SMX_ERR_INVALID; }
  hipLaunchKernelGGL(mixlat_bwd_kernel, dim3((a.B + 3) / 4), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

__global__ __launch_bounds__(256) void latent_bwd_kernel(LatentArgs a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.B * a.Dp) return;
  const int b = idx / a.Dp, d = idx % a.Dp;
  float dz = 0.f;
  for (int s = 0; s < a.dz_slabs; ++s) dz += a.dz[(long)s * a.dz_slab_stride + idx];
  if (a.stochastic) {
    float dmu = 0.f, ds = 0.f;
    if (d < a.D) {
      const float mu = a.lat[(long)b * a.ld + d];
      const float sraw = a.lat[(long)b * a.ld + a.Dp + d];
      const float sig = a.sig[idx], eps = a.eps[idx];
      dmu = dz + a.kl_scale * mu;
      ds = (dz * eps + a.kl_scale * (sig - frcp(sig))) * sigmoidf(sraw + SMX_SOFTPLUS_INV_1);
    }
    a.dlat[(long)b * a.ld + d] = dmu;
    a.dlat[(long)b * a.ld + a.Dp + d] = ds;
  } else {
    float g = 0.f;
    if (d < a.D) g = (a.relu && !(a.lat[(long)b * a.ld + d] > 0.f)) ? 0.f : dz;
    a.dlat[(long)b * a.ld + d] = g;
  }
}

int launch_latent_bwd(hipStream_t st, const LatentArgs& a) {
  hipLaunchKernelGGL(latent_bwd_kernel, dim3((a.B * a.Dp + 255) / 256), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// ---- SCALE: Gaussian-mixture prior, one-sample Monte-Carlo KL (scale.py:13-49; Xiong et al. 2019) -------------------
// one wave per cell; lanes over the latent dims; the C (<= 32) components are walked serially (C D ~ 100 terms)
__global__ __launch_bounds__(256) void scale_prior_fwd_kernel(ScalePriorArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const float HALF_LOG_2PI = 0.9189385332046727f;
  // log softmax of the mixture logits (lane c holds logit c)
  const float lg = lane < a.C ? a.logits[lane] : -3.0e38f;
  const float lmx = wave_max(lg);
  const float lse = lmx + flog(wave_sum(lane < a.C ? fexp(lg - lmx) : 0.f));
  float comp_mine = -3.0e38f;          // lane c keeps component c's joint log density
  if (a.D <= 64) {
    // (eight components' parameters requested together: a load pair per component inside the loop was a chain of C dependent
    // memory round trips -- 13 us per launch at 10 components)
    const bool live = lane < a.D;
    const float zd = live ? a.z[(long)b * a.Dp + lane] : 0.f;
    for (int c0 = 0; c0 < a.C; c0 += 8) {
      float sr[8], lc[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool on = live && c0 + k < a.C;
        sr[k] = on ? a.scale_raw[(long)(c0 + k) * a.Dp + lane] : 0.f;
        lc[k] = on ? a.loc[(long)(c0 + k) * a.Dp + lane] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (c0 + k < a.C) {   // (uniform)
          const float s = softplusf(sr[k] + SMX_SOFTPLUS_INV_1);
          const float u = (zd - lc[k]) * frcp(s);
          const float t = wave_sum(live ? -0.5f * u * u - flog(s) - HALF_LOG_2PI : 0.f) + (a.logits[c0 + k] - lse);
          if (lane == c0 + k) comp_mine = t;
        }
      }
    }
  } else
  for (int c = 0; c < a.C; ++c) {
    float t = 0.f;
    for (int d = lane; d < a.D; d += 64) {
      const float s = softplusf(a.scale_raw[(long)c * a.Dp + d] + SMX_SOFTPLUS_INV_1);
      const float u = (a.z[(long)b * a.Dp + d] - a.loc[(long)c * a.Dp + d]) * frcp(s);
      t += -0.5f * u * u - flog(s) - HALF_LOG_2PI;
    }
    t = wave_sum(t) + (a.logits[c] - lse);
    if (lane == c) comp_mine = t;
  }
  const float cmx = wave_max(comp_mine);
  const float log_p = cmx + flog(wave_sum(lane < a.C ? fexp(comp_mine - cmx) : 0.f));
  const float resp = lane < a.C ? fexp(comp_mine - log_p) : 0.f;
  if (lane < 32) a.resp[(long)b * 32 + lane] = resp;
  float lq = 0.f;
  for (int d = lane; d < a.D; d += 64) {
    const float e = a.eps[(long)b * a.Dp + d];
    lq += -0.5f * e * e - flog(a.sig[(long)b * a.Dp + d]) - HALF_LOG_2PI;
  }
  lq = wave_sum(lq);
  if (lane == 0) a.kl[b] = lq - log_p;
  // d(-log p)/dz_d = sum_c resp_c (z_d - m_cd) / s_cd^2
  for (int d = lane; d < a.Dp; d += 64) {
    float g = 0.f;
    const float zd = a.z[(long)b * a.Dp + d];
    for (int c0 = 0; c0 < a.C; c0 += 8) {   // (every lane takes part in the broadcasts; padded dims contribute nothing)
      float sr[8], lc[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool on = d < a.D && c0 + k < a.C;
        sr[k] = on ? a.scale_raw[(long)(c0 + k) * a.Dp + d] : 0.f;
        lc[k] = on ? a.loc[(long)(c0 + k) * a.Dp + d] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (c0 + k < a.C) {
          const float rc = lane_bcast(resp, c0 + k);
          const float s = softplusf(sr[k] + SMX_SOFTPLUS_INV_1);
          g += (d < a.D) ? rc * (zd - lc[k]) * frcp(s * s) : 0.f;
        }
      }
    }
    a.dklz[(long)b * a.Dp + d] = g;
  }
}
// ---- covariance = 'tril' (scale.py:28,35): component c = N(m_c, L_c L_c^T), diag L = softplus(raw) + 1e-5, strict lower triangle raw ----
// One wave per cell, lane p = latent dimension p (D <= 32); the component's factor sits in the wave's LDS tile [D][D + 1].
//   u = L^-1 (z - m) forward substitution, w = L^-T u back substitution;  log N = -1/2 |u|^2 - sum log L_pp - D/2 log 2 pi
//   d(-log p)/dz = sum_c resp_c w_c.  Two sweeps over the components (densities -> responsibilities, then the gradient): C D^2 is small.
// lane p fetches row p of L_c: eight 16-byte loads, all in flight at once (a load per entry, each followed by its LDS store, was a
// chain of D dependent memory round trips per component: 566 us per step at D = 32, C = 10)
struct TrilRow { float4 q[8]; };
__device__ inline TrilRow tril_fetch(const ScalePriorArgs& a, int c, int lane) {
  TrilRow r;
  const float4* row = reinterpret_cast<const float4*>(a.scale_raw + ((long)c * a.D + (lane < a.D ? lane : 0)) * a.Dp);   // (Dp = 32)
#pragma unroll
  for (int k = 0; k < 8; ++k) r.q[k] = row[k];
  return r;
}
__device__ inline void tril_store(const ScalePriorArgs& a, const TrilRow& r, int lane, float* L, float& lpp, float& sg) {
  const int D = a.D, ldl = D + 1;
  lpp = 1.f; sg = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float e[4] = {r.q[k].x, r.q[k].y, r.q[k].z, r.q[k].w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = 4 * k + t;
      float v = e[t];
      if (j == lane) { const SpSg s = softplus_sigmoid(v); v = s.sp + 1e-5f; lpp = v; sg = s.sg; }
      if (lane < D && j < D) L[lane * ldl + j] = j <= lane ? v : 0.f;
    }
  }
}
__device__ inline void tril_load(const ScalePriorArgs& a, int c, int lane, float* L, float& lpp, float& sg) {
  const TrilRow r = tril_fetch(a, c, lane);
  tril_store(a, r, lane, L, lpp, sg);
}
// The two substitutions with the factor in REGISTERS (lane p: row p and column p, read once from the LDS tile) and the pivot
// broadcast by v_readlane: a step is multiply -> readlane -> fused multiply-add.  (Its first form read L from LDS inside the loop and
// broadcast with __shfl = ds_bpermute, two LDS round trips per step: ~330 cycles per step, 9 us per 32-dimensional solve.)
struct TrilRegs { float row[32]; float col[32]; };
__device__ inline void tril_regs(int D, int lane, const float* L, TrilRegs& t) {
  const int ldl = D + 1;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    t.row[j] = (lane < D && j < D) ? L[lane * ldl + j] : 0.f;    // (zero above the diagonal)
    t.col[j] = (lane < D && j < D) ? L[j * ldl + lane] : 0.f;    // L[j][lane]: zero for j < lane
  }
}
__device__ inline void tril_solve(int D, int lane, const TrilRegs& t, float lpp, float r, float& u, float& w) {
  const float inv = frcp(lpp);
  u = 0.f; w = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    if (j < D) {   // (uniform)
      const float uj = lane_bcast(r * inv, j);
      if (lane == j) u = uj;
      else if (lane > j) r -= t.row[j] * uj;
    }
  }
  float s = u;
#pragma unroll
  for (int i = 31; i >= 0; --i) {
    if (i < D) {
      const float wi = lane_bcast(s * inv, i);
      if (lane == i) w = wi;
      else if (lane < i) s -= t.col[i] * wi;
    }
  }
}
// forward: one WORKGROUP per cell, one wave per component (SMX_TRILF_WAVES at a time): a component's substitutions are ~4 000
// dependent instructions of one wave -- ten of them in a row per cell took 97 us, side by side they take one's time
#define SMX_TRILF_WAVES 12
__global__ __launch_bounds__(64 * SMX_TRILF_WAVES) void scale_prior_tril_fwd_kernel(ScalePriorArgs a) {
  extern __shared__ float sm[];   // waves x [D][D + 1] | w of every component [C][D] | joint log density of every component [32]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, b = blockIdx.x;
  const int D = a.D;
  float* L = sm + wv * D * (D + 1);
  float* Wc = sm + SMX_TRILF_WAVES * D * (D + 1);
  float* comp = Wc + a.C * D;
  const float HALF_LOG_2PI = 0.9189385332046727f;
  const float lg = lane < a.C ? a.logits[lane] : -3.0e38f;
  const float lmx = wave_max(lg);
  const float lse = lmx + flog(wave_sum(lane < a.C ? fexp(lg - lmx) : 0.f));
  const float zd = lane < D ? a.z[(long)b * a.Dp + lane] : 0.f;
  for (int c = wv; c < a.C; c += SMX_TRILF_WAVES) {
    float lpp, sg, u, w;
    tril_load(a, c, lane, L, lpp, sg);
    __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
    TrilRegs tr;
    tril_regs(D, lane, L, tr);
    tril_solve(D, lane, tr, lpp, zd - (lane < D ? a.loc[(long)c * a.Dp + lane] : 0.f), u, w);
    if (lane < D) Wc[c * D + lane] = w;
    const float t = wave_sum(lane < D ? -0.5f * u * u - flog(lpp) - HALF_LOG_2PI : 0.f) + (a.logits[c] - lse);
    if (lane == 0) comp[c] = t;
    __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();   // (the tile is rewritten by this wave's next component)
  }
  __syncthreads();
  if (wv != 0) return;
  const float comp_mine = lane < a.C ? comp[lane] : -3.0e38f;
  const float cmx = wave_max(comp_mine);
  const float log_p = cmx + flog(wave_sum(lane < a.C ? fexp(comp_mine - cmx) : 0.f));
  const float resp = lane < a.C ? fexp(comp_mine - log_p) : 0.f;
  if (lane < 32) a.resp[(long)b * 32 + lane] = resp;
  float lq = 0.f;
  if (lane < D) {
    const float e = a.eps[(long)b * a.Dp + lane];
    lq = -0.5f * e * e - flog(a.sig[(long)b * a.Dp + lane]) - HALF_LOG_2PI;
  }
  lq = wave_sum(lq);
  if (lane == 0) a.kl[b] = lq - log_p;
  float g = 0.f;   // d(-log p)/dz = sum_c resp_c w_c
  for (int c = 0; c < a.C; ++c) g += lane_bcast(resp, c) * (lane < D ? Wc[c * D + lane] : 0.f);
  for (int d = lane; d < a.Dp; d += 64) a.dklz[(long)b * a.Dp + d] = d < D ? g : 0.f;   // (d == lane for d < D <= 32)
}
// gradients of the prior's parameters: a workgroup per (component, group of SMX_TRILB_CELLS cells) -- the factor once in registers,
// waves over the group's cells, lane p = row p -- leaves its partial sums in `part`; scale_prior_tril_reduce_kernel adds the groups
// in order (deterministic) and applies the diagonal's derivative.  (One workgroup per component walking all cells: 99 - 142 us.)
#define SMX_TRILB_WAVES 8
#define SMX_TRILB_CELLS 8
__global__ __launch_bounds__(64 * SMX_TRILB_WAVES) void scale_prior_tril_bwd_kernel(ScalePriorArgs a, float* gpart) {
  extern __shared__ float sm[];   // L [D][D + 1] | partial sums [waves][D][D + 2]
  const int c = blockIdx.x, grp = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int D = a.D, ldl = D + 1, lda = D + 2;
  float* L = sm;
  float* part = sm + D * ldl;
  float lpp = 1.f, sg = 0.f;
  if (wv == 0) tril_load(a, c, lane, L, lpp, sg);
  __syncthreads();
  if (wv != 0 && lane < D) { lpp = L[lane * ldl + lane]; }
  TrilRegs tr;
  tril_regs(D, lane, L, tr);
  float acc[32];   // row p of sum_b r (w u^T); [j = p] also carries the -r / L_pp term
#pragma unroll
  for (int j = 0; j < 32; ++j) acc[j] = 0.f;
  float g_loc = 0.f, g_lg = 0.f;
  const float mloc = lane < D ? a.loc[(long)c * a.Dp + lane] : 0.f;
  const float invl = frcp(lpp);
  const int b_end = min(a.B, (grp + 1) * SMX_TRILB_CELLS);
  for (int b = grp * SMX_TRILB_CELLS + wv; b < b_end; b += SMX_TRILB_WAVES) {
    const float rc = a.resp[(long)b * 32 + c];
    float u, w;
    tril_solve(D, lane, tr, lpp, (lane < D ? a.z[(long)b * a.Dp + lane] : 0.f) - mloc, u, w);
    g_loc += rc * w;
    g_lg += rc;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      if (j < D) {   // (uniform)
        const float uj = lane_bcast(u, j);
        acc[j] += rc * (w * uj - (j == lane ? invl : 0.f));
      }
    }
  }
  if (lane < D) {
#pragma unroll
    for (int j = 0; j < 32; ++j)
      if (j < D) part[(wv * D + lane) * lda + j] = acc[j];
    part[(wv * D + lane) * lda + D] = g_loc;
  }
  if (lane == 0) part[(wv * D) * lda + D + 1] = g_lg;
  __syncthreads();
  // this group's sums over its waves (wave order) -> gpart[c][grp][D][D + 2]
  float* out = gpart + ((long)c * gridDim.y + grp) * D * lda;
  for (int i = threadIdx.x; i < D * lda; i += 64 * SMX_TRILB_WAVES) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < SMX_TRILB_WAVES; ++q) t += part[q * D * lda + i];
    out[i] = t;
  }
}
__global__ __launch_bounds__(256) void scale_prior_tril_reduce_kernel(ScalePriorArgs a, const float* gpart, int n_grp) {
  const int c = blockIdx.x;
  const int D = a.D, lda = D + 2;
  const float* base = gpart + (long)c * n_grp * D * lda;
  for (int i = threadIdx.x; i < D * lda; i += 256) {   // one thread per entry of the component's [D][D + 2] sums, the groups in order
    float t = 0.f;
    for (int q0 = 0; q0 < n_grp; q0 += 8) {   // eight groups' loads in flight, added in group order
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = q0 + k < n_grp ? base[(long)(q0 + k) * D * lda + i] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += v[k];
    }
    const int p = i / lda, j = i - p * lda;
    if (j < D) {
      float g = 0.f;
      if (j < p) g = -a.kl_scale * t;
      else if (j == p) g = -a.kl_scale * t * sigmoidf(a.scale_raw[((long)c * D + p) * a.Dp + p]);
      a.g_scale[((long)c * D + p) * a.Dp + j] = g;
    } else if (j == D) {
      a.g_loc[(long)c * a.Dp + p] = -a.kl_scale * t;
    } else if (p == 0) {   // j == D + 1: the sum of the responsibilities
      float mx = -3.0e38f;
      for (int q = 0; q < a.C; ++q) mx = fmaxf(mx, a.logits[q]);
      float se = 0.f;
      for (int q = 0; q < a.C; ++q) se += fexp(a.logits[q] - mx);
      a.g_logits[c] = a.kl_scale * ((float)a.B * fexp(a.logits[c] - mx) * frcp(se) - t);
    }
  }
  for (int i = threadIdx.x; i < D * (a.Dp - D); i += 256) {   // the padded columns
    const int p = i / (a.Dp - D), j = D + i % (a.Dp - D);
    a.g_scale[((long)c * D + p) * a.Dp + j] = 0.f;
  }
  for (int d = D + (int)threadIdx.x; d < a.Dp; d += 256) a.g_loc[(long)c * a.Dp + d] = 0.f;
}

int launch_scale_prior_fwd(hipStream_t st, const ScalePriorArgs& a) {
  if (a.C < 2 || a.C > 32 || a.B <= 0) { set_error("scale prior: 2..32 components"); return SMX_ERR_INVALID; }
  if (a.tril) {
    if (a.D < 1 || a.D > 32) { set_error("scale prior: full-covariance components take at most 32 latent dimensions"); return SMX_ERR_INVALID; }
    if (a.Dp != 32) { set_error("scale prior: full-covariance components expect a 32-wide padded latent"); return SMX_ERR_INVALID; }
    hipLaunchKernelGGL(scale_prior_tril_fwd_kernel, dim3(a.B), dim3(64 * SMX_TRILF_WAVES), (size_t)(SMX_TRILF_WAVES * a.D * (a.D + 1) + a.C * a.D + 32) * sizeof(float), st, a);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
  }
  hipLaunchKernelGGL(scale_prior_fwd_kernel, dim3((a.B + 3) / 4), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}
// gradients of the prior's parameters: one workgroup per component, lanes over the latent dims, waves over the cells
__global__ __launch_bounds__(256) void scale_prior_bwd_kernel(ScalePriorArgs a) {
  __shared__ float sh[4][3][64];
  const int c = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int d0 = 0; d0 < a.Dp; d0 += 64) {
    const int d = d0 + lane;
    float g_loc = 0.f, g_sc = 0.f, g_lg = 0.f;
    const bool live = d < a.D;
    const float raw = live ? a.scale_raw[(long)c * a.Dp + d] : 0.f;
    const float s = softplusf(raw + SMX_SOFTPLUS_INV_1), m = live ? a.loc[(long)c * a.Dp + d] : 0.f;
    const float is = frcp(s);
    for (int b0 = w; b0 < a.B; b0 += 32) {   // eight of this wave's cells at a time: their loads in flight together
      float rcv[8], zv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int b = b0 + 4 * k;
        rcv[k] = b < a.B ? a.resp[(long)b * 32 + c] : 0.f;
        zv[k] = (b < a.B && live) ? a.z[(long)b * a.Dp + d] : m;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {   // (cell order as before: the sums keep their bits)
        if (b0 + 4 * k < a.B) {
          const float rc = rcv[k];
          if (live) {
            const float u = (zv[k] - m) * is;
            g_loc += rc * u * is;
            g_sc += rc * (u * u - 1.f) * is;
          }
          if (d0 == 0 && lane == 0) g_lg += rc;
        }
      }
    }
    sh[w][0][lane] = g_loc; sh[w][1][lane] = g_sc; sh[w][2][lane] = g_lg;
    __syncthreads();
    if (w == 0) {
      const float t0 = (sh[0][0][lane] + sh[1][0][lane]) + (sh[2][0][lane] + sh[3][0][lane]);
      const float t1 = (sh[0][1][lane] + sh[1][1][lane]) + (sh[2][1][lane] + sh[3][1][lane]);
      if (d < a.Dp) {
        a.g_loc[(long)c * a.Dp + d] = live ? -a.kl_scale * t0 : 0.f;
        a.g_scale[(long)c * a.Dp + d] = live ? -a.kl_scale * t1 * sigmoidf(raw + SMX_SOFTPLUS_INV_1) : 0.f;
      }
      if (d0 == 0 && lane == 0) {
        const float rsum = (sh[0][2][0] + sh[1][2][0]) + (sh[2][2][0] + sh[3][2][0]);
        // softmax(logits)_c * B - sum_b resp_bc
        float mx = -3.0e38f;
        for (int q = 0; q < a.C; ++q) mx = fmaxf(mx, a.logits[q]);
        float se = 0.f;
        for (int q = 0; q < a.C; ++q) se += fexp(a.logits[q] - mx);
        a.g_logits[c] = a.kl_scale * ((float)a.B * fexp(a.logits[c] - mx) * frcp(se) - rsum);
      }
    }
    __syncthreads();
  }
}
// tied mixture parameters (scale.py:29-33): one location / one scale vector shared by every component = a [C][D] tensor whose rows
// are equal and all receive the SUM of the rows' gradients (in component order); fixed uniform weights = no logits gradient
__global__ __launch_bounds__(256) void scale_prior_tie_kernel(ScalePriorArgs a) {
  for (int d = threadIdx.x; d < a.Dp; d += 256) {
    if (a.tie_loc) {
      float t = 0.f;
      for (int c = 0; c < a.C; ++c) t += a.g_loc[(long)c * a.Dp + d];
      for (int c = 0; c < a.C; ++c) a.g_loc[(long)c * a.Dp + d] = t;
    }
    if (a.tie_scale) {
      float t = 0.f;
      for (int c = 0; c < a.C; ++c) t += a.g_scale[(long)c * a.Dp + d];
      for (int c = 0; c < a.C; ++c) a.g_scale[(long)c * a.Dp + d] = t;
    }
  }
  if (a.tie_mixtures && (int)threadIdx.x < a.C) a.g_logits[threadIdx.x] = 0.f;
}

int launch_scale_prior_bwd(hipStream_t st, const ScalePriorArgs& a) {
  if (a.tril) {
    if (a.D < 1 || a.D > 32 || a.tie_mixtures || a.tie_loc || a.tie_scale) { set_error("scale prior: full-covariance components take at most 32 latent dimensions and no tied parameters"); return SMX_ERR_INVALID; }
    const int n_grp = (a.B + SMX_TRILB_CELLS - 1) / SMX_TRILB_CELLS;
    if (!a.tril_part || (size_t)a.C * n_grp * a.D * (a.D + 2) > a.tril_part_floats) { set_error("scale prior: no scratch for the full-covariance gradients"); return SMX_ERR_INVALID; }
    hipLaunchKernelGGL(scale_prior_tril_bwd_kernel, dim3(a.C, n_grp), dim3(64 * SMX_TRILB_WAVES), (size_t)(a.D * (a.D + 1) + SMX_TRILB_WAVES * a.D * (a.D + 2)) * sizeof(float), st, a, a.tril_part);
    hipLaunchKernelGGL(scale_prior_tril_reduce_kernel, dim3(a.C), dim3(256), 0, st, a, a.tril_part, n_grp);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
  }
  hipLaunchKernelGGL(scale_prior_bwd_kernel, dim3(a.C), dim3(256), 0, st, a);
  if (a.tie_mixtures || a.tie_loc || a.tie_scale) hipLaunchKernelGGL(scale_prior_tie_kernel, dim3(1), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// ---- scvi library latent (scvi.py:37-45, 88-106, 117) -------------------------
__global__ void lib_latent_fwd_kernel(LibLatentArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const long src = a.rows ? a.rows[b] : b;
  const float mu = a.latl[(long)b * a.ld], sig = softplusf(a.latl[(long)b * a.ld + 1] + SMX_SOFTPLUS_INV_1);
  float eps;
  if (a.inj_eps) eps = a.inj_eps[(long)b * a.inj_ld];
  else eps = normal4(philox_block(a.nk, a.cell_base + (uint32_t)src, 0u)).x;
  const float mp = a.library[src * 2], vp = a.library[src * 2 + 1];
  const float sp = sqrtf(vp);
  a.l[b] = mu + sig * eps;
  a.sig[b] = sig;
  a.eps[b] = eps;
  a.kl[b] = logf(sp / sig) + (sig * sig + (mu - mp) * (mu - mp)) / (2.f * vp) - 0.5f;
}
int launch_lib_latent_fwd(hipStream_t st, const LibLatentArgs& a) {
  hipLaunchKernelGGL(lib_latent_fwd_kernel, dim3((a.B + 255) / 256), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}
__global__ void lib_latent_bwd_kernel(LibLatentArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const long src = a.rows ? a.rows[b] : b;
  const float mu = a.latl[(long)b * a.ld], sraw = a.latl[(long)b * a.ld + 1];
  const float sig = a.sig[b], eps = a.eps[b];
  const float mp = a.library[src * 2], vp = a.library[src * 2 + 1];
  const float dl = a.dl[b];
  for (int j = 2; j < a.ld; ++j) a.dlatl[(long)b * a.ld + j] = 0.f;
  a.dlatl[(long)b * a.ld] = dl + a.kl_scale * (mu - mp) / vp;
  a.dlatl[(long)b * a.ld + 1] = (dl * eps + a.kl_scale * (sig / vp - 1.f / sig)) * sigmoidf(sraw + SMX_SOFTPLUS_INV_1);
}
int launch_lib_latent_bwd(hipStream_t st, const LibLatentArgs& a) {
  hipLaunchKernelGGL(lib_latent_bwd_kernel, dim3((a.B + 255) / 256), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// ===========================================================================
// scvi head: one workgroup per cell, three sweeps over G (max, sum, write)
// ===========================================================================
__device__ inline float block_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__device__ inline float block_max(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

__global__ __launch_bounds__(256) void scvi_head_fwd_kernel(ScviHeadArgs a) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  const float* raw = a.raw + (long)b * a.ld;
  float* pl = a.planes + (long)b * a.ld;
  float mx = -3.0e38f;
  for (int g = threadIdx.x; g < a.G; g += 256) mx = fmaxf(mx, raw[g]);
  mx = block_max(mx, sh);
  float sum = 0.f;
  for (int g = threadIdx.x; g < a.G; g += 256) sum += fexp(raw[g] - mx);
  sum = block_sum(sum, sh);
  const float inv = 1.f / sum;
  const float el = expf(fminf(fmaxf(a.l[b], 0.f), a.clip_library));
  for (int g = threadIdx.x; g < a.Gp; g += 256) {
    float rho = 0.f, rate = 0.f, th = 0.f, gate = 0.f;
    if (g < a.G) {
      rho = fexp(raw[g] - mx) * inv;
      rate = el * fminf(fmaxf(rho, 1e-7f), 1.f - 1e-7f);
      th = fexp(raw[a.plane_stride + g]);
      if (a.k == 3) gate = raw[2 * a.plane_stride + g];
    }
    a.rho_raw[(long)b * a.Gp + g] = rho;
    pl[g] = rate;
    pl[a.plane_stride + g] = th;
    if (a.k == 3) pl[2 * a.plane_stride + g] = gate;
  }
}
// Register-resident forms for gene panels up to 1024 * NV genes: one read of the raw planes (16-byte accesses),
// the softmax terms stay in registers between the max, the sum and the write (the generic kernels above sweep
// the row three times).
template <int NV>
__global__ __launch_bounds__(256) void scvi_head_fwd_reg_kernel(ScviHeadArgs a) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  const float* raw = a.raw + (long)b * a.ld;
  float* pl = a.planes + (long)b * a.ld;
  float4 r0[NV], r1[NV], r2[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int g = (threadIdx.x + 256 * j) * 4;
    const bool ok = g < a.Gp;
    r0[j] = ok ? *reinterpret_cast<const float4*>(raw + g) : zero4();
    r1[j] = ok ? *reinterpret_cast<const float4*>(raw + a.plane_stride + g) : zero4();
    r2[j] = (ok && a.k == 3) ? *reinterpret_cast<const float4*>(raw + 2 * a.plane_stride + g) : zero4();
  }
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int g = (threadIdx.x + 256 * j) * 4;
    const float v[4] = {r0[j].x, r0[j].y, r0[j].z, r0[j].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) if (g + e < a.G) mx = fmaxf(mx, v[e]);
  }
  mx = block_max(mx, sh);
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
  sum = block_sum(sum, sh);
  const float inv = 1.f / sum;
  const float el = expf(fminf(fmaxf(a.l[b], 0.f), a.clip_library));
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int g = (threadIdx.x + 256 * j) * 4;
    if (g >= a.Gp) continue;
    const float t[4] = {r1[j].x, r1[j].y, r1[j].z, r1[j].w};
    const float gt[4] = {r2[j].x, r2[j].y, r2[j].z, r2[j].w};
    float rho[4], rate[4], th[4], gate[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool live = g + e < a.G;
      rho[e] = live ? ex[j][e] * inv : 0.f;
      rate[e] = live ? el * fminf(fmaxf(rho[e], 1e-7f), 1.f - 1e-7f) : 0.f;
      th[e] = live ? fexp(t[e]) : 0.f;
      gate[e] = live ? gt[e] : 0.f;
    }
    *reinterpret_cast<float4*>(a.rho_raw + (long)b * a.Gp + g) = make_float4(rho[0], rho[1], rho[2], rho[3]);
    *reinterpret_cast<float4*>(pl + g) = make_float4(rate[0], rate[1], rate[2], rate[3]);
    *reinterpret_cast<float4*>(pl + a.plane_stride + g) = make_float4(th[0], th[1], th[2], th[3]);
    if (a.k == 3) *reinterpret_cast<float4*>(pl + 2 * a.plane_stride + g) = make_float4(gate[0], gate[1], gate[2], gate[3]);
  }
}

template <int NV>
__global__ __launch_bounds__(256) void scvi_head_bwd_reg_kernel(ScviHeadArgs a) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  const float* pl = a.planes + (long)b * a.ld;
  const float* dp = a.dplanes + (long)b * a.ld;
  float* dr = a.draw + (long)b * a.ld;
  const float* rho = a.rho_raw + (long)b * a.Gp;
  const float lraw = a.l[b];
  const float el = expf(fminf(fmaxf(lraw, 0.f), a.clip_library));
  float4 rh[NV], d0v[NV], p0v[NV], d1v[NV], p1v[NV], d2v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int g = (threadIdx.x + 256 * j) * 4;
    const bool ok = g < a.Gp;
    rh[j] = ok ? *reinterpret_cast<const float4*>(rho + g) : zero4();
    d0v[j] = ok ? *reinterpret_cast<const float4*>(dp + g) : zero4();
    p0v[j] = ok ? *reinterpret_cast<const float4*>(pl + g) : zero4();
    d1v[j] = ok ? *reinterpret_cast<const float4*>(dp + a.plane_stride + g) : zero4();
    p1v[j] = ok ? *reinterpret_cast<const float4*>(pl + a.plane_stride + g) : zero4();
    d2v[j] = (ok && a.k == 3) ? *reinterpret_cast<const float4*>(dp + 2 * a.plane_stride + g) : zero4();
  }
  float s = 0.f, dlh = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int g = (threadIdx.x + 256 * j) * 4;
    const float r[4] = {rh[j].x, rh[j].y, rh[j].z, rh[j].w};
    const float dd[4] = {d0v[j].x, d0v[j].y, d0v[j].z, d0v[j].w};
    const float pp[4] = {p0v[j].x, p0v[j].y, p0v[j].z, p0v[j].w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (g + e < a.G) {
        const float inside = (r[e] > 1e-7f && r[e] < 1.f - 1e-7f) ? 1.f : 0.f;
        s += dd[e] * el * inside * r[e];
        dlh += dd[e] * pp[e];
      }
  }
  s = block_sum(s, sh);
  dlh = block_sum(dlh, sh);
  if (threadIdx.x == 0) a.dl[b] = (lraw > 0.f && lraw < a.clip_library) ? dlh : 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int g = (threadIdx.x + 256 * j) * 4;
    if (g >= a.Gp) continue;
    const float r[4] = {rh[j].x, rh[j].y, rh[j].z, rh[j].w};
    const float dd[4] = {d0v[j].x, d0v[j].y, d0v[j].z, d0v[j].w};
    const float d1[4] = {d1v[j].x, d1v[j].y, d1v[j].z, d1v[j].w};
    const float p1[4] = {p1v[j].x, p1v[j].y, p1v[j].z, p1v[j].w};
    const float d2[4] = {d2v[j].x, d2v[j].y, d2v[j].z, d2v[j].w};
    float o0[4], o1[4], o2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool live = g + e < a.G;
      const float inside = (r[e] > 1e-7f && r[e] < 1.f - 1e-7f) ? 1.f : 0.f;
      o0[e] = live ? r[e] * (dd[e] * el * inside - s) : 0.f;
      o1[e] = live ? d1[e] * p1[e] : 0.f;
      o2[e] = live ? d2[e] : 0.f;
    }
    *reinterpret_cast<float4*>(dr + g) = make_float4(o0[0], o0[1], o0[2], o0[3]);
    *reinterpret_cast<float4*>(dr + a.plane_stride + g) = make_float4(o1[0], o1[1], o1[2], o1[3]);
    if (a.k == 3) *reinterpret_cast<float4*>(dr + 2 * a.plane_stride + g) = make_float4(o2[0], o2[1], o2[2], o2[3]);
  }
}

// Panels beyond the register forms (more than 8192 genes outside a training step's row-local launch, more than 20 480 inside one): the three
// sweeps of the generic kernels with 1024 threads and 16-byte accesses (round 6: the generic forms walk a row with 256 threads and 4-byte loads --
// 87 + 104 us per step at 20 000 genes).  Same arithmetic per element; the row sums are taken in another order.
__device__ inline float block_sum16(float v, float* sh) {   // 16 waves
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) t += sh[w];
  return t;
}
__device__ inline float block_max16(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = sh[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) t = fmaxf(t, sh[w]);
  return t;
}
__global__ __launch_bounds__(1024) void scvi_head_fwd_vec_kernel(ScviHeadArgs a) {
  __shared__ float sh[16];
  const int b = blockIdx.x;
  const float* raw = a.raw + (long)b * a.ld;
  float* pl = a.planes + (long)b * a.ld;
  float mx = -3.0e38f;
  for (int g = threadIdx.x * 4; g < a.Gp; g += 4096) {
    const float4 v = *reinterpret_cast<const float4*>(raw + g);
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) if (g + e < a.G) mx = fmaxf(mx, x[e]);
  }
  mx = block_max16(mx, sh);
  float sum = 0.f;
  for (int g = threadIdx.x * 4; g < a.Gp; g += 4096) {
    const float4 v = *reinterpret_cast<const float4*>(raw + g);
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) if (g + e < a.G) sum += fexp(x[e] - mx);
  }
  sum = block_sum16(sum, sh);
  const float inv = 1.f / sum;
  const float el = expf(fminf(fmaxf(a.l[b], 0.f), a.clip_library));
  for (int g = threadIdx.x * 4; g < a.Gp; g += 4096) {
    const float4 v0 = *reinterpret_cast<const float4*>(raw + g), v1 = *reinterpret_cast<const float4*>(raw + a.plane_stride + g);
    const float4 v2 = a.k == 3 ? *reinterpret_cast<const float4*>(raw + 2 * a.plane_stride + g) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float x0[4] = {v0.x, v0.y, v0.z, v0.w}, x1[4] = {v1.x, v1.y, v1.z, v1.w}, x2[4] = {v2.x, v2.y, v2.z, v2.w};
    float rho[4], rate[4], th[4], gate[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool live = g + e < a.G;
      rho[e] = live ? fexp(x0[e] - mx) * inv : 0.f;
      rate[e] = live ? el * fminf(fmaxf(rho[e], 1e-7f), 1.f - 1e-7f) : 0.f;
      th[e] = live ? fexp(x1[e]) : 0.f;
      gate[e] = live ? x2[e] : 0.f;
    }
    *reinterpret_cast<float4*>(a.rho_raw + (long)b * a.Gp + g) = make_float4(rho[0], rho[1], rho[2], rho[3]);
    *reinterpret_cast<float4*>(pl + g) = make_float4(rate[0], rate[1], rate[2], rate[3]);
    *reinterpret_cast<float4*>(pl + a.plane_stride + g) = make_float4(th[0], th[1], th[2], th[3]);
    if (a.k == 3) *reinterpret_cast<float4*>(pl + 2 * a.plane_stride + g) = make_float4(gate[0], gate[1], gate[2], gate[3]);
  }
}
__global__ __launch_bounds__(1024) void scvi_head_bwd_vec_kernel(ScviHeadArgs a) {
  __shared__ float sh[16];
  const int b = blockIdx.x;
  const float* pl = a.planes + (long)b * a.ld;
  const float* dp = a.dplanes + (long)b * a.ld;
  float* dr = a.draw + (long)b * a.ld;
  const float* rho = a.rho_raw + (long)b * a.Gp;
  const float lraw = a.l[b];
  const float el = expf(fminf(fmaxf(lraw, 0.f), a.clip_library));
  float s = 0.f, dlh = 0.f;
  for (int g = threadIdx.x * 4; g < a.Gp; g += 4096) {
    const float4 r4 = *reinterpret_cast<const float4*>(rho + g), d4 = *reinterpret_cast<const float4*>(dp + g), p4 = *reinterpret_cast<const float4*>(pl + g);
    const float r[4] = {r4.x, r4.y, r4.z, r4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (g + e < a.G) {
        const float inside = (r[e] > 1e-7f && r[e] < 1.f - 1e-7f) ? 1.f : 0.f;
        s += dd[e] * el * inside * r[e];
        dlh += dd[e] * pp[e];
      }
  }
  s = block_sum16(s, sh);
  dlh = block_sum16(dlh, sh);
  if (threadIdx.x == 0) a.dl[b] = (lraw > 0.f && lraw < a.clip_library) ? dlh : 0.f;
  for (int g = threadIdx.x * 4; g < a.Gp; g += 4096) {
    const float4 r4 = *reinterpret_cast<const float4*>(rho + g), d4 = *reinterpret_cast<const float4*>(dp + g);
    const float4 d14 = *reinterpret_cast<const float4*>(dp + a.plane_stride + g), p14 = *reinterpret_cast<const float4*>(pl + a.plane_stride + g);
    const float4 d24 = a.k == 3 ? *reinterpret_cast<const float4*>(dp + 2 * a.plane_stride + g) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float r[4] = {r4.x, r4.y, r4.z, r4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w}, d1[4] = {d14.x, d14.y, d14.z, d14.w}, p1[4] = {p14.x, p14.y, p14.z, p14.w},
                d2[4] = {d24.x, d24.y, d24.z, d24.w};
    float o0[4], o1[4], o2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool live = g + e < a.G;
      const float inside = (r[e] > 1e-7f && r[e] < 1.f - 1e-7f) ? 1.f : 0.f;
      o0[e] = live ? r[e] * (dd[e] * el * inside - s) : 0.f;
      o1[e] = live ? d1[e] * p1[e] : 0.f;
      o2[e] = live ? d2[e] : 0.f;
    }
    *reinterpret_cast<float4*>(dr + g) = make_float4(o0[0], o0[1], o0[2], o0[3]);
    *reinterpret_cast<float4*>(dr + a.plane_stride + g) = make_float4(o1[0], o1[1], o1[2], o1[3]);
    if (a.k == 3) *reinterpret_cast<float4*>(dr + 2 * a.plane_stride + g) = make_float4(o2[0], o2[1], o2[2], o2[3]);
  }
}
static bool scvi_head_vec_ok(const ScviHeadArgs& a) {
  return (a.ld % 4) == 0 && (a.plane_stride % 4) == 0 && (a.Gp % 4) == 0 && !tuning_on("no_scvi_head_vec");
}

static bool scvi_head_reg_ok(const ScviHeadArgs& a) {
  return (a.ld % 4) == 0 && (a.plane_stride % 4) == 0 && (a.Gp % 4) == 0 && a.Gp <= 8192;
}

int launch_scvi_head_fwd(hipStream_t st, const ScviHeadArgs& a) {
  if (scvi_head_reg_ok(a)) {
    if (a.Gp <= 2048) hipLaunchKernelGGL(scvi_head_fwd_reg_kernel<2>, dim3(a.B), dim3(256), 0, st, a);
    else if (a.Gp <= 4096) hipLaunchKernelGGL(scvi_head_fwd_reg_kernel<4>, dim3(a.B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(scvi_head_fwd_reg_kernel<8>, dim3(a.B), dim3(256), 0, st, a);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
  }
  if (scvi_head_vec_ok(a)) hipLaunchKernelGGL(scvi_head_fwd_vec_kernel, dim3(a.B), dim3(1024), 0, st, a);
  else hipLaunchKernelGGL(scvi_head_fwd_kernel, dim3(a.B), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

__global__ __launch_bounds__(256) void scvi_head_bwd_kernel(ScviHeadArgs a) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  const float* pl = a.planes + (long)b * a.ld;
  const float* dp = a.dplanes + (long)b * a.ld;
  float* dr = a.draw + (long)b * a.ld;
  const float* rho = a.rho_raw + (long)b * a.Gp;
  const float lraw = a.l[b];
  const float el = expf(fminf(fmaxf(lraw, 0.f), a.clip_library));
  float s = 0.f, dlh = 0.f;
  for (int g = threadIdx.x; g < a.G; g += 256) {
    const float r = rho[g];
    const float drate = dp[g];
    const float inside = (r > 1e-7f && r < 1.f - 1e-7f) ? 1.f : 0.f;
    s += drate * el * inside * r;
    dlh += drate * pl[g];
  }
  s = block_sum(s, sh);
  dlh = block_sum(dlh, sh);
  if (threadIdx.x == 0) a.dl[b] = (lraw > 0.f && lraw < a.clip_library) ? dlh : 0.f;
  for (int g = threadIdx.x; g < a.Gp; g += 256) {
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (g < a.G) {
      const float r = rho[g];
      const float inside = (r > 1e-7f && r < 1.f - 1e-7f) ? 1.f : 0.f;
      d0 = r * (dp[g] * el * inside - s);
      d1 = dp[a.plane_stride + g] * pl[a.plane_stride + g];
      if (a.k == 3) d2 = dp[2 * a.plane_stride + g];
    }
    dr[g] = d0;
    dr[a.plane_stride + g] = d1;
    if (a.k == 3) dr[2 * a.plane_stride + g] = d2;
  }
}
int launch_scvi_head_bwd(hipStream_t st, const ScviHeadArgs& a) {
  if (scvi_head_reg_ok(a)) {
    if (a.Gp <= 2048) hipLaunchKernelGGL(scvi_head_bwd_reg_kernel<2>, dim3(a.B), dim3(256), 0, st, a);
    else if (a.Gp <= 4096) hipLaunchKernelGGL(scvi_head_bwd_reg_kernel<4>, dim3(a.B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(scvi_head_bwd_reg_kernel<8>, dim3(a.B), dim3(256), 0, st, a);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
  }
  if (scvi_head_vec_ok(a)) hipLaunchKernelGGL(scvi_head_bwd_vec_kernel, dim3(a.B), dim3(1024), 0, st, a);
  else hipLaunchKernelGGL(scvi_head_bwd_kernel, dim3(a.B), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// ===========================================================================
// SISUA label heads: one wave per cell, lanes over the label columns (P is tens of columns); unlabelled cells
// (mask == 0, 90 % of them at labels_percent = 0.1) contribute nothing and only zero their gradient rows.
// ===========================================================================
__global__ __launch_bounds__(256) void label_loss_kernel(LabelArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const long src = a.rows ? a.rows[b] : b;
  const float* y = a.Y + src * a.ldy;
  const float* raw = a.raw + (long)b * a.ld;
  const float m = a.observed ? 1.f : a.mask ? (a.mask[src] ? 1.f : 0.f) : 0.f;
  const float gs = a.grad_scale * m;
  float llk = 0.f;
  if (m == 0.f) {   // wave-uniform
    if (a.backward) {
      const int width = ((a.kind == SMX_LABEL_NB || a.kind == SMX_LABEL_NBD) ? 2 : (a.kind == SMX_LABEL_ZINB || a.kind == SMX_LABEL_ZINBD) ? 3 :
                         (a.kind == SMX_LABEL_MIXNB || a.kind == SMX_LABEL_MIXGAUSS) ? 3 * a.C : a.kind == SMX_LABEL_MIXZINB ? 4 * a.C : 1) * a.Pp;
      for (int p = lane; p < width; p += 64) a.draw[(long)b * a.ld + p] = 0.f;
    }
  } else if (a.kind == SMX_LABEL_MIXNB || a.kind == SMX_LABEL_MIXGAUSS || a.kind == SMX_LABEL_MIXZINB) {
    // MISA: log p(y_p) = logsumexp_c(log softmax(mix)_c + log f_c(y_p)); f_c = NB(exp(r_c), l_c) with planes C mixture logits,
    // C log total_counts, C logits -- or, for continuous labels ('mixgaussian', vae.py:86-92), f_c = Normal(loc_c,
    // softplus(s_c + softplus_inverse(1))) with planes C mixture logits, C locations, C raw scales.
    // Gradients: d mix_c = resp_c - pi_c, d (component parameters) = resp_c * d log f_c.
    const int C = a.C;
    // MISA(zero_inflated=True) (vae.py:76-84): f_c = ZINB with a fourth group of C gate-logit planes.
    const bool gauss = a.kind == SMX_LABEL_MIXGAUSS, zi = a.kind == SMX_LABEL_MIXZINB;   // (launch-uniform)
    for (int p = lane; p < a.Pp; p += 64) {
      float e[4], d0[4], d1[4], dg[4], mx[4];
      float am = -3.0e38f, jm = -3.0e38f;
      const bool live = p < a.P;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        e[c] = 0.f; d0[c] = 0.f; d1[c] = 0.f; dg[c] = 0.f; mx[c] = 0.f;
        if (c < C && live) {
          float d2;
          mx[c] = raw[c * a.Pp + p];
          if (gauss) {
            const float mu = raw[(C + c) * a.Pp + p];
            const SpSg s = softplus_sigmoid(raw[(2 * C + c) * a.Pp + p] + SMX_SOFTPLUS_INV_1);   // sp = sigma, sg = d sigma / d raw
            const float inv = frcp(s.sp), zz = (y[p] - mu) * inv;
            e[c] = -0.5f * zz * zz - flog(s.sp) - 0.9189385332046727f;   // 0.5 log(2 pi)
            d0[c] = zz * inv;
            d1[c] = (zz * zz - 1.f) * inv * s.sg;
          } else if (zi)
          count_elem<SMX_LLK_ZINB, 0>(y[p], raw[(C + c) * a.Pp + p], raw[(2 * C + c) * a.Pp + p], raw[(3 * C + c) * a.Pp + p], e[c], d0[c], d1[c], dg[c]);
          else
          count_elem<SMX_LLK_NB, 0>(y[p], raw[(C + c) * a.Pp + p], raw[(2 * C + c) * a.Pp + p], 0.f, e[c], d0[c], d1[c], d2);
          am = fmaxf(am, mx[c]);
          jm = fmaxf(jm, mx[c] + e[c]);
        }
      }
      float sa = 0.f, sj = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < C && live) { sa += expf(mx[c] - am); sj += expf(mx[c] + e[c] - jm); }
      const float lse_a = am + logf(sa), lse_j = jm + logf(sj);
      if (live) llk += lse_j - lse_a - (gauss ? 0.f : lgammaf(y[p] + 1.f));
      if (a.backward) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < C) {
            const float resp = live ? expf(mx[c] + e[c] - lse_j) : 0.f, pi = live ? expf(mx[c] - lse_a) : 0.f;
            a.draw[(long)b * a.ld + c * a.Pp + p] = (resp - pi) * gs;
            a.draw[(long)b * a.ld + (C + c) * a.Pp + p] = resp * d0[c] * gs;
            a.draw[(long)b * a.ld + (2 * C + c) * a.Pp + p] = resp * d1[c] * gs;
            if (zi) a.draw[(long)b * a.ld + (3 * C + c) * a.Pp + p] = resp * dg[c] * gs;
          }
      }
    }
    llk = wave_sum(llk);
  } else if (a.kind == SMX_LABEL_NB || a.kind == SMX_LABEL_NBD || a.kind == SMX_LABEL_ZINB || a.kind == SMX_LABEL_ZINBD) {
    // a count posterior over the head's columns, planes as for the gene output (the elementwise likelihood of smx_loss.h)
    const bool zi = a.kind == SMX_LABEL_ZINB || a.kind == SMX_LABEL_ZINBD;   // (launch-uniform)
    for (int p = lane; p < a.Pp; p += 64) {
      float d0 = 0.f, d1 = 0.f, d2 = 0.f;
      if (p < a.P) {
        float e;
        const float p2 = zi ? raw[2 * a.Pp + p] : 0.f;
        if (a.kind == SMX_LABEL_NB) count_elem<SMX_LLK_NB, 0>(y[p], raw[p], raw[a.Pp + p], 0.f, e, d0, d1, d2);
        else if (a.kind == SMX_LABEL_NBD) count_elem<SMX_LLK_NBD, 0>(y[p], raw[p], raw[a.Pp + p], 0.f, e, d0, d1, d2);
        else if (a.kind == SMX_LABEL_ZINB) count_elem<SMX_LLK_ZINB, 0>(y[p], raw[p], raw[a.Pp + p], p2, e, d0, d1, d2);
        else count_elem<SMX_LLK_ZINBD, 0>(y[p], raw[p], raw[a.Pp + p], p2, e, d0, d1, d2);
        llk += e - lgammaf(y[p] + 1.f);
      }
      if (a.backward) {
        a.draw[(long)b * a.ld + p] = d0 * gs; a.draw[(long)b * a.ld + a.Pp + p] = d1 * gs;
        if (zi) a.draw[(long)b * a.ld + 2 * a.Pp + p] = d2 * gs;
      }
    }
    llk = wave_sum(llk);
  } else {
    float mx = -3.0e38f, ysum = 0.f;
    for (int p = lane; p < a.P; p += 64) { mx = fmaxf(mx, raw[p]); ysum += y[p]; }
    mx = wave_max(mx);
    ysum = wave_sum(ysum);
    float se = 0.f;
    for (int p = lane; p < a.P; p += 64) se += expf(raw[p] - mx);
    se = wave_sum(se);
    const float lse = mx + logf(se);
    for (int p = lane; p < a.Pp; p += 64) {
      float d = 0.f;
      if (p < a.P) {
        const float lp = raw[p] - lse;
        llk += y[p] * lp;
        d = y[p] - expf(lp) * ysum;
      }
      if (a.backward) a.draw[(long)b * a.ld + p] = d * gs;
    }
    llk = wave_sum(llk);
  }
  if (lane == 0) a.llk[b] = (a.add ? a.llk[b] : 0.f) + m * llk;
}
// MISA's 'mixtril' head (sisua/models/vae.py:58, the class's own example): ONE C-component mixture over the whole label vector,
// component c = MultivariateNormalTriL(loc_c, L_c), diag(L) = softplus(raw) + 1e-5 (TFP's FillScaleTriL), strict lower triangle = raw.
// Planes of width Pp (config.label_planes): C logit planes (column 0), C location planes, per component P planes = the columns of L
// (plane j, row p >= j).  One wave per cell, lane p = label dimension p (P <= 64); a component's L sits in LDS as [P][P + 1]:
//   u = L^-1 (y - mu)   forward substitution: lane j publishes u_j, the lanes below subtract L[p][j] u_j
//   w = L^-T u          back substitution through the column view
//   log N = -1/2 |u|^2 - sum log L_pp - P/2 log 2 pi;   d mu = w,   d L[p][j] = w_p u_j - [p == j] / L_pp
// and the mixture over components as in label_loss_kernel: d logit_c = resp_c - pi_c, component gradients times resp_c.
// One workgroup per cell, one WAVE per component (the components' substitution chains side by side instead of one after the other
// in a single wave: 30 -> 16 us per launch at 38 label dimensions and two components); wave c keeps its factor in its own LDS tile,
// the components' log densities meet in LDS, every wave then writes its own component's gradient planes.
__global__ __launch_bounds__(256) void label_tril_kernel(LabelArgs a) {
  extern __shared__ float Lsm[];   // C x [P][P + 1] | e [4]
  const int lane = threadIdx.x & 63, c = threadIdx.x >> 6, b = blockIdx.x;   // (blockDim = 64 C)
  const long src = a.rows ? a.rows[b] : b;
  const float* raw = a.raw + (long)b * a.ld;
  float* draw = a.draw + (long)b * a.ld;
  const float m = a.observed ? 1.f : a.mask ? (a.mask[src] ? 1.f : 0.f) : 0.f;
  const float gs = a.grad_scale * m;
  const int C = a.C, P = a.P, Pp = a.Pp, ldl = P + 1;
  if (m == 0.f) {   // (block-uniform)
    if (a.backward) for (int i = threadIdx.x; i < C * (2 + P) * Pp; i += 64 * C) draw[i] = 0.f;
    if (threadIdx.x == 0) a.llk[b] = a.add ? a.llk[b] : 0.f;
    return;
  }
  float* Ls = Lsm + c * P * ldl;
  float* esh = Lsm + C * P * ldl;
  const bool live = lane < P;
  const float yv = live ? a.Y[src * a.ldy + lane] : 0.f;
  const float mxc = raw[c * Pp];
  const float mu = live ? raw[(C + c) * Pp + lane] : 0.f;
  float lpp = 1.f, sg = 0.f;
  for (int j0 = 0; j0 < P; j0 += 8) {   // plane j = column j of L, coalesced over the rows; eight planes' loads in flight at once
    float v8[8];                        // (one load per plane, each followed by its LDS store, was a chain of P memory round trips)
#pragma unroll
    for (int t = 0; t < 8; ++t) v8[t] = (live && j0 + t < P) ? raw[(2 * C + c * P + j0 + t) * Pp + lane] : 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int j = j0 + t;
      float v = v8[t];
      if (j == lane) { const SpSg sp = softplus_sigmoid(v); v = sp.sp + 1e-5f; lpp = v; sg = sp.sg; }
      if (live && j < P) Ls[lane * ldl + j] = j <= lane ? v : 0.f;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();   // (the tile is this wave's own)
  const float inv = frcp(lpp);
  // (pivots broadcast by v_readlane: with __shfl = ds_bpermute a substitution step took two LDS round trips.  Fully unrolled forms with
  // the factor in registers measured SLOWER: the unrolled code's predicated steps beyond P are thousands of serial instructions)
  float r = yv - mu, u = 0.f;
  for (int j = 0; j < P; ++j) {
    const float lj = live ? Ls[lane * ldl + j] : 0.f;
    const float uj = lane_bcast(r * inv, j);
    if (lane == j) u = uj;
    else if (lane > j) r -= lj * uj;
  }
  float sw = u, w = 0.f;
  for (int i = P - 1; i >= 0; --i) {
    const float li = live ? Ls[i * ldl + lane] : 0.f;   // (zero above the diagonal: lanes beyond i add nothing)
    const float wi = lane_bcast(sw * inv, i);
    if (lane == i) w = wi;
    else if (lane < i) sw -= li * wi;
  }
  const float quad = wave_sum(live ? u * u : 0.f), logdet = wave_sum(live ? flog(lpp) : 0.f);
  const float ec = -0.5f * quad - logdet - 0.9189385332046727f * (float)P;
  if (lane == 0) { esh[c] = ec; esh[4 + c] = mxc; }
  __syncthreads();
  float am = -3.0e38f, jm = -3.0e38f;
  for (int q = 0; q < C; ++q) { am = fmaxf(am, esh[4 + q]); jm = fmaxf(jm, esh[4 + q] + esh[q]); }
  float sa = 0.f, sj = 0.f;
  for (int q = 0; q < C; ++q) { sa += expf(esh[4 + q] - am); sj += expf(esh[4 + q] + esh[q] - jm); }   // (component order: every wave the same bits)
  const float lse_a = am + logf(sa), lse_j = jm + logf(sj);
  if (threadIdx.x == 0) a.llk[b] = (a.add ? a.llk[b] : 0.f) + (lse_j - lse_a);
  if (!a.backward) return;
  const float resp = expf(mxc + ec - lse_j), pi = expf(mxc - lse_a);
  for (int p = lane; p < Pp; p += 64) {
    draw[c * Pp + p] = p == 0 ? (resp - pi) * gs : 0.f;
    draw[(C + c) * Pp + p] = p < P ? resp * w * gs : 0.f;   // (p == lane here: P <= 64)
  }
  for (int j = 0; j < P; ++j) {
    const float uj = lane_bcast(u, j);
    float d = 0.f;
    if (live && j < lane) d = w * uj;
    else if (live && j == lane) d = (w * uj - inv) * sg;
    for (int p = lane; p < Pp; p += 64) draw[(2 * C + c * P + j) * Pp + p] = p < P ? resp * d * gs : 0.f;
  }
}

int launch_label_loss(hipStream_t st, const LabelArgs& a) {
  if (a.kind == SMX_LABEL_MIXTRIL) {
    if (a.P < 1 || a.P > 64 || a.C < 2 || a.C > 4) { set_error("label_loss: 'mixtril' heads take 1..64 label dimensions and 2..4 components"); return SMX_ERR_INVALID; }
    const size_t lds = ((size_t)a.C * a.P * (a.P + 1) + 8) * sizeof(float);   // (66.6 KB at C = 4, P = 64)
    static const bool big_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&label_tril_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) == hipSuccess;
    if (lds > 64 * 1024 && !big_ok) { set_error("label_loss: cannot reserve the LDS of the 'mixtril' head"); return SMX_ERR_HIP; }
    hipLaunchKernelGGL(label_tril_kernel, dim3(a.B), dim3(64 * a.C), lds, st, a);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
  }
  hipLaunchKernelGGL(label_loss_kernel, dim3((a.B + 3) / 4), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// ===========================================================================
// per-step scalars, metrics
// ===========================================================================
__device__ inline float adam_lr_t(float lr, float b1, float b2, uint32_t t /* 1-based */) {
  return lr * sqrtf(1.f - powf(b2, (float)t)) / (1.f - powf(b1, (float)t));
}

__global__ void step_begin_kernel(StepState* master, StepState* dst, const int32_t* order, int32_t* rows, int batch,
                                  int cursor_from_master, uint32_t cursor, float lr, float b1, float b2) {
  const uint32_t cur = cursor_from_master ? master->cursor : cursor;
  if (order)
    for (int i = threadIdx.x; i < batch; i += blockDim.x) rows[i] = order[(long)cur * batch + i];
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t step = master->next;
    dst->step = step;
    dst->cursor = cur;
    dst->lr_t = adam_lr_t(lr, b1, b2, step + 1);
    if (cursor_from_master) master->cursor = cur + 1;
  }
}
int launch_step_begin(hipStream_t st, StepState* master, StepState* dst, const int32_t* order, int32_t* rows,
                      int batch, int cursor_from_master, uint32_t cursor, float lr, float b1, float b2) {
  hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(256), 0, st, master, dst, order, rows, batch, cursor_from_master,
                     cursor, lr, b1, b2);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

__device__ inline void metrics_body(const MetricsArgs& a) {
  // ONE workgroup riding with another launch: its life is that launch's.  Every load is part of a batch (a thread's 32 likelihood partials at
  // a time; the per-cell terms and the history cursor beside them), the seven sums meet in LDS behind ONE pair of barriers -- as a sweep of
  // eight loads per round, a remainder loop of dependent loads and a block_sum per term it was ~15 memory round trips and 14 barriers in a
  // row.  The additions keep their order: the same bits.
  __shared__ float sh[7][4];
  float sx = 0.f, sy = 0.f, sk = 0.f, sl = 0.f, st = 0.f, sd = 0.f, so = 0.f;
  const int tid = (int)threadIdx.x;
  const uint32_t cursor = a.hist ? a.state->cursor : 0u;
  // the per-cell terms of this thread's first cell (minibatches of up to 256 cells: every cell), requested ahead of the sweep
  float c_lg = 0.f, c_y = 0.f, c_o = 0.f, c_k = 0.f, c_l = 0.f, c_t = 0.f, c_d0 = 0.f, c_d1 = 0.f;
  {
    const int b = min(tid, a.B - 1);
    if (a.lgx1) c_lg = a.lgx1[a.rows ? a.rows[b] : b];
    if (a.llk_y) c_y = a.llk_y[b];
    if (a.llk_o) c_o = a.llk_o[b];
    if (a.kl) c_k = a.kl[b];
    if (a.kl_l) c_l = a.kl_l[b];
    if (a.tc) { c_t = a.tc[b]; c_d0 = a.dl[b]; c_d1 = a.dl[a.B + b]; }
  }
  // only the batch total of the count log-likelihood is needed: a flat, coalesced sweep of [B][n_chunks]; part[u] takes the elements
  // tid + 256 u + 2048 k of the full rounds (k ascending), part[0] then the remainder one by one -- as the loop this replaces
  const int total = a.B * a.n_chunks;
  {
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nfull = (total > tid + 7 * 256) ? (total - 1 - tid - 7 * 256) / 2048 + 1 : 0;   // rounds with all eight elements in range
    for (int k0 = 0; k0 < nfull; k0 += 4) {
      float v[4][8];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int u = 0; u < 8; ++u) v[kk][u] = a.llk_part[min(tid + (k0 + kk) * 2048 + u * 256, total - 1)];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        if (k0 + kk < nfull) {
#pragma unroll
          for (int u = 0; u < 8; ++u) part[u] += v[kk][u];
        }
    }
    {
      const int i0 = tid + nfull * 2048;   // fewer than eight elements are left for this thread
      float v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = a.llk_part[min(max(i0 + r * 256, 0), total - 1)];
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (i0 + r * 256 < total) part[0] += v[r];
    }
    sx = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
  }
  if (tid < a.B) {
    if (a.lgx1) sx -= c_lg;
    if (a.llk_y) sy += c_y;
    if (a.llk_o) so += c_o;
    if (a.kl) sk += c_k;
    if (a.kl_l) sl += c_l;
    if (a.tc) { st += c_t; sd += c_d0 + c_d1; }
  }
  for (int b = tid + 256; b < a.B; b += 256) {
    if (a.lgx1) sx -= a.lgx1[a.rows ? a.rows[b] : b];
    if (a.llk_y) sy += a.llk_y[b];
    if (a.llk_o) so += a.llk_o[b];
    if (a.kl) sk += a.kl[b];
    if (a.kl_l) sl += a.kl_l[b];
    if (a.tc) { st += a.tc[b]; sd += a.dl[b] + a.dl[a.B + b]; }
  }
  // seven block sums (wave_sum, then the four waves as (0 + 1) + (2 + 3): block_sum's order) behind one pair of barriers
  sx = wave_sum(sx); sy = wave_sum(sy); sk = wave_sum(sk); sl = wave_sum(sl); st = wave_sum(st); sd = wave_sum(sd); so = wave_sum(so);
  __syncthreads();
  if ((tid & 63) == 0) {
    const int w = tid >> 6;
    sh[0][w] = sx; sh[1][w] = sy; sh[2][w] = sk; sh[3][w] = sl; sh[4][w] = st; sh[5][w] = sd; sh[6][w] = so;
  }
  __syncthreads();
  if (tid == 0) {
    auto tot = [&](int k) { return (sh[k][0] + sh[k][1]) + (sh[k][2] + sh[k][3]); };
    sx = tot(0); sy = tot(1); sk = tot(2); sl = tot(3);
    st = a.tc ? tot(4) : st; sd = a.tc ? tot(5) : sd; so = a.llk_o ? tot(6) : so;
    const float s = a.inv_global_batch;
    float o[8];
    o[0] = (a.gamma * st - (sx + so + a.alpha * sy - a.beta * (sk + sl))) * s;
    o[1] = -sx * s;
    o[2] = -sy * s;
    o[3] = sk * s;
    o[4] = sl * s;
    o[5] = st * s; o[6] = a.tc ? (sd - a.alpha * sy) * s : 0.f; o[7] = -so * s;
#pragma unroll
    for (int i = 0; i < 8; ++i) a.out[i] = o[i];
    if (a.hist) {
      float* h = a.hist + (long)cursor * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) h[i] = o[i];
    }
  }
}

__global__ __launch_bounds__(256) void metrics_kernel(MetricsArgs a) { metrics_body(a); }
int launch_metrics(hipStream_t st, const MetricsArgs& a) {
  hipLaunchKernelGGL(metrics_kernel, dim3(1), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// ===========================================================================
// optimiser: per-tensor clipnorm + Adam over the flat buffer
// ===========================================================================
__global__ __launch_bounds__(256) void grad_sqsum_kernel(AdamArgs a) {
  const int extra = (int)blockIdx.x - (a.sq_chunks >= 0 ? a.sq_chunks : a.n_chunks);
  if (extra >= 0) {   // extra workgroups: the ELBO scalars of this step, then the moving BatchNorm statistics (data parallel)
    if (a.with_metrics && extra == 0) { metrics_body(a.metrics); return; }
    const int i = (extra - (a.with_metrics ? 1 : 0)) * 256 + (int)threadIdx.x;
    if (i < a.bn_total) a.bn_moving[i] = a.bn_moving[i] * a.bn_momentum + a.bn_batch[i] * a.bn_inv_world * (1.f - a.bn_momentum);
    return;
  }
  __shared__ float sh[4];
  const OptChunk ch = a.chunks[blockIdx.x];
  float s = 0.f;
  const float4* g4 = reinterpret_cast<const float4*>(a.grads + ch.offset);
  for (int i = threadIdx.x; i < ch.count / 4; i += 256) {
    const float4 g = g4[i];
    s += (g.x * g.x + g.y * g.y) + (g.z * g.z + g.w * g.w);
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) a.partial[blockIdx.x] = s;
}

// sum of a range of sum-of-squares slots (fixed order: deterministic); 256 threads
__device__ inline void sq_reduce_body(const float* sl, int cnt, float* dst) {
  __shared__ float sh[4];
  float p[4] = {0.f, 0.f, 0.f, 0.f};
  int i = threadIdx.x;
  for (; i + 3 * 256 < cnt; i += 4 * 256) {
#pragma unroll
    for (int u = 0; u < 4; ++u) p[u] += sl[i + u * 256];
  }
  for (; i < cnt; i += 256) p[0] += sl[i];
  const float s = block_sum((p[0] + p[1]) + (p[2] + p[3]), sh);
  if (threadIdx.x == 0) *dst = s;
}

__global__ __launch_bounds__(256) void adam_update_kernel(AdamArgs a) {
  if ((int)blockIdx.x == a.n_launch && a.use_sq && a.with_metrics) {  // use_sq form: the ELBO scalars ride along here
    metrics_body(a.metrics);
    return;
  }
  if ((int)blockIdx.x < a.n_launch) {
    adam_chunk_body(a, (int)blockIdx.x >= a.gap_from ? (int)blockIdx.x + a.gap_len : (int)blockIdx.x);
    return;
  }
  // the LAST workgroup closes the step (nobody reads next_state / next_rows during this step).  As a duty of workgroup 0 behind its chunk --
  // state -> row ids -> stores: two more dependent round trips and a powf -- it was the launch's critical path.
  if (a.master) {
    const uint32_t step = a.state->step, cur = a.state->cursor;
    if (a.hist_dp && threadIdx.x < 8) a.hist_dp[(long)cur * 8 + threadIdx.x] = a.tail_metrics[threadIdx.x];
    if (a.prepare_next)
      for (int i = threadIdx.x; i < a.batch; i += 256) a.next_rows[i] = a.order[(long)(cur + 1) * a.batch + i];
    if (threadIdx.x == 0) {
      a.master->next = step + 1;
      if (a.prepare_next) {
        a.next_state->step = step + 1;
        a.next_state->cursor = cur + 1;
        a.next_state->lr_t = adam_lr_t(a.lr, a.b1, a.b2, step + 2);
      }
    }
  }
}

// the heads' update as a background sweep beside the launches that follow the output head (smx_step.hip: head_sweep_*): a FIXED
// number of workgroups walk the chunks, so the sweep never holds more than a few wave slots per CU and the small dependent
// launches of the main stream are placed at once
template <int NT>
__global__ __launch_bounds__(NT) void adam_sweep_kernel(AdamArgs a, int first, int count) {
  adam_sweep_body<NT>(a, first, count);
}
// the chunks' sums of squares for a RANGE of chunks (data parallel, chained form: the heads' chunks on the communication stream behind
// their bucket's all-reduce; the optimiser launch's own pass then covers the front chunks only)
__global__ __launch_bounds__(256) void grad_sqsum_range_kernel(AdamArgs a, int first) {
  __shared__ float sh[4];
  const int chunk = first + (int)blockIdx.x;
  const OptChunk ch = a.chunks[chunk];
  float s = 0.f;
  const float4* g4 = reinterpret_cast<const float4*>(a.grads + ch.offset);
  for (int i = threadIdx.x; i < ch.count / 4; i += 256) {
    const float4 g = g4[i];
    s += (g.x * g.x + g.y * g.y) + (g.z * g.z + g.w * g.w);
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) a.partial[chunk] = s;
}
int launch_grad_sqsum_range(hipStream_t st, const AdamArgs& a, int first, int count) {
  if (count <= 0) return SMX_OK;
  hipLaunchKernelGGL(grad_sqsum_range_kernel, dim3((unsigned)count), dim3(256), 0, st, a, first);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}
int launch_adam_sweep(hipStream_t st, const AdamArgs& a, int first, int count, int wgs) {
  if (count <= 0) return SMX_OK;
  // (norms from the products' sum-of-squares partials, or -- use_sq = 0 -- from a.partial, filled by launch_grad_sqsum_range before)
  if (wgs <= 0) { set_error("adam sweep: no workgroups"); return SMX_ERR_INVALID; }
  hipLaunchKernelGGL(adam_sweep_kernel<256>, dim3((unsigned)std::min(wgs, count)), dim3(256), 0, st, a, first, count);   // (512-thread workgroups: 181-184 us per c5-shard step against 175-176)
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// ---- flag opt_shard (data parallel, chained form): the heads' optimiser state sharded over the ranks --------------------------------------
// This rank owns the floats [shard_lo, shard_hi) of the head bucket -- a slice cut at a 64-float boundary, through chunks where it falls.
// (1) per chunk, the sum of squares of the chunk's part inside the slice (0 for a chunk outside it): summed over the ranks these are the
//     chunks' sums of squares of the whole reduced gradient;  (2) the tensors' norms from them, on every rank, for the metrics;
// (3) clip + Adam of the elements inside the slice, the tensor's norm from the summed partials (adam_tensor_clip's use_sq = 0 path).
__device__ inline void shard_range(const AdamArgs& a, const OptChunk& ch, int& i_lo, int& i_hi) {
  const long n4 = ch.count / 4;
  const long lo = (a.shard_lo - (long)ch.offset) / 4, hi = (a.shard_hi - (long)ch.offset + 3) / 4;   // (offsets and bounds are multiples of 4)
  i_lo = (int)(lo < 0 ? 0 : (lo > n4 ? n4 : lo));
  i_hi = (int)(hi < 0 ? 0 : (hi > n4 ? n4 : hi));
}
__global__ __launch_bounds__(256) void grad_sqsum_shard_kernel(AdamArgs a, int first) {
  __shared__ float sh[4];
  const int chunk = first + (int)blockIdx.x;
  const OptChunk ch = a.chunks[chunk];
  int i_lo, i_hi;
  shard_range(a, ch, i_lo, i_hi);
  float s = 0.f;
  const float4* g4 = reinterpret_cast<const float4*>(a.grads + ch.offset);
  for (int i = i_lo + (int)threadIdx.x; i < i_hi; i += 256) {
    const float4 g = g4[i];
    s += (g.x * g.x + g.y * g.y) + (g.z * g.z + g.w * g.w);
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) a.partial[chunk] = s;
}
// one workgroup per tensor of the chunk range [first, first + count): the b-th tensor is found by walking the chunk table
__global__ __launch_bounds__(256) void head_norms_kernel(AdamArgs a, int first, int count) {
  __shared__ float sh[4];
  int c = first;
  for (int b = 0; b < (int)blockIdx.x && c < first + count; ++b) c += a.chunks[c].n_chunks;
  if (c >= first + count) return;
  const OptChunk ch = a.chunks[c];
  float s = 0.f;
  for (int k = threadIdx.x; k < ch.n_chunks; k += 256) s += a.partial[ch.first_chunk + k];
  s = block_sum(s, sh);
  if (ch.tensor == a.tied_t0 || ch.tensor == a.tied_t1) s *= a.tied_inv;
  if (threadIdx.x == 0) a.tensor_norm[ch.tensor] = sqrtf(s) * a.grad_scale;
}
__global__ __launch_bounds__(256) void adam_shard_kernel(AdamArgs a, int first, int count) {
  int cur_t = -1;
  float clip = 0.f;
  const float lr_t = a.state->lr_t;
  for (int c = (int)blockIdx.x; c < count; c += (int)gridDim.x) {
    const int chunk = first + c;
    const OptChunk ch = a.chunks[chunk];
    int i_lo, i_hi;
    shard_range(a, ch, i_lo, i_hi);
    if (i_lo >= i_hi) continue;   // (block-uniform) a chunk of another rank's slice
    if (ch.tensor != cur_t) {     // the factor of this chunk's tensor, from the summed partials (the launcher insists on use_sq = 0; every thread takes part: two barriers)
      clip = adam_tensor_clip<256>(a, ch, -1);   // (chunk -1: tensor_norm is head_norms_kernel's to write)
      cur_t = ch.tensor;
    }
    const smx_f32x4* g4 = reinterpret_cast<const smx_f32x4*>(a.grads + ch.offset);
    smx_f32x4* m4 = reinterpret_cast<smx_f32x4*>(a.m + ch.offset);
    smx_f32x4* v4 = reinterpret_cast<smx_f32x4*>(a.v + ch.offset);
    smx_f32x4* p4 = reinterpret_cast<smx_f32x4*>(a.params + ch.offset);
    for (int i = i_lo + (int)threadIdx.x; i < i_hi; i += 256) {
      smx_f32x4 m = m4[i], v = v4[i], p = p4[i];
      adam_apply4(a, clip, lr_t, g4[i], m, v, p);
      m4[i] = m; v4[i] = v; p4[i] = p;
    }
  }
}
int launch_grad_sqsum_shard(hipStream_t st, const AdamArgs& a, int first, int count) {
  if (count <= 0) return SMX_OK;
  hipLaunchKernelGGL(grad_sqsum_shard_kernel, dim3((unsigned)count), dim3(256), 0, st, a, first);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}
int launch_head_norms(hipStream_t st, const AdamArgs& a, int first, int count) {
  if (count <= 0) return SMX_OK;
  hipLaunchKernelGGL(head_norms_kernel, dim3((unsigned)std::min(count, SMX_MAX_TENSORS)), dim3(256), 0, st, a, first, count);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}
int launch_adam_shard(hipStream_t st, const AdamArgs& a, int first, int count, int wgs) {
  if (count <= 0) return SMX_OK;
  if (wgs <= 0 || a.use_sq || a.shard_hi <= a.shard_lo || (a.shard_lo % 4) || (a.shard_hi % 4)) { set_error("adam shard: bad arguments"); return SMX_ERR_INVALID; }
  hipLaunchKernelGGL(adam_shard_kernel, dim3((unsigned)std::min(wgs, count)), dim3(256), 0, st, a, first, count);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// Two launches.  A single-launch form (per-tensor arrival counters, the chunk's gradient kept in registers
// while the workgroup waits for its tensor's other chunks) was measured at 39 us against 16 us for this pair:
// an agent-scope acquire/release round across the 8 XCDs costs far more than a kernel boundary (1.5 us).
int launch_adam(hipStream_t st, const AdamArgs& a) {
  if (a.use_sq) {   // norms come from the weight-gradient products: no pass over the gradient buffer
    hipLaunchKernelGGL(adam_update_kernel, dim3(a.n_launch + (a.with_metrics ? 1 : 0) + (a.master ? 1 : 0)), dim3(256), 0, st, a);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
  }
  hipLaunchKernelGGL(grad_sqsum_kernel, dim3((a.sq_chunks >= 0 ? a.sq_chunks : a.n_chunks) + (a.with_metrics ? 1 : 0) + (a.bn_total + 255) / 256), dim3(256), 0, st, a);
  hipLaunchKernelGGL(adam_update_kernel, dim3(a.n_launch + (a.master ? 1 : 0)), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// ===========================================================================
// noise probe (tests): what the kernels draw for (cell, column)
// ===========================================================================
__global__ void noise_probe_kernel(NoiseKey nk, const int64_t* cell_ids, int B, int width, float p, float* mult,
                                   float* normal) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * width) return;
  const int b = idx / width, c = idx % width;
  const U4 w = philox_block(nk, (uint32_t)cell_ids[b], (uint32_t)(c >> 2));
  if (mult) mult[idx] = (p > 0.f) ? dropout_mult1(w, c & 3, p, 1.f / (1.f - p)) : 1.f;
  if (normal) {
    const float4 n = normal4(w);
    normal[idx] = (c & 3) == 0 ? n.x : (c & 3) == 1 ? n.y : (c & 3) == 2 ? n.z : n.w;
  }
}
int launch_noise_probe(hipStream_t st, NoiseKey nk, const int64_t* cell_ids, int B, int width, float p, float* mult,
                       float* normal) {
  hipLaunchKernelGGL(noise_probe_kernel, dim3((B * width + 255) / 256), dim3(256), 0, st, nk, cell_ids, B, width, p,
                     mult, normal);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

}  // namespace smx

#ifdef SMX_STAMPS
extern "C" int smx_dbg_stamps_kernels(long long* out) {   // development builds only (tools/c2_stamps.sh): this unit's stamp table [16][16]
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(smx::smx_tu_stamps), sizeof(long long) * 256) == hipSuccess ? 0 : -1;
}
#endif
