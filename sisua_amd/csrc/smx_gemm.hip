// smx_gemm.hip -- fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32:
// exact fp32 products, k-ordered fmaf accumulation) for every dense product of
// the SISUA step: forward h*W, weight gradient h^T*dy, input gradient dy*W^T.
//
// Shapes are skinny (M or K is the minibatch), so the kernel is organised around
// HBM/L2 traffic and occupancy rather than MFMA peak:
//   * a workgroup is 4 waves, each wave owns one 32x32 accumulator tile; the
//     waves are laid out WM x WN over the output tile and WK over the K tile
//     (WK > 1: in-workgroup split-K, reduced through LDS), plus gridDim.z-way
//     split-K into slabs that the consumer kernel sums in a fixed order
//     (deterministic, no float atomics);
//   * both operands are staged K-major in LDS ([k][m] / [k][n]) so every MFMA
//     operand read is a conflict-free ds_read_b32 row;
//   * the A operand can be produced on the fly from the resident cells x genes
//     matrix: row gather, log1p and input dropout are applied while staging
//     (SingleCellModel.encode, sisua/models/single_cell_model.py:126-134).
#include <stdlib.h>
#include <string.h>

#include "smx_internal.h"
#include "smx_adam.h"
#include "../../include/sisua_hip.h"

namespace smx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int XF>
__device__ inline float4 xform4(float4 v, const AXform& xf, int batch_idx, int src_row, int gene) {
  if (XF) {
    if (xf.log1p) {
      v.x = log1p_count(v.x); v.y = log1p_count(v.y); v.z = log1p_count(v.z); v.w = log1p_count(v.w);
    }
    if (xf.inj_mask) {
      const float4 m = *reinterpret_cast<const float4*>(xf.inj_mask + (long)batch_idx * xf.inj_ld + gene);
      v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
    } else if (xf.drop_p > 0.f) {
      const U4 w = philox_block(xf.nk, xf.cell_base + (uint32_t)src_row, (uint32_t)(gene >> 2));
      const float4 m = dropout_mult4(w, xf.drop_p, xf.drop_scale);
      v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
    }
  }
  return v;
}

// four consecutive elements of the A operand; with the gather transform A may be the compact uint16 store
// XF: 0 plain operand, 1 gather transform on a float32 store, 2 gather transform on the compact uint16 store
template <int XF>
__device__ inline float4 load_a4(const GemmArgs& g, long off) {
  if (XF == 2) {
    const ushort4 h = *reinterpret_cast<const ushort4*>(reinterpret_cast<const uint16_t*>(g.A) + off);
    return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
  }
  return *reinterpret_cast<const float4*>(g.A + off);
}

// A_KM: A stored [K][M] (direct staging); else [M][K] (transposed staging).
// B_NM: B stored [N][K] (transposed staging); else [K][N] (direct staging).
template <int WM, int WN, int WK, int A_KM, int B_NM>
struct GemmSmem {
  static constexpr int BM = 32 * WM, BN = 32 * WN, BK = 32 * WK;
  static constexpr int LDAS = A_KM ? BM + 4 : BM + 1;
  static constexpr int LDBS = B_NM ? BN + 1 : BN + 4;
  static constexpr int TILES = BK * LDAS + BK * LDBS;
  static constexpr int RED = (WK > 1) ? WK * WM * WN * 1024 : 0;
  static constexpr int STORE = (WK == 1) ? 4 * 32 * 36 : 0;   // per-wave transposed output tiles (wide_store)
  static constexpr int FLOATS0 = TILES > RED ? TILES : RED;
  static constexpr int FLOATS = FLOATS0 > STORE ? FLOATS0 : STORE;
};

template <int WM, int WN, int WK, int A_KM, int B_NM, int XF, int EPI = 0>
__device__ inline void gemm_body(const GemmArgs& g, const int bx, const int by, const int bz, float* smem) {
  constexpr int BM = 32 * WM, BN = 32 * WN, BK = 32 * WK;
  constexpr int LDAS = A_KM ? BM + 4 : BM + 1;
  constexpr int LDBS = B_NM ? BN + 1 : BN + 4;
  constexpr int NA = BM * BK / 1024;  // float4 per thread for the A tile
  constexpr int NB = BN * BK / 1024;
  static_assert(WM * WN * WK == 4, "4 waves per workgroup");
  static_assert(NA >= 1 && NB >= 1, "tile too small for 256 threads");
  float* As = smem;
  float* Bs = smem + BK * LDAS;

  preload(g.A, g.lda, g.B, g.ldb, g.C, g.ldc, g.M, g.N, g.K, g.k_chunk, g.slab_stride, g.bias, g.colsum, g.sq_part, g.act, g.c_colmajor);   // (one batch: smx_device.h)
  if (XF) preload(g.xf.rows, g.xf.log1p, g.xf.inj_mask, g.xf.drop_p, g.xf.cell_base, g.xf.drop_scale, g.xf.inj_ld, g.wide_store);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wk = wave / (WM * WN), wm = (wave / WN) % WM, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = bx * BM, n0 = by * BN;
  const int k_begin = bz * g.k_chunk;
  const int k_end = min(g.K, k_begin + g.k_chunk);

  float4 ra[NA], rb[NB];
  uint2 ra_h[NA];   // (XF == 2: the uint16 counts as loaded)
  int a_src[NA];    // gathered source row of the A piece (XF)

  // The A pieces of a tile are requested in two batches -- every row id, then every piece from a clamped address -- and transformed
  // (uint16 -> float, log1p, dropout) only when they go to LDS.  As a lane-predicated block per piece (row id -> piece -> log1p inside
  // `if (m < M && k < k_end)`) the gathered first layer's four pieces were EIGHT dependent memory round trips one after the other
  // (BASELINE configs[1]: the encoder product's whole launch, 6.3 us).
  auto load_tiles = [&](int k0) {
    int a_m[NA], a_k[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int f = tid + 256 * j;
      if (A_KM) {
        const int kr = f / (BM / 4), mq = f % (BM / 4);
        a_k[j] = min(k0 + kr, k_end - 1); a_m[j] = min(m0 + mq * 4, g.M - 4);
      } else {
        const int mr = f / (BK / 4), kq = f % (BK / 4);
        a_m[j] = min(m0 + mr, g.M - 1); a_k[j] = min(k0 + kq * 4, k_end - 4);
      }
      const int id = A_KM ? a_k[j] : a_m[j];
      a_src[j] = id;
      if (XF && g.xf.rows) a_src[j] = g.xf.rows[id];
    }
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const long off = (long)a_src[j] * g.lda + (A_KM ? a_m[j] : a_k[j]);
      if (XF == 2) ra_h[j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(g.A) + off);
      else ra[j] = *reinterpret_cast<const float4*>(g.A + off);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int f = tid + 256 * j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (B_NM) {
        const int nr = f / (BK / 4), kq = f % (BK / 4);
        const int n = n0 + nr, k = k0 + kq * 4;
        if (k < k_end) v = *reinterpret_cast<const float4*>(g.B + (long)n * g.ldb + k);
      } else {
        const int kr = f / (BN / 4), nq = f % (BN / 4);
        const int k = k0 + kr, n = n0 + nq * 4;
        if (k < k_end) v = *reinterpret_cast<const float4*>(g.B + (long)k * g.ldb + n);
      }
      rb[j] = v;
    }
  };

  auto store_tiles = [&](int k0) {   // k0: the tile the registers hold
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int f = tid + 256 * j;
      float4 v = ra[j];
      if (XF == 2) v = make_float4((float)(ra_h[j].x & 0xFFFFu), (float)(ra_h[j].x >> 16), (float)(ra_h[j].y & 0xFFFFu), (float)(ra_h[j].y >> 16));
      if (A_KM) {
        const int kr = f / (BM / 4), mq = f % (BM / 4);
        const int k = k0 + kr, m = m0 + mq * 4;
        if (k < k_end && m < g.M) v = xform4<XF>(v, g.xf, k, a_src[j], m);
        else v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&As[kr * LDAS + mq * 4]) = v;
      } else {
        const int mr = f / (BK / 4), kq = f % (BK / 4);
        const int m = m0 + mr, k = k0 + kq * 4;
        if (m < g.M && k < k_end) v = xform4<XF>(v, g.xf, m, a_src[j], k);
        else v = make_float4(0.f, 0.f, 0.f, 0.f);
        As[(kq * 4 + 0) * LDAS + mr] = v.x;
        As[(kq * 4 + 1) * LDAS + mr] = v.y;
        As[(kq * 4 + 2) * LDAS + mr] = v.z;
        As[(kq * 4 + 3) * LDAS + mr] = v.w;
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int f = tid + 256 * j;
      if (B_NM) {
        const int nr = f / (BK / 4), kq = f % (BK / 4);
        Bs[(kq * 4 + 0) * LDBS + nr] = rb[j].x;
        Bs[(kq * 4 + 1) * LDBS + nr] = rb[j].y;
        Bs[(kq * 4 + 2) * LDBS + nr] = rb[j].z;
        Bs[(kq * 4 + 3) * LDBS + nr] = rb[j].w;
      } else {
        const int kr = f / (BN / 4), nq = f % (BN / 4);
        *reinterpret_cast<float4*>(&Bs[kr * LDBS + nq * 4]) = rb[j];
      }
    }
  };

  // EPI = 2: what the latent-backward epilogue reads beside the product -- requested NOW, ahead of the tile loads (behind the product and the
  // partial tiles' exchange they were one more memory round trip at the end of a four-workgroup launch)
  constexpr int RPW_E = 16 / WK;
  float e_mu[RPW_E], e_sr[RPW_E], e_sg[RPW_E], e_ep[RPW_E], e_dk[RPW_E], e_za[RPW_E];
  if (EPI == 2) {
    const EpiLatentBwd& e = g.lb;
    // (unconditional loads from clamped indices: a lane-predicated load per register was a block with a wait of its own -- four round trips in a row; what a
    // lane beyond the tile reads is never used.  And NO branch on the launch-uniform switches either: an absent operand is read from `lat` instead (valid
    // at every index used here) and replaced by its constant afterwards -- inside `if (e.stochastic) { loads }` the compiler folded the first USE of the
    // loaded value (s_raw + log(e - 1)) into the branch, against a constant in the other arm, and with it a wait per register: four round trips in a row again)
    const int d = min(n0 + wn * 32 + li, e.Dp - 1);
    const bool has_dk = e.dklz != nullptr, has_za = e.dz_add != nullptr, sto = e.stochastic != 0;
    const float* p_dk = has_dk ? e.dklz : e.lat;
    const float* p_za = has_za ? e.dz_add : e.lat;
    const float* p_sr = sto ? e.lat + e.Dp : e.lat;
    const long ld_sr = sto ? (long)e.ld : (long)e.Dp;   // (a deterministic latent's rows may be Dp wide)
    const float* p_sg = sto ? e.sig : e.lat;
    const float* p_ep = sto ? e.eps : e.lat;
    long bo[RPW_E];
#pragma unroll
    for (int j = 0; j < RPW_E; ++j) {
      const int r = wk * RPW_E + j;
      bo[j] = (long)min(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, g.M - 1);
    }
#pragma unroll
    for (int j = 0; j < RPW_E; ++j) {
      e_dk[j] = p_dk[bo[j] * e.Dp + d];
      e_za[j] = p_za[bo[j] * e.Dp + d];
      e_mu[j] = e.lat[bo[j] * e.ld + d];
      e_sr[j] = p_sr[bo[j] * ld_sr + d];
      e_sg[j] = p_sg[bo[j] * e.Dp + d];
      e_ep[j] = p_ep[bo[j] * e.Dp + d];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < RPW_E; ++j) {
      e_dk[j] = has_dk ? e_dk[j] : 0.f;
      e_za[j] = has_za ? e_za[j] : 0.f;
      e_sr[j] = sto ? e_sr[j] : 0.f;
      e_sg[j] = sto ? e_sg[j] : 1.f;
      e_ep[j] = sto ? e_ep[j] : 0.f;
    }
  }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float csum = 0.f;  // column sum of op(B) (bias gradient), threads < BN of M-tile 0
  const bool do_colsum = (g.colsum != nullptr) && (bx == 0) && (tid < BN);

  if (k_begin < k_end) load_tiles(k_begin);
  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
    store_tiles(k0);
    __syncthreads();
    if (k0 + BK < k_end) load_tiles(k0 + BK);
    const float* as = As + (wk * 32 + lh) * LDAS + wm * 32 + li;
    const float* bs = Bs + (wk * 32 + lh) * LDBS + wn * 32 + li;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float a = as[2 * s * LDAS];
      const float b = bs[2 * s * LDBS];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (do_colsum) {
#pragma unroll 8
      for (int k = 0; k < BK; ++k) csum += Bs[k * LDBS + tid];
    }
    __syncthreads();
  }

  // In-workgroup split-K: every wave parks its partial tile in LDS, then wave wk finishes registers
  // [wk * RPW, (wk + 1) * RPW) of the tile (sum over the WK partials in wave order) -- the reduction, the
  // epilogue arithmetic and the stores are spread over all four waves instead of serialised on one.
  constexpr int RPW = 16 / WK;
  float out[RPW];
  if (WK > 1) {
    float* red = smem;  // tiles are dead after the last barrier of the loop
    float* dst = red + ((wk * WM * WN + wm * WN + wn) * 1024);
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[r * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int r = wk * RPW + j;
      float v = red[((wm * WN + wn) * 1024) + r * 64 + lane];
#pragma unroll
      for (int q = 1; q < WK; ++q) v += red[((q * WM * WN + wm * WN + wn) * 1024) + r * 64 + lane];
      out[j] = v;
    }
  } else {
#pragma unroll
    for (int j = 0; j < RPW; ++j) out[j] = acc[j];
  }
  const int r_base = wk * RPW;   // out[j] is register r_base + j of the wave-level 32x32 tile

  if (g.sq_part) {   // sum of squares of this wave's part of the (weight-gradient) tile
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int r = r_base + j;
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < g.M) sq += out[j] * out[j];
    }
    sq = wave_sum(sq);
    if (lane == 0) g.sq_part[((long)by * g.sq_gx + bx) * 4 + wave] = sq;
  }

  if (EPI == 2) {
    // latent-head backward on the d z tile: rows = cells, cols = latent dims
    const EpiLatentBwd& e = g.lb;
    const int d = n0 + wn * 32 + li;
    const bool live = d < e.D;
    // (the operands were requested at entry)
    const float (&mu)[RPW_E] = e_mu; const float (&sr)[RPW_E] = e_sr; const float (&sg)[RPW_E] = e_sg;
    const float (&ep)[RPW_E] = e_ep; const float (&dk)[RPW_E] = e_dk; const float (&za)[RPW_E] = e_za;
    float o0[RPW], o1[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const float dz = out[j] + za[j];
      if (e.stochastic && e.dklz) {   // SCALE: Monte-Carlo KL, log q depends on (sigma, eps) only
        const float dzt = dz + e.kl_scale * dk[j];
        o0[j] = live ? dzt : 0.f;
        o1[j] = live ? (dzt * ep[j] - e.kl_scale * frcp(sg[j])) * sigmoidf(sr[j] + SMX_SOFTPLUS_INV_1) : 0.f;
      } else if (e.stochastic) {
        o0[j] = live ? dz + e.kl_scale * mu[j] : 0.f;
        o1[j] = live ? (dz * ep[j] + e.kl_scale * (sg[j] - frcp(sg[j]))) * sigmoidf(sr[j] + SMX_SOFTPLUS_INV_1) : 0.f;
      } else {
        o0[j] = (live && !(e.relu && !(mu[j] > 0.f))) ? dz : 0.f;
        o1[j] = 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int r = r_base + j;
      const int b = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (b < g.M) {
        e.dlat[(long)b * e.ld + d] = o0[j];
        if (e.stochastic) e.dlat[(long)b * e.ld + e.Dp + d] = o1[j];
      }
    }
  } else if (WK == 1 && g.wide_store) {
    // Whole 32 x 32 tile per wave: transpose it through LDS so that every lane stores 16 bytes (4 store
    // instructions of 1 KB per wave instead of 16 of 256 B -- wide outputs are store-issue bound otherwise).
    float* C = g.C + (long)bz * g.slab_stride;
    float* tile = smem + wave * (32 * 36);   // operand tiles are dead after the loop's last barrier
    const float bias = g.bias ? g.bias[n0 + wn * 32 + li] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = out[r] + bias;
      if (g.act == 1) v = fmaxf(v, 0.f) + g.leak * fminf(v, 0.f);
      else if (g.act == 2) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int arow = (g.act_wrap > 0 && row >= g.act_wrap) ? row - g.act_wrap : row;
        const float y = row < g.M ? g.act_out[(long)arow * g.act_ld + n0 + wn * 32 + li] : 0.f;
        v = y > 0.f ? v : v * g.leak;
      }
      tile[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = v;
    }
    // same wave wrote and reads: no workgroup barrier needed, only the LDS counter
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    const int rq = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = rq + 8 * i;
      const int row = m0 + wm * 32 + rr;
      const float4 v = *reinterpret_cast<const float4*>(&tile[rr * 36 + c4]);
      if (row < g.M) *reinterpret_cast<float4*>(&C[(long)row * g.ldc + n0 + wn * 32 + c4]) = v;
    }
  } else if (RPW == 4 && g.c_colmajor) {
    // column-major slab [N][128]: wave wk finishes rows 8 wk + 4 lh .. + 3 of column li -- one 16-byte store
    float* C = g.C + (long)bz * g.slab_stride;
    const int col = n0 + wn * 32 + li, row = m0 + wm * 32 + 8 * wk + 4 * lh;
    float4 v;
    v.x = row + 0 < g.M ? out[0] : 0.f;
    v.y = row + 1 < g.M ? out[1] : 0.f;
    v.z = row + 2 < g.M ? out[2] : 0.f;
    v.w = row + 3 < g.M ? out[3] : 0.f;
    *reinterpret_cast<float4*>(&C[(long)col * 128 + row]) = v;
  } else {
    float* C = g.C + (long)bz * g.slab_stride;
    const int col = n0 + wn * 32 + li;
    const float bias = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int r = r_base + j;
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < g.M) {
        float v = out[j] + bias;
        if (g.act == 1) v = fmaxf(v, 0.f) + g.leak * fminf(v, 0.f);
        else if (g.act == 2) v = g.act_out[(long)((g.act_wrap > 0 && row >= g.act_wrap) ? row - g.act_wrap : row) * g.act_ld + col] > 0.f ? v : v * g.leak;
        C[(long)row * g.ldc + col] = v;
      }
    }
  }
  if (do_colsum) g.colsum[n0 + tid] = csum;
}

template <int WM, int WN, int WK, int A_KM, int B_NM, int XF>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[GemmSmem<WM, WN, WK, A_KM, B_NM>::FLOATS];
  gemm_body<WM, WN, WK, A_KM, B_NM, XF>(g, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// two products side by side along N in one grid: N tiles [0, ny1) belong to g, the rest to g2
template <int XF>
__global__ __launch_bounds__(256) void gemm_dual_kernel(GemmArgs g, GemmArgs g2, int ny1) {
  __shared__ __attribute__((aligned(16))) float smem[GemmSmem<1, 1, 4, 0, 0>::FLOATS];
  if ((int)blockIdx.y < ny1) gemm_body<1, 1, 4, 0, 0, XF>(g, blockIdx.x, blockIdx.y, blockIdx.z, smem);
  else gemm_body<1, 1, 4, 0, 0, XF>(g2, blockIdx.x, (int)blockIdx.y - ny1, blockIdx.z, smem);
}

// the K4 W^T product with the latent-head backward epilogue as its own kernel
__global__ __launch_bounds__(256) void gemm_latent_bwd_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[GemmSmem<1, 1, 4, 0, 1>::FLOATS];
  gemm_body<1, 1, 4, 0, 1, 0, 2>(g, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}
// ... and with optimiser chunks riding along (the heads' update at a wide gene panel, smx_step.hip: attach_early_adam): workgroups
// [gx, gx + ride_count) of row 0 each apply one chunk; the product's own 4 workgroups leave the chip idle for ~6 us otherwise
__global__ __launch_bounds__(256) void gemm_latent_bwd_ride_kernel(GemmArgs g, AdamArgs a, int gx, int ride_first) {
  __shared__ __attribute__((aligned(16))) float smem[GemmSmem<1, 1, 4, 0, 1>::FLOATS];
  if ((int)blockIdx.x >= gx) {   // (block-uniform)
    if (blockIdx.y == 0) adam_chunk_body<256>(a, ride_first + (int)blockIdx.x - gx);
    return;
  }
  gemm_body<1, 1, 4, 0, 1, 0, 2>(g, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// ---------------------------------------------------------------------------
// Grouped launch: several independent products share one grid, so the narrow ones fill the CUs
// the wide one leaves idle and a whole kernel boundary disappears.  Tiles 128x32 and 32x32(K4).
// ---------------------------------------------------------------------------
struct GemmGroup {
  int n;
  int start[SMX_GROUP_MAX + 1];   // first flat block of each problem
  int gx[SMX_GROUP_MAX], gy[SMX_GROUP_MAX];
  int variant[SMX_GROUP_MAX];     // tile (0: 128x32, 1: 32x32 K4) * 8 + a_kmajor * 4 + b_nmajor * 2 + xform
  GemmArgs p[SMX_GROUP_MAX];
};

__global__ __launch_bounds__(256) void gemm_group_kernel(GemmGroup G) {
  __shared__ __attribute__((aligned(16))) float smem[GemmSmem<1, 1, 4, 1, 0>::FLOATS];
  static_assert(GemmSmem<1, 1, 4, 1, 0>::FLOATS >= GemmSmem<4, 1, 1, 0, 0>::FLOATS &&
                GemmSmem<1, 1, 4, 1, 0>::FLOATS >= GemmSmem<1, 1, 4, 0, 1>::FLOATS &&
                GemmSmem<1, 1, 4, 1, 0>::FLOATS >= GemmSmem<4, 1, 1, 1, 0>::FLOATS, "LDS of the widest variant");
  // Read the descriptor through the kernarg segment pointer: indexing the by-value struct with a
  // run-time problem id would make the compiler copy all of it to scratch in every thread.
  const GemmGroup& Gr = *(const GemmGroup*)__builtin_amdgcn_kernarg_segment_ptr();
  int pi = 0;
  while (pi + 1 < Gr.n && (int)blockIdx.x >= Gr.start[pi + 1]) ++pi;
  const int local = blockIdx.x - Gr.start[pi];
  const int gx = Gr.gx[pi], gy = Gr.gy[pi];
  const int bx = local % gx, by = (local / gx) % gy, bz = local / (gx * gy);
  const GemmArgs& g = Gr.p[pi];
  switch (Gr.variant[pi]) {
    case 0: gemm_body<4, 1, 1, 0, 0, 0>(g, bx, by, bz, smem); break;
    case 1: gemm_body<4, 1, 1, 0, 0, 1>(g, bx, by, bz, smem); break;
    case 2: gemm_body<4, 1, 1, 0, 1, 0>(g, bx, by, bz, smem); break;
    case 4: gemm_body<4, 1, 1, 1, 0, 0>(g, bx, by, bz, smem); break;
    case 5: gemm_body<4, 1, 1, 1, 0, 1>(g, bx, by, bz, smem); break;
    case 8: gemm_body<1, 1, 4, 0, 0, 0>(g, bx, by, bz, smem); break;
    case 9: gemm_body<1, 1, 4, 0, 0, 1>(g, bx, by, bz, smem); break;
    case 10: gemm_body<1, 1, 4, 0, 1, 0>(g, bx, by, bz, smem); break;
    case 11: gemm_body<1, 1, 4, 0, 1, 0, 2>(g, bx, by, bz, smem); break;   // + latent-head backward epilogue
    case 12: gemm_body<1, 1, 4, 1, 0, 0>(g, bx, by, bz, smem); break;
    case 13: gemm_body<1, 1, 4, 1, 0, 1>(g, bx, by, bz, smem); break;
    default: break;
  }
}

template <int WM, int WN, int WK>
static int launch_cfg(hipStream_t st, GemmArgs g, int* eff_split) {
  constexpr int BM = 32 * WM, BN = 32 * WN, BK = 32 * WK;
  g.k_chunk = round_up((g.K + g.split_k - 1) / g.split_k, BK);
  g.split_k = (g.K + g.k_chunk - 1) / g.k_chunk;  // drop empty slices
  if (eff_split) *eff_split = g.split_k;
  if (g.N % BN != 0) { set_error("gemm: N not a multiple of the tile width"); return SMX_ERR_INVALID; }
  dim3 grid((g.M + BM - 1) / BM, g.N / BN, g.split_k), block(256);
  {
    static const int ws = (int)tuning("wide_store", -1);
    g.wide_store = (WK == 1) && (ws >= 0 ? ws != 0 : (long)g.M * g.N >= 65536);
  }
  if (g.sq_part) {
    if (g.split_k != 1) { set_error("gemm: sum-of-squares partials need split_k == 1"); return SMX_ERR_INVALID; }
    g.sq_gx = (int)grid.x;
    if (g.sq_count) *g.sq_count = (int)(grid.x * grid.y) * 4;
  }
  const int mode = (g.a_kmajor ? 2 : 0) | (g.b_nmajor ? 1 : 0);
  if (g.use_xform && g.xf.u16) {
    if (mode == 0) hipLaunchKernelGGL((gemm_kernel<WM, WN, WK, 0, 0, 2>), grid, block, 0, st, g);
    else if (mode == 2) hipLaunchKernelGGL((gemm_kernel<WM, WN, WK, 1, 0, 2>), grid, block, 0, st, g);
    else { set_error("gemm: gather transform only with k-major B"); return SMX_ERR_INVALID; }
  } else if (g.use_xform) {
    if (mode == 0) hipLaunchKernelGGL((gemm_kernel<WM, WN, WK, 0, 0, 1>), grid, block, 0, st, g);
    else if (mode == 2) hipLaunchKernelGGL((gemm_kernel<WM, WN, WK, 1, 0, 1>), grid, block, 0, st, g);
    else { set_error("gemm: gather transform only with k-major B"); return SMX_ERR_INVALID; }
  } else {
    if (mode == 0) hipLaunchKernelGGL((gemm_kernel<WM, WN, WK, 0, 0, 0>), grid, block, 0, st, g);
    else if (mode == 1) hipLaunchKernelGGL((gemm_kernel<WM, WN, WK, 0, 1, 0>), grid, block, 0, st, g);
    else if (mode == 2) hipLaunchKernelGGL((gemm_kernel<WM, WN, WK, 1, 0, 0>), grid, block, 0, st, g);
    else { set_error("gemm: A k-major with B n-major is not used by the model"); return SMX_ERR_INVALID; }
  }
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

int suggest_split_k(int M, int N, int K) {
  if (K < 512) return 1;
  const int tiles = ((M + 31) / 32) * (N / 32);
  static const int cap = (int)tuning("split_cap", 16);
  static const int target = (int)tuning("split_target", 256);
  int s = target / (tiles > 0 ? tiles : 1);
  if (s > cap) s = cap;       // the consumer re-reads every slab
  if (s < 1) s = 1;
  // very deep K (wide gene panels): a slice longer than 8 tiles is a latency-bound loop on a
  // half-empty chip; trade slab traffic for occupancy, up to 64 slices
  static const int deep = (int)tuning("split_deep", 1024);
  while (s < 64 && K / s > deep) s *= 2;
  if (s > 64) s = 64;
  const int chunk = round_up((K + s - 1) / s, 128);  // whole 128-deep tiles per slice
  return (K + chunk - 1) / chunk;
}

static int validate_gemm(GemmArgs& g);

// Products whose 128x32 tiling leaves CUs idle (fewer tiles than this) take the 32x32 K4 tile instead:
// four times the workgroups, each with a quarter of the dependent K loop.  In isolation K4 wins up to ~256 tiles
// (tools/gemm_sweep.py); inside the step, with cold operands, 128 and 256 measure the same.
static int k4_tiles() {
  static const int v = (int)tuning("k4_tiles", 128);
  return v;
}

int launch_gemm_group(hipStream_t st, const GemmArgs* list, int n, int* eff_splits) {
  if (n <= 0 || n > SMX_GROUP_MAX) { set_error("gemm group: 1..SMX_GROUP_MAX problems"); return SMX_ERR_INVALID; }
  GemmGroup G;
  memset(&G, 0, sizeof(G));
  G.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    GemmArgs g = list[i];
    int rc = validate_gemm(g);
    if (rc != SMX_OK) return rc;
    if (g.a_kmajor && g.b_nmajor) { set_error("gemm group: unsupported layout"); return SMX_ERR_INVALID; }
    if (g.use_xform && g.b_nmajor) { set_error("gemm group: gather transform only with k-major B"); return SMX_ERR_INVALID; }
    // tile: K4 (32x32, in-workgroup split-K) for K-deep or narrow problems, 128x32 otherwise
    const int kper = g.K / g.split_k;
    int tile = g.tile;
    if (tile != TILE_128x32 && tile != TILE_32x32_K4)
      tile = (kper >= 128 && (long)((g.M + 127) / 128) * (g.N / 32) < k4_tiles()) ? TILE_32x32_K4 : TILE_128x32;
    const int BM = tile == TILE_128x32 ? 128 : 32, BK = tile == TILE_128x32 ? 32 : 128;
    g.k_chunk = round_up((g.K + g.split_k - 1) / g.split_k, BK);
    g.split_k = (g.K + g.k_chunk - 1) / g.k_chunk;
    if (eff_splits) eff_splits[i] = g.split_k;
    G.gx[i] = (g.M + BM - 1) / BM; G.gy[i] = g.N / 32;
    G.variant[i] = (tile == TILE_128x32 ? 0 : 8) + (g.a_kmajor ? 4 : 0) + (g.b_nmajor ? 2 : 0) + (g.use_xform ? 1 : 0);
    if (g.epi == 2) {
      if (G.variant[i] != 10) { set_error("gemm group: latent epilogue needs the K4 W^T variant"); return SMX_ERR_INVALID; }
      G.variant[i] = 11;
    }
    {
      static const int ws = (int)tuning("wide_store", -1);
      // measured: inside the grouped kernel the transpose costs more than the wider stores save (+0.9 us per step)
      static const bool grp_on = tuning_on("wide_store_group");
      g.wide_store = grp_on && (tile == TILE_128x32) && (ws >= 0 ? ws != 0 : (long)g.M * g.N >= 65536);
    }
    if (g.sq_part) {
      if (g.split_k != 1) { set_error("gemm group: sum-of-squares partials need split_k == 1"); return SMX_ERR_INVALID; }
      g.sq_gx = G.gx[i];
      if (g.sq_count) *g.sq_count = G.gx[i] * G.gy[i] * 4;
    }
    G.start[i] = total;
    total += G.gx[i] * G.gy[i] * g.split_k;
    G.p[i] = g;
  }
  G.start[n] = total;
  // The grouped kernel carries the register budget of its widest variant (2 workgroups per CU).  That is free
  // while the whole group fits the chip in about one wave of workgroups; wide problems (gene panels of 20 000)
  // run faster as separate launches with their own occupancy.
  bool has_epi = false, has_u16 = false;
  for (int i = 0; i < n; ++i) { has_epi |= (G.p[i].epi != 0); has_u16 |= (G.p[i].use_xform && G.p[i].xf.u16); }
  static const bool split_all = tuning_on("split_groups");
  // (the grouped kernel carries no uint16-store variant: those products keep their own launches)
  if ((total > 768 && !has_epi) || n == 1 || split_all || has_u16) {   // a lone product also runs leaner as its own kernel
    for (int i = 0; i < n; ++i) {
      if (G.variant[i] == 11) {
        const GemmArgs& g = G.p[i];
        if (g.ride_adam && g.ride_count > 0)
          hipLaunchKernelGGL(gemm_latent_bwd_ride_kernel, dim3(G.gx[i] + g.ride_count, G.gy[i], 1), dim3(256), 0, st, g, *g.ride_adam, G.gx[i], g.ride_first);
        else
          hipLaunchKernelGGL(gemm_latent_bwd_kernel, dim3(G.gx[i], G.gy[i], 1), dim3(256), 0, st, g);
        continue;
      }
      GemmArgs g = list[i];
      g.tile = (G.variant[i] & 8) ? TILE_32x32_K4 : TILE_128x32;
      int rc = launch_gemm(st, g, nullptr);
      if (rc != SMX_OK) return rc;
    }
    SMX_HIP(hipGetLastError());
    return SMX_OK;
  }
  hipLaunchKernelGGL(gemm_group_kernel, dim3(total), dim3(256), 0, st, G);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

static int validate_gemm(GemmArgs& g) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) { set_error("gemm: empty problem"); return SMX_ERR_INVALID; }
  if ((g.N % 32) || (g.lda % 4) || (g.ldb % 4) || (g.ldc % 4)) {
    set_error("gemm: N must be a multiple of 32 and leading dimensions multiples of 4");
    return SMX_ERR_INVALID;
  }
  if (g.a_kmajor && (g.M % 4)) { set_error("gemm: k-major A needs M % 4 == 0"); return SMX_ERR_INVALID; }
  if (!g.a_kmajor && (g.K % 4)) { set_error("gemm: row-major A needs K % 4 == 0"); return SMX_ERR_INVALID; }
  if (g.b_nmajor && (g.K % 4)) { set_error("gemm: n-major B needs K % 4 == 0"); return SMX_ERR_INVALID; }
  if (g.split_k < 1) g.split_k = 1;
  if (g.split_k > 1 && (g.bias || g.colsum || g.act)) { set_error("gemm: bias/colsum/activation need split_k == 1"); return SMX_ERR_INVALID; }
  if (g.act == 2 && (!g.act_out || g.act_ld < g.N)) { set_error("gemm: activation backward needs the forward output"); return SMX_ERR_INVALID; }
  if (g.colsum && g.b_nmajor) { set_error("gemm: colsum needs k-major B"); return SMX_ERR_INVALID; }
  if (g.epi == 2 && (g.split_k != 1 || !g.lb.dlat || g.N != g.lb.Dp)) {
    set_error("gemm: latent-backward epilogue needs split_k == 1 and N == Dp");
    return SMX_ERR_INVALID;
  }
  return SMX_OK;
}

int launch_gemm_dual(hipStream_t st, const GemmArgs& g1_in, const GemmArgs& g2_in, int* eff_split) {
  GemmArgs g1 = g1_in, g2 = g2_in;
  int rc = validate_gemm(g1);
  if (rc == SMX_OK) rc = validate_gemm(g2);
  if (rc != SMX_OK) return rc;
  if (g1.a_kmajor || g1.b_nmajor || g2.a_kmajor || g2.b_nmajor || g1.epi || g2.epi || g1.sq_part || g2.sq_part || g1.colsum || g2.colsum ||
      g1.A != g2.A || g1.lda != g2.lda || g1.M != g2.M || g1.K != g2.K || g1.use_xform != g2.use_xform ||
      (g1.use_xform && (g1.xf.rows != g2.xf.rows || g1.xf.u16 != g2.xf.u16 || g1.xf.log1p != g2.xf.log1p || g1.xf.drop_p != 0.f || g2.xf.drop_p != 0.f ||
                        g1.xf.inj_mask || g2.xf.inj_mask))) {
    set_error("gemm dual: the two products must share the A operand and its transform");
    return SMX_ERR_INVALID;
  }
  g2.split_k = g1.split_k;
  if (g1.split_k > 1 && (g2.bias || g1.bias)) { set_error("gemm dual: bias needs split_k == 1"); return SMX_ERR_INVALID; }
  g1.k_chunk = round_up((g1.K + g1.split_k - 1) / g1.split_k, 128);
  g1.split_k = (g1.K + g1.k_chunk - 1) / g1.k_chunk;
  g2.k_chunk = g1.k_chunk; g2.split_k = g1.split_k;
  if (eff_split) *eff_split = g1.split_k;
  g1.wide_store = g2.wide_store = 0;
  const int ny1 = g1.N / 32;
  dim3 grid((g1.M + 31) / 32, ny1 + g2.N / 32, g1.split_k), block(256);
  if (g1.use_xform && g1.xf.u16) hipLaunchKernelGGL(gemm_dual_kernel<2>, grid, block, 0, st, g1, g2, ny1);
  else if (g1.use_xform) hipLaunchKernelGGL(gemm_dual_kernel<1>, grid, block, 0, st, g1, g2, ny1);
  else hipLaunchKernelGGL(gemm_dual_kernel<0>, grid, block, 0, st, g1, g2, ny1);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

int launch_gemm(hipStream_t st, const GemmArgs& g_in, int* eff_split) {
  GemmArgs g = g_in;
  int rc = validate_gemm(g);
  if (rc != SMX_OK) return rc;
  if (g.epi != 0) { set_error("gemm: epilogues run through launch_gemm_group"); return SMX_ERR_INVALID; }
  int tile = g.tile;
  if (tile == TILE_AUTO) {
    const int kper = g.K / g.split_k;
    static const int xf_tile = (int)tuning("xf_tile", 0);
    if (g.use_xform && !g.a_kmajor && xf_tile) tile = xf_tile;
    else if (g.use_xform && !g.a_kmajor) tile = TILE_32x32_K4;  // measured best for the gathered log1p operand (log is one v_log)
    else if (kper >= 128 && (long)((g.M + 127) / 128) * (g.N / 32) < k4_tiles()) tile = TILE_32x32_K4;
    else if (g.M > 64 || g.N % 64) tile = TILE_128x32;
    else if (g.M > 32) tile = TILE_64x64;
    else tile = (g.N % 128 == 0) ? TILE_32x128 : TILE_64x64;
  }
  if (g.c_colmajor && (tile != TILE_32x32_K4 || g.M > 128 || g.bias || g.act || g.sq_part || g.colsum || g.slab_stride < (long)g.N * 128)) {
    set_error("gemm: column-major slabs take the 32 x 32 K4 tile, at most 128 rows and no epilogue");
    return SMX_ERR_INVALID;
  }
  switch (tile) {
    case TILE_128x32: return launch_cfg<4, 1, 1>(st, g, eff_split);
    case TILE_64x64: return launch_cfg<2, 2, 1>(st, g, eff_split);
    case TILE_32x128: return launch_cfg<1, 4, 1>(st, g, eff_split);
    case TILE_32x32_K4: return launch_cfg<1, 1, 4>(st, g, eff_split);
    case TILE_64x32_K2: return launch_cfg<2, 1, 2>(st, g, eff_split);
    default: set_error("gemm: unknown tile configuration"); return SMX_ERR_INVALID;
  }
}

}  // namespace smx
