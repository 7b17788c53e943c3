// smx_dgemm.h -- the workgroup body of smx_dgemm.hip's kernel (see there), shared with the launch that runs a layer's weight
// gradient and its input gradient side by side (smx_headbwd.hip: wgrad_dgemm_kernel).
#pragma once
#include "smx_internal.h"
#include "smx_panel.h"

namespace smx {

#define SMX_DG_LD 36   // floats per row of a wave's operand tile: 32 k + 4 (16-byte aligned, rows 4 banks apart)

#define SMX_DG_SMEM_FLOATS(B_KC) ((B_KC ? 2 : 1) * 8 * 32 * SMX_DG_LD < 8 * 1024 ? 8 * 1024 : (B_KC ? 2 : 1) * 8 * 32 * SMX_DG_LD)

// one 32 x 32 tile of C by a 512-thread workgroup; bid = the workgroup's index among the product's; smem: SMX_DG_SMEM_FLOATS(B_KC) floats
template <int B_KC>
__device__ inline void dgemm_body(const GemmArgs& g, const int bid, float* smem) {
  const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (q in a scalar register)
  const int i = lane & 31, hh = lane >> 5;
  const int n_mt = (g.M + 31) / 32, n_nt = g.N / 32;
  const int xcd = bid & 7, idx = bid >> 3;
  const int mt = idx % n_mt, nt = (idx / n_mt) * 8 + xcd;
  if (nt >= n_nt) return;
  const int m0 = mt * 32, n0 = nt * 32;
  float* ta = smem + q * (32 * SMX_DG_LD);                         // this wave's tile of A
  float* tb = smem + (8 + q) * (32 * SMX_DG_LD);                   // ... and of a k-contiguous Bm
  smx_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // coalesced side: load j of a round covers rows 8 j + lane / 8, the 16 bytes at k = 4 (lane % 8) of the wave's 32
  const int lr = lane >> 3, lc = lane & 7;
  const float* ap[4];
  const float* bq[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ap[j] = g.A + (long)min(m0 + 8 * j + lr, g.M - 1) * g.lda + 4 * lc;   // rows beyond M compute garbage that nobody stores
    bq[j] = g.B + (long)(n0 + 8 * j + lr) * g.ldb + 4 * lc;               // (B_KC; N is a multiple of 32)
  }
  // n-contiguous Bm: a raw buffer over its K rows
  const __amdgpu_buffer_rsrc_t rb = panel_rsrc(g.B, (long)g.K * g.ldb * 4);
  const int vo_b = (8 * hh * g.ldb + n0 + i) * 4;
  // (named quads, not arrays: an array of float4 went through scratch on its way from the loads to the LDS stores)
  float4 a0, a1, a2, a3, b0, b1, b2, b3;
  float bx[2][8], bn[2][8];
  auto load_round = [&](int kb, float (&bo)[2][8]) {
    a0 = *reinterpret_cast<const float4*>(ap[0] + kb); a1 = *reinterpret_cast<const float4*>(ap[1] + kb);
    a2 = *reinterpret_cast<const float4*>(ap[2] + kb); a3 = *reinterpret_cast<const float4*>(ap[3] + kb);
    if (B_KC) {
      b0 = *reinterpret_cast<const float4*>(bq[0] + kb); b1 = *reinterpret_cast<const float4*>(bq[1] + kb);
      b2 = *reinterpret_cast<const float4*>(bq[2] + kb); b3 = *reinterpret_cast<const float4*>(bq[3] + kb);
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 8; ++s)
          bo[t][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, vo_b, (kb + 16 * t + s) * g.ldb * 4, 0));
    }
  };
  float* wa = ta + lr * SMX_DG_LD + 4 * lc;
  float* wb = tb + lr * SMX_DG_LD + 4 * lc;
  if (32 * q < g.K) load_round(32 * q, bx);
  for (int kb = 32 * q; kb < g.K; kb += 256) {   // K is a multiple of 32: a wave's 32 k are all in
    *reinterpret_cast<float4*>(wa) = a0; *reinterpret_cast<float4*>(wa + 8 * SMX_DG_LD) = a1;
    *reinterpret_cast<float4*>(wa + 16 * SMX_DG_LD) = a2; *reinterpret_cast<float4*>(wa + 24 * SMX_DG_LD) = a3;
    if (B_KC) {
      *reinterpret_cast<float4*>(wb) = b0; *reinterpret_cast<float4*>(wb + 8 * SMX_DG_LD) = b1;
      *reinterpret_cast<float4*>(wb + 16 * SMX_DG_LD) = b2; *reinterpret_cast<float4*>(wb + 24 * SMX_DG_LD) = b3;
    }
    // the next round's loads go out as soon as the LDS stores have read their registers
    if (kb + 256 < g.K) load_round(kb + 256, bn);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed (the tile is private to the wave)
    __builtin_amdgcn_wave_barrier();
    // step t: lane (i, hh) supplies k = 16 t + 8 hh + s of the wave's 32, for A's row i and Bm's column i
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float* pa = ta + i * SMX_DG_LD + 16 * t + 8 * hh;
      const float4 x0 = *reinterpret_cast<const float4*>(pa), x1 = *reinterpret_cast<const float4*>(pa + 4);
      const float ax[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      if (B_KC) {
        const float* pb = tb + i * SMX_DG_LD + 16 * t + 8 * hh;
        const float4 y0 = *reinterpret_cast<const float4*>(pb), y1 = *reinterpret_cast<const float4*>(pb + 4);
        const float by[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
        acc = mfma_bf16x3(split3x8(ax), split3x8(by), acc);
      } else {
        acc = mfma_bf16x3(split3x8(ax), split3x8(bx[t]), acc);
      }
    }
    if (!B_KC) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 8; ++s) bx[t][s] = bn[t][s];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // the reads are done before the next round overwrites the tile
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();   // every wave is done with its operand tiles: the partial tiles go over them
  float* red = smem;
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(q * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  const int col = n0 + i;
  const float bias = g.bias ? g.bias[col] : 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = 2 * q + j;   // accumulator register r of a 32 x 32 tile is row (r & 3) + 8 (r >> 2) + 4 hh, column i
    float t = red[(0 * 16 + r) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) t += red[(w * 16 + r) * 64 + lane];
    const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
    if (row < g.M) {
      float v = t + bias;
      if (g.act == 1) v = fmaxf(v, 0.f) + g.leak * fminf(v, 0.f);
      else if (g.act == 2) v = g.act_out[(long)((g.act_wrap > 0 && row >= g.act_wrap) ? row - g.act_wrap : row) * g.act_ld + col] > 0.f ? v : v * g.leak;
      g.C[(long)row * g.ldc + col] = v;
    }
  }
}

}  // namespace smx
