// smx_adam.h -- the optimiser's workgroup body (per-tensor clipnorm + Adam over one chunk of the flat buffer),
// shared by the optimiser launch (smx_kernels.hip), by the launches that carry chunks as riders -- the BatchNorm-backward kernels
// (smx_kernels.hip) and the latent head's backward product (smx_gemm.hip) -- and by the heads' background sweep on the second stream
// (adam_sweep_body; smx_step.hip: head_sweep_*).
#pragma once
#include "smx_device.h"
#include "smx_internal.h"

namespace smx {

typedef float smx_f32x4 __attribute__((ext_vector_type(4)));

// the chunk's tensor: its gradient norm (written once per tensor, by the tensor's first chunk) and the factor its gradients are scaled by.
// Every thread of the workgroup calls it (two barriers); the first 256 threads sum in the same order whatever NT is.
template <int NT = 256>
__device__ inline float adam_tensor_clip(const AdamArgs& a, const OptChunk& ch, int chunk) {
  __shared__ float sh[4];
  float s = 0.f;
  if (NT > 256 && threadIdx.x >= 256) {
  } else if (a.use_sq) {
    const int cnt = a.sq_count[ch.tensor];
    if (cnt > 0) {   // partial sums written by the weight-gradient product's workgroups
      // (a thread's slots eight at a time in flight -- unconditional loads from a clamped index, masked at the add, added in the same order
      // as one by one: same bits.  As a rider of a 5-10 us launch a workgroup's LIFE is what counts, and 7 dependent round trips for the
      // 1672 slots of the one-launch output head were most of it)
      const float* sl = a.sq_slots + a.sq_first[ch.tensor];
      for (int k0 = threadIdx.x; k0 < cnt; k0 += 8 * 256) {
        float q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = sl[min(k0 + 256 * u, cnt - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + 256 * u < cnt) s += q[u];
      }
    } else {         // small tensor (bias, BatchNorm scale / shift): sweep its whole gradient
      const float4* t4 = reinterpret_cast<const float4*>(a.grads + a.chunks[ch.first_chunk].offset);
      for (int k = threadIdx.x; k < ch.tensor_count / 4; k += 256) {
        const float4 q = t4[k];
        s += (q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w);
      }
    }
  } else {
    for (int k = threadIdx.x; k < ch.n_chunks; k += 256) s += a.partial[ch.first_chunk + k];
  }
  s = wave_sum(s);   // (the first four waves' sums, in block_sum's order; later waves stand by)
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  s = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  if (ch.tensor == a.tied_t0 || ch.tensor == a.tied_t1) s *= a.tied_inv;   // (C identical rows of one shared variable)
  const float norm = sqrtf(s) * a.grad_scale;
  float clip = a.grad_scale;
  if (a.clipnorm > 0.f && norm > a.clipnorm) clip *= a.clipnorm / norm;
  if (threadIdx.x == 0 && chunk == ch.first_chunk) a.tensor_norm[ch.tensor] = norm;
  return clip;
}

// one quad of a tensor: the update itself.  Every contraction is SPELLED: the launches that carry this body (optimiser, riders, sweep, the sharded
// chain) must give the same bits, and which multiply-adds the compiler fuses depends on the code around them (round 6: inlined into the output
// head's launch -- tools/dev/head_lazy_update.patch -- the unspelled form differed from the sweep's in the last bit of some elements).
__device__ inline void adam_apply4(float b1, float b2, float eps, float clip, float lr_t, const smx_f32x4& g, smx_f32x4& m, smx_f32x4& v, smx_f32x4& p) {
#pragma clang fp contract(off)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float ge = g[e] * clip;
    m[e] = __builtin_fmaf(b1, m[e], (1.f - b1) * ge);
    v[e] = __builtin_fmaf(b2, v[e], ((1.f - b2) * ge) * ge);
    p[e] = __builtin_fmaf(-(lr_t * m[e]), frcp(fsqrt(v[e]) + eps), p[e]);   // v_sqrt + v_rcp (1 ulp each) instead of 22 instructions
  }
}
__device__ inline void adam_apply4(const AdamArgs& a, float clip, float lr_t, const smx_f32x4& g, smx_f32x4& m, smx_f32x4& v, smx_f32x4& p) {
  adam_apply4(a.b1, a.b2, a.eps, clip, lr_t, g, m, v, p);
}

// clip + Adam for one chunk of the flat buffer; NT = 256 threads, or 512 as a rider of a 512-thread launch (the tensor's norm is
// summed by the first 256 threads in the same order either way: both forms give the same bits).  A thread's first operands are
// requested BEFORE the norm is reduced and each later round's before the current round's arithmetic: a workgroup lives for 2-4
// rounds, so the reduction's barrier and the first loads' latency were a third of its life.  (Nontemporal loads / stores of the
// moments, to keep the weights in the last-level cache, measured slower: c5-shard 198.0 -> 200.0 us, C2 80.4 -> 81.7.)
template <int NT = 256>
__device__ inline void adam_chunk_body(const AdamArgs& a, int chunk) {
  const OptChunk ch = a.chunks[chunk];
  const smx_f32x4* g4 = reinterpret_cast<const smx_f32x4*>(a.grads + ch.offset);
  smx_f32x4* m4 = reinterpret_cast<smx_f32x4*>(a.m + ch.offset);
  smx_f32x4* v4 = reinterpret_cast<smx_f32x4*>(a.v + ch.offset);
  smx_f32x4* p4 = reinterpret_cast<smx_f32x4*>(a.params + ch.offset);
  const int n4 = ch.count / 4;
  int i = threadIdx.x;
  smx_f32x4 g = {0.f, 0.f, 0.f, 0.f}, m = g, v = g, p = g;
  auto fetch = [&](int j, smx_f32x4& go, smx_f32x4& mo, smx_f32x4& vo, smx_f32x4& po) {
    go = g4[j]; mo = m4[j]; vo = v4[j]; po = p4[j];
  };
  const float lr_t = a.state->lr_t;   // (ahead of the norm's barriers: behind them it was a round trip of its own)
  if (i < n4) fetch(i, g, m, v, p);
  const float clip = adam_tensor_clip<NT>(a, ch, chunk);
  while (i < n4) {
    const int j = i + NT;
    smx_f32x4 gn = g, mn = m, vn = v, pn = p;
    if (j < n4) fetch(j, gn, mn, vn, pn);
    adam_apply4(a, clip, lr_t, g, m, v, p);
    m4[i] = m; v4[i] = v; p4[i] = p;
    g = gn; m = mn; v = vn; p = pn; i = j;
  }
}

// the background sweep's workgroup (smx_kernels.hip: adam_sweep_kernel): chunks first + blockIdx.x, + gridDim.x, ... -- the tensor's factor is
// worked out when the tensor changes (the output head's matrix is ~1900 chunks of one tensor), and a thread keeps two quads of every operand in
// flight; element by element the same arithmetic as adam_chunk_body: the same bits
template <int NT>
__device__ inline void adam_sweep_body(const AdamArgs& a, int first, int count) {
  int cur_t = -1;
  float clip = 0.f;
  const float lr_t = a.state->lr_t;
  for (int c = (int)blockIdx.x; c < count; c += (int)gridDim.x) {
    const int chunk = first + c;
    const OptChunk ch = a.chunks[chunk];
    if (ch.tensor != cur_t || chunk == ch.first_chunk) { clip = adam_tensor_clip<NT>(a, ch, chunk); cur_t = ch.tensor; }
    const smx_f32x4* g4 = reinterpret_cast<const smx_f32x4*>(a.grads + ch.offset);
    smx_f32x4* m4 = reinterpret_cast<smx_f32x4*>(a.m + ch.offset);
    smx_f32x4* v4 = reinterpret_cast<smx_f32x4*>(a.v + ch.offset);
    smx_f32x4* p4 = reinterpret_cast<smx_f32x4*>(a.params + ch.offset);
    const int n4 = ch.count / 4;
    for (int i = threadIdx.x; i < n4; i += 2 * NT) {
      const int j = i + NT;
      const bool two = j < n4;
      const int jj = two ? j : i;
      const smx_f32x4 g0 = g4[i], m0 = m4[i], v0 = v4[i], p0 = p4[i];
      const smx_f32x4 g1 = g4[jj], m1 = m4[jj], v1 = v4[jj], p1 = p4[jj];
      smx_f32x4 m = m0, v = v0, p = p0;
      adam_apply4(a, clip, lr_t, g0, m, v, p);
      m4[i] = m; v4[i] = v; p4[i] = p;
      if (two) {
        m = m1; v = v1; p = p1;
        adam_apply4(a, clip, lr_t, g1, m, v, p);
        m4[j] = m; v4[j] = v; p4[j] = p;
      }
    }
  }
}

}  // namespace smx
