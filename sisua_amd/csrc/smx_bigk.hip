// smx_bigk.hip -- products whose CONTRACTION axis is the gene axis of a wide panel (K = tens of thousands) and whose output
// is the small [minibatch][hidden] tile:
//
//   encoder front      pre = log1p(X[rows]) W_enc        [B][Hp]   K = Gp      (B operand k-major [K][N])   SURVEY.md 8 row a-6
//   head backward      d d = dP W_out^T                  [B][Hp]   K = k Gp    (B operand n-major [N][K])   row a-16
//
// The 32 x 32-tile kernels cut such a product into (B / 32)(Hp / 32) tiles x <= 32 K slices because the consumer (a
// BatchNorm launch) sums at most a few dozen split-K slabs: 16 tiles per slice means every operand byte crosses L2 -> CU
// four times, and at 128 x 20 000 that traffic, not the MFMAs, is the launch (d d: 245 MB for 61 MB of operands, 49 us;
// encoder: 27 us at 15 % MFMA busy).  Here ONE workgroup owns the whole [128][128] output of its K slice, there are as many
// slices as the chip has CUs, and the slabs are summed by a second, bandwidth-bound launch (bigk_reduce_kernel) in slice
// order -- deterministic, and the consumer reads ONE slab:
//   * both operands go global -> LDS by LDS-DMA (16 B per lane, landed linearly; the lane -> address map is chosen so that
//     the linear image IS the swizzled tile; inline asm: smx_device.h glds16), up to three 32-deep stages in flight across the
//     barriers (SMX_BIGK_STAGES; default ONE: measured fastest, smx_internal.h), one counted `s_waitcnt vmcnt(N)` + raw
//     `s_barrier` per stage;
//   * 8 waves = 4 row tiles x 2 column halves, two 32 x 32 accumulators each; operands read from LDS as the lanes' runs of
//     8 consecutive k, split three ways in registers: bf16 MFMAs, f32 accuracy (smx_device.h);
//   * the gather by row id, the uint16 store and log1p of the encoder front are applied on the way (address of the DMA / at
//     the LDS read).
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "smx_internal.h"
#include "../../include/sisua_hip.h"

namespace smx {

#define BIGK_STAGES 4   // most stages in flight (LDS of the deepest form); a.stages of them are used

// A_U16: A is the compact uint16 store; B_KM: B stored [K][N] (else [N][K]); LOG1P on A
template <int A_U16, int B_KM, int LOG1P>
__global__ __launch_bounds__(512) void bigk_kernel(BigKArgs a) {
  constexpr int A_STAGE = A_U16 ? 128 * 32 * 2 : 128 * 32 * 4;   // bytes: [128 rows][32 k]
  constexpr int B_STAGE = 128 * 32 * 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* As = lds;                               // BIGK_STAGES x A_STAGE
  const int NS = a.stages;                               // stage buffers in use (2..BIGK_STAGES)
  unsigned char* Bs = lds + NS * A_STAGE;                // NS x B_STAGE
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 31, hh = lane >> 5;
  const int rt = w & 3, ch = w >> 2;
  const int z = blockIdx.x, m0 = blockIdx.y * 128, n0 = blockIdx.z * 128;
  const long kbeg = (long)z * a.k_chunk, kend = std::min<long>((long)a.K, kbeg + a.k_chunk);   // multiples of 32
  const int n_st = (int)((kend - kbeg + 31) / 32);

  // ---- DMA of one stage: A rows m0 .. m0 + 127 (clamped to M - 1; gathered), B rows / columns n0 .. n0 + 127 (clamped).  The
  // lane's source pointers are formed once and advance by one stage per call (stages are issued in order) ----
  // A, uint16: a row is 64 B = 4 lanes; one wave-instruction = 16 rows; 8 per stage, one per wave; LDS chunk c' of row m holds
  //            global chunk c' ^ ((m >> 2) & 3).
  // A, float32 (and B stored [N][K]): a row is 128 B = 8 lanes; one wave-instruction = 8 rows; 16 per stage, two per wave; LDS
  //            chunk c' of row m holds global chunk c' ^ ((m >> 1) & 7).
  // B stored [K][N]: [32 k][128 n]: a k row is 512 B = 32 lanes; one wave-instruction = 2 k rows; 16 per stage, two per wave; LDS
  //            chunk c' of row k holds global chunk c' ^ (8 ((k >> 3) & 1)) (the two lane halves of an operand read are 8 rows apart).
  const unsigned char* ag[2];
  const float* bg[2];
  {
    if (A_U16) {
      const int m = 16 * w + (lane >> 2), cq = (lane & 3) ^ ((m >> 2) & 3);
      const int row = std::min(m0 + m, a.M - 1);
      const long src = a.rows ? (long)a.rows[row] : (long)row;
      ag[0] = reinterpret_cast<const unsigned char*>(reinterpret_cast<const uint16_t*>(a.A) + src * a.lda + kbeg + 8 * cq);
      ag[1] = ag[0];
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int m = 8 * (2 * w + u) + (lane >> 3), cq = (lane & 7) ^ ((m >> 1) & 7);
        const int row = std::min(m0 + m, a.M - 1);
        const long src = a.rows ? (long)a.rows[row] : (long)row;
        ag[u] = reinterpret_cast<const unsigned char*>(a.A + src * a.lda + kbeg + 4 * cq);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (B_KM) {
        const int k = 2 * (2 * w + u) + (lane >> 5), cq = (lane & 31) ^ (8 * ((k >> 3) & 1));
        bg[u] = a.Bm + (kbeg + k) * a.ldb + std::min(n0 + 4 * cq, a.N - 4);
      } else {
        const int n = 8 * (2 * w + u) + (lane >> 3), cq = (lane & 7) ^ ((n >> 1) & 7);
        bg[u] = a.Bm + (long)std::min(n0 + n, a.N - 1) * a.ldb + kbeg + 4 * cq;
      }
    }
  }
  const long b_step = B_KM ? 32 * a.ldb : 32;
  const uint32_t as_l = lds_addr(As), bs_l = lds_addr(Bs);
  auto issue = [&](int st) {
    const uint32_t ad = as_l + (st % NS) * A_STAGE, bd = bs_l + (st % NS) * B_STAGE;
    if (A_U16) {
      glds16(ag[0], __builtin_amdgcn_readfirstlane(ad + 1024 * w));
      ag[0] += 64;
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        glds16(ag[u], __builtin_amdgcn_readfirstlane(ad + 1024 * (2 * w + u)));
        ag[u] += 128;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      glds16(bg[u], __builtin_amdgcn_readfirstlane(bd + 1024 * (2 * w + u)));
      bg[u] += b_step;
    }
  };

  smx_f32x16 acc[2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  for (int st = 0; st < NS - 1 && st < n_st; ++st) issue(st);
  for (int st = 0; st < n_st; ++st) {
    // this wave's DMA instructions complete in issue order: stage st has landed once at most the instructions of the
    // stages st + 1, st + 2 are outstanding (IPS per stage; fewer stages follow near the end), then the barrier collects the
    // other waves' pieces.  The buffer of stage st + 3, requested below, was last read in stage st - 1, which every wave has
    // left once it passes this barrier.  (`s_waitcnt vmcnt(0)` here would wait for the stage requested one iteration ago:
    // a prefetch distance of ONE stage -- 24 us for d d at 128 x 60 000 against 9.)
    constexpr int IPS = A_U16 ? 3 : 4;
    const int after = n_st - 1 - st;
    if (NS >= 4 && after >= 2) __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * IPS));
    else if (NS >= 3 && after >= 1) __builtin_amdgcn_s_waitcnt(0x0F70 | IPS);
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    if (st + NS - 1 < n_st) issue(st + NS - 1);
    const unsigned char* ab = As + (st % NS) * A_STAGE;
    const unsigned char* bb = Bs + (st % NS) * B_STAGE;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // ---- A: row m = 32 rt + i, k = 16 t + 8 hh .. + 7 ----
      float av[8];
      const int m = 32 * rt + i;
      if (A_U16) {
        const int c = (2 * t + hh) ^ ((m >> 2) & 3);
        const uint4 raw = *reinterpret_cast<const uint4*>(ab + m * 64 + c * 16);
        const uint32_t wd[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { av[2 * e] = (float)(wd[e] & 0xFFFFu); av[2 * e + 1] = (float)(wd[e] >> 16); }
      } else {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int c = (4 * t + 2 * hh + e) ^ ((m >> 1) & 7);
          const float4 v = *reinterpret_cast<const float4*>(ab + m * 128 + c * 16);
          av[4 * e] = v.x; av[4 * e + 1] = v.y; av[4 * e + 2] = v.z; av[4 * e + 3] = v.w;
        }
      }
      if (LOG1P) {
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = log1p_count(av[e]);
      }
      const Split8 sa = split3x8(av);
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const int n = 64 * ch + 32 * c2 + i;   // column of this lane within the workgroup's 128
        float bv[8];
        if (B_KM) {
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            const int k = 16 * t + 8 * hh + s;
            bv[s] = *reinterpret_cast<const float*>(bb + k * 512 + 4 * (n ^ (32 * ((k >> 3) & 1))));
          }
        } else {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int c = (4 * t + 2 * hh + e) ^ ((n >> 1) & 7);
            const float4 v = *reinterpret_cast<const float4*>(bb + n * 128 + c * 16);
            bv[4 * e] = v.x; bv[4 * e + 1] = v.y; bv[4 * e + 2] = v.z; bv[4 * e + 3] = v.w;
          }
        }
        acc[c2] = mfma_bf16x3(sa, split3x8(bv), acc[c2]);
      }
    }
  }
  // ---- this slice's slab: register r of a tile is row (r & 3) + 8 (r >> 2) + 4 hh, column i ----
  float* out = a.part + (long)z * a.slab_stride;
  if (a.colmajor) {
    // column-major slab [N][128]: a lane's four consecutive rows of a column are one 16-byte store (M <= 128: one row block; rows beyond M zero)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      const int col = n0 + 64 * ch + 32 * c2 + i;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = 32 * rt + 8 * q + 4 * hh;
        float4 v;
        v.x = row + 0 < a.M ? acc[c2][4 * q + 0] : 0.f;
        v.y = row + 1 < a.M ? acc[c2][4 * q + 1] : 0.f;
        v.z = row + 2 < a.M ? acc[c2][4 * q + 2] : 0.f;
        v.w = row + 3 < a.M ? acc[c2][4 * q + 3] : 0.f;
        if (col < a.N) *reinterpret_cast<float4*>(out + (long)col * 128 + row) = v;
      }
    }
    return;
  }
#pragma unroll
  for (int c2 = 0; c2 < 2; ++c2) {
    const int col = n0 + 64 * ch + 32 * c2 + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (row < a.M && col < a.N) out[(long)row * a.ldc + col] = acc[c2][r];
    }
  }
}

// out[e] = sum over the slices of part[z][e], in a FIXED order: 16 threads share an output float4, thread t of them sums the
// slices z = t, t + 16, ... (increasing), the 16 partial sums are added in t order.  A workgroup owns 16 consecutive float4
// (its threads' loads of one slab are 256 contiguous bytes): 256 workgroups for a 128 x 128 output instead of 16, whose
// 4096 threads walked the 200+ slabs one after the other (11 us for 14 MB).
__global__ __launch_bounds__(256) void bigk_reduce_kernel(const float* __restrict__ part, long slab_stride, int n_slices, long n4, float* __restrict__ out) {
  __shared__ float4 sh[16][16];
  const int el = threadIdx.x & 15, t = threadIdx.x >> 4;
  const long e = (long)blockIdx.x * 16 + el;
  const long s4 = slab_stride >> 2;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < n4) {
    const float4* p = reinterpret_cast<const float4*>(part) + e;
    // ALL of this thread's slices of a batch of 256 in flight at once (~13 loads at 209 slices: ONE memory round trip instead of four;
    // unconditional loads from a clamped slice -- a predicated 16-byte load is split into four 4-byte ones -- masked at the add; same
    // order of additions as before: same bits)
    for (int z0 = t; z0 < n_slices; z0 += 256) {
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = p[(long)min(z0 + 16 * u, n_slices - 1) * s4];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (z0 + 16 * u < n_slices) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  sh[t][el] = acc;
  __syncthreads();
  if (t == 0 && e < n4) {
    float4 r = sh[0][el];
#pragma unroll
    for (int u = 1; u < 16; ++u) { const float4 v = sh[u][el]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
    reinterpret_cast<float4*>(out)[e] = r;
  }
}

// the ordered sum by itself (smx_headfused.hip: the per-workgroup d d slabs of the fused output head)
int launch_bigk_reduce(hipStream_t st, const float* part, long slab_stride, int n_slices, long n4, float* out) {
  if (!part || !out || n_slices <= 0 || n4 <= 0 || (slab_stride % 4)) { set_error("bigk_reduce: bad arguments"); return SMX_ERR_INVALID; }
  hipLaunchKernelGGL(bigk_reduce_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, part, slab_stride, n_slices, n4, out);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// slices: whole 32-deep stages per slice, about one slice per CU (at most `max_slices`)
int bigk_slices(long K, int max_slices, int* k_chunk) {
  static const int cap = (int)tuning("bigk_slices", 0);   // (sweeps)
  if (cap > 0) max_slices = std::min(max_slices, cap);
  long chunk = (K + max_slices - 1) / max_slices;
  chunk = std::max<long>(64, (chunk + 31) / 32 * 32);
  if (k_chunk) *k_chunk = (int)chunk;
  return (int)((K + chunk - 1) / chunk);
}

bool bigk_supported(const BigKArgs& a) {
  return a.A && a.Bm && a.part && a.out && a.M > 0 && a.N >= 32 && a.N % 32 == 0 && a.K % 32 == 0 && a.K >= 4096 && (a.lda % 8) == 0 && (a.ldb % 4) == 0 &&
         (a.ldc % 4) == 0 && (a.slab_stride % 4) == 0 && ((long)a.M * a.ldc) <= a.slab_stride;
}

int launch_bigk(hipStream_t st, const BigKArgs& a_in) {
  BigKArgs a = a_in;
  if (!bigk_supported(a)) { set_error("bigk: unsupported shapes"); return SMX_ERR_INVALID; }
  if (a.colmajor && (a.M > 128 || a.N > 128 || a.slab_stride < (long)a.N * 128)) { set_error("bigk: column-major slabs take one 128 x 128 output"); return SMX_ERR_INVALID; }
  if (a.n_slices <= 0 || a.k_chunk % 32 || (long)a.n_slices * a.k_chunk < a.K) { set_error("bigk: bad slicing"); return SMX_ERR_INVALID; }
  const dim3 grid((unsigned)a.n_slices, (unsigned)((a.M + 127) / 128), (unsigned)((a.N + 127) / 128));
  static const int stages_env = (int)tuning("bigk_stages", 0);
  a.stages = (stages_env >= 2 && stages_env <= BIGK_STAGES) ? stages_env : SMX_BIGK_STAGES_DEFAULT;
  const size_t lds = (size_t)a.stages * ((a.a_u16 ? 128 * 32 * 2 : 128 * 32 * 4) + 128 * 32 * 4);
#define SMX_BIGK_LAUNCH(U, KM, L)                                                                                         \
  do {                                                                                                                  \
    static bool raised = false;                                                                                         \
    if (!raised) {                                                                                   \
      SMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bigk_kernel<U, KM, L>), hipFuncAttributeMaxDynamicSharedMemorySize, BIGK_STAGES * 2 * 128 * 32 * 4)); \
      raised = true;                                                                                                    \
    }                                                                                                                   \
    hipLaunchKernelGGL((bigk_kernel<U, KM, L>), grid, dim3(512), lds, st, a);                                              \
  } while (0)
  // the two uses: the encoder front (gathered counts, log1p, W stored [K][N]) and d d of the head (plain operands, W stored [N][K])
  if (a.b_kmajor && a.log1p) { if (a.a_u16) SMX_BIGK_LAUNCH(1, 1, 1); else SMX_BIGK_LAUNCH(0, 1, 1); }
  else if (a.b_kmajor && !a.a_u16) SMX_BIGK_LAUNCH(0, 1, 0);
  else if (!a.b_kmajor && !a.a_u16 && !a.log1p) SMX_BIGK_LAUNCH(0, 0, 0);
  else { set_error("bigk: operand form not built"); return SMX_ERR_INVALID; }
#undef SMX_BIGK_LAUNCH
  SMX_HIP(hipGetLastError());
  if (a.colmajor) return SMX_OK;   // (the consumer sums the slabs)
  const long n4 = ((long)a.M * a.ldc) >> 2;
  hipLaunchKernelGGL(bigk_reduce_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, a.part, a.slab_stride, a.n_slices, n4, a.out);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

}  // namespace smx
