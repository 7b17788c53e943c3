// smx_data.hip -- kernels over the HBM-resident cells x genes matrix: per-row constants
// and library-size statistics (get_library_size, sisua/data/utils.py:231-263), and the 'binomial'
// artificial corruption (apply_artificial_corruption, sisua/data/utils.py:168-228) with the counter RNG.
//
// All of it is once-per-dataset streaming work: one pass over X per launch, a wave per row (stats) or a
// workgroup-strided sweep (corruption); nothing here is on the per-step path.
#include <algorithm>

#include "smx_internal.h"
#include "../../include/sisua_hip.h"

namespace smx {

__device__ inline double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// One wave per row.  lgx1[row] = sum_g lgamma(x+1) (the likelihood's data-only constant);
// logcount[row] = log(float(sum_g x) + 1e-8) in fp32 as NumPy evaluates it on the float32 matrix.
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ X, int u16, long ld, long N, int G,
                                                        float* __restrict__ lgx1, double* __restrict__ logcount) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const int lane = threadIdx.x & 63;
  const float* r = X + row * ld;
  const uint16_t* rh = reinterpret_cast<const uint16_t*>(X) + row * ld;
  double tot = 0.0, lg = 0.0;
  for (int g = lane * 4; g < G; g += 256) {
    float e[4];
    if (u16) { const ushort4 h = *reinterpret_cast<const ushort4*>(rh + g); e[0] = h.x; e[1] = h.y; e[2] = h.z; e[3] = h.w; }
    else { const float4 v = *reinterpret_cast<const float4*>(r + g); e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w; }   // ld is padded with zeros
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (g + q < G) {
        tot += (double)e[q];
        if (e[q] > 0.f) lg += lgamma((double)e[q] + 1.0);
      }
  }
  tot = wave_sum_f64(tot);
  lg = wave_sum_f64(lg);
  if (lane == 0) {
    lgx1[row] = (float)lg;
    if (logcount) logcount[row] = (double)logf((float)tot + 1e-8f);
  }
}

// ---- compact sparse store (CSR: SURVEY.md 8f-2, the dense float32 memmap semantics of sisua/data/utils.py:401-452
// over 7-12 % of its bytes) ------------------------------------------------------------------------------------------
// lgx1 of every resident row, one wave per row (the data-only constant of the likelihood, as row_stats_kernel)
__global__ __launch_bounds__(256) void csr_row_stats_kernel(const int64_t* __restrict__ indptr, const float* __restrict__ vals,
                                                            long N, float* __restrict__ lgx1) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const int lane = threadIdx.x & 63;
  double lg = 0.0;
  for (int64_t i = indptr[row] + lane; i < indptr[row + 1]; i += 64) {
    const float v = vals[i];
    if (v > 0.f) lg += lgamma((double)v + 1.0);
  }
  lg = wave_sum_f64(lg);
  if (lane == 0) lgx1[row] = (float)lg;
}
int launch_csr_row_stats(hipStream_t st, const int64_t* indptr, const float* vals, long N, float* lgx1) {
  hipLaunchKernelGGL(csr_row_stats_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, indptr, vals, N, lgx1);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// The rows of a minibatch as a dense [B][ld] float32 tile (zeroed by the caller): the per-step reader of the sparse
// store.  One wave per row; every consumer of X then reads the tile exactly as it reads a host batch.
__global__ __launch_bounds__(256) void csr_expand_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols,
                                                         const float* __restrict__ vals, const int32_t* __restrict__ rows, long row0,
                                                         int B, long ld, float* __restrict__ out) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  const long src = rows ? rows[b] : row0 + b;
  float* o = out + (long)b * ld;
  for (int64_t i = indptr[src] + lane; i < indptr[src + 1]; i += 64) o[cols[i]] = vals[i];
}
int launch_csr_expand(hipStream_t st, const int64_t* indptr, const int32_t* cols, const float* vals, const int32_t* rows, long row0,
                      int B, long ld, float* out) {
  SMX_HIP(hipMemsetAsync(out, 0, (size_t)B * (size_t)ld * sizeof(float), st));
  hipLaunchKernelGGL(csr_expand_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, indptr, cols, vals, rows, row0, B, ld, out);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// ---- on-device generator of the synthetic scaling configuration (BASELINE.json configs[4]; SURVEY.md 8d: "x =
// floor(LogNormal(mu_g, 1)) thinned to ~93 % zeros, generated on-device per shard from (seed, rank)") --------------------
// Entry (cell c, gene g) is a pure function of (seed, GLOBAL cell id, g) -- a shard is the rows [rank n, (rank + 1) n) of
// ONE virtual matrix, whatever the number of ranks:
//   block q = g / 2 of cell c: w = philox(counter = (q, cell id, 0, ST_GENERATE), key = seed)
//   (n0, n1) = Box-Muller(w.x, w.y);  keep_e = u24(w.z / w.w) < density;  x = keep ? min(floor(exp(mu_g + n_e)), 65535) : 0
//   mu_g = 0.5 * standard normal of gene g (gene_mu_kernel: philox(counter = (g / 4, 0xFFFFFFFF, 0, ST_GENERATE_MU)))
//   gene 0 of every cell is at least 1 (no all-zero cell).
// oracle/sisua_oracle.py:generate_lognormal_rows restates it; the fast v_log / v_sin / v_cos / v_exp forms leave ~1e-6
// relative on exp(.), so a value within that of an integer may floor differently there (tests/test_gpu_dataset.py allows
// exactly those).
__global__ void gene_mu_kernel(uint32_t k0, uint32_t k1, int G, float* __restrict__ mu) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q * 4 >= G) return;
  const float4 n = normal4(philox4x32_10((uint32_t)q, 0xFFFFFFFFu, 0u, (uint32_t)ST_GENERATE_MU, k0, k1));
  const float v[4] = {n.x, n.y, n.z, n.w};
  for (int e = 0; e < 4; ++e)
    if (q * 4 + e < G) mu[q * 4 + e] = 0.5f * v[e];
}
// one thread per pair of genes; a workgroup walks one row (rows are gridDim.y-strided: N up to 2^31 - 1)
template <int U16>
__global__ __launch_bounds__(256) void generate_lognormal_kernel(void* __restrict__ Xv, long ld, long N, int G, uint32_t k0, uint32_t k1,
                                                                 uint32_t cell_base, float density, const float* __restrict__ mu) {
  for (long row = blockIdx.y; row < N; row += gridDim.y) {
    const uint32_t cell = cell_base + (uint32_t)row;
    for (int q = blockIdx.x * 256 + threadIdx.x; 2 * q < (int)ld; q += gridDim.x * 256) {   // the padded columns too: zeros beyond G
      const U4 w = philox4x32_10((uint32_t)q, cell, 0u, (uint32_t)ST_GENERATE, k0, k1);
      const float u1 = ((float)(w.x >> 8) + 1.0f) * 5.9604644775390625e-08f;
      const float u2 = u24(w.y);
      const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
      const float nrm[2] = {r * __builtin_amdgcn_cosf(u2), r * __builtin_amdgcn_sinf(u2)};
      const float uk[2] = {u24(w.z), u24(w.w)};
      float x[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int g = 2 * q + e;
        const float v = g < G ? fminf(floorf(fexp(mu[min(g, G - 1)] + nrm[e])), 65535.f) : 0.f;
        x[e] = (g < G && uk[e] < density) ? v : 0.f;
        if (g == 0) x[e] = fmaxf(x[e], 1.f);
      }
      if (U16) reinterpret_cast<ushort2*>(reinterpret_cast<uint16_t*>(Xv) + row * ld)[q] = make_ushort2((unsigned short)x[0], (unsigned short)x[1]);
      else reinterpret_cast<float2*>(reinterpret_cast<float*>(Xv) + row * ld)[q] = make_float2(x[0], x[1]);
    }
  }
}
int launch_generate_lognormal(hipStream_t st, void* X, int u16, long ld, long N, int G, uint64_t seed, uint32_t cell_base, float density,
                              float* mu) {
  const uint32_t k0 = (uint32_t)(seed & 0xFFFFFFFFu), k1 = (uint32_t)(seed >> 32);
  hipLaunchKernelGGL(gene_mu_kernel, dim3((unsigned)((G + 1023) / 1024)), dim3(256), 0, st, k0, k1, G, mu);
  const unsigned gx = (unsigned)std::max(1, std::min(64, ((int)ld / 2 + 255) / 256));
  const unsigned gy = (unsigned)std::min<long>(N, 65535);
  if (u16) hipLaunchKernelGGL(generate_lognormal_kernel<1>, dim3(gx, gy), dim3(256), 0, st, X, ld, N, G, k0, k1, cell_base, density, mu);
  else hipLaunchKernelGGL(generate_lognormal_kernel<0>, dim3(gx, gy), dim3(256), 0, st, X, ld, N, G, k0, k1, cell_base, density, mu);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// mean and (population) variance of logcount[0..N): one workgroup, two passes, fp64
__global__ __launch_bounds__(1024) void library_moments_kernel(const double* __restrict__ logcount, long N,
                                                               double* __restrict__ stats) {
  __shared__ double sh[16];
  __shared__ double mean_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  double s = 0.0;
  for (long i = tid; i < N; i += 1024) s += logcount[i];
  s = wave_sum_f64(s);
  if (lane == 0) sh[w] = s;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += sh[i];
    mean_s = t / (double)N;
  }
  __syncthreads();
  const double mean = mean_s;
  double v = 0.0;
  for (long i = tid; i < N; i += 1024) { const double d = logcount[i] - mean; v += d * d; }
  v = wave_sum_f64(v);
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += sh[i];
    stats[0] = mean;
    stats[1] = t / (double)N;
  }
}

__global__ void library_fill_kernel(float* __restrict__ library, long N, const double* __restrict__ stats) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { library[2 * i] = (float)stats[0]; library[2 * i + 1] = (float)stats[1]; }
}

int launch_row_stats(hipStream_t st, const float* X, int u16, long ld, long N, int G, float* lgx1, double* logcount) {
  hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, X, u16, ld, N, G, lgx1, logcount);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

int launch_library_stats(hipStream_t st, const double* logcount, long N, double* stats, float* library) {
  hipLaunchKernelGGL(library_moments_kernel, dim3(1), dim3(1024), 0, st, logcount, N, stats);
  if (library)
    hipLaunchKernelGGL(library_fill_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, library, N, stats);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

// ---------------------------------------------------------------------------
// Corruption.  Entry (row, gene) with x > 0 carries the 64-bit key (w0 << 32 | w1) of
// philox(counter = (gene, cell_id, 0, SELECT)); the floor(dropout * nnz) smallest keys are corrupted.
// The threshold key is found by an 8-pass radix select (one 256-bin histogram of the next byte per pass).
// ---------------------------------------------------------------------------
__device__ inline float corrupt_get(const CorruptArgs& a, long row, int g) {
  return a.u16 ? (float)reinterpret_cast<const uint16_t*>(a.X)[row * a.ld + g] : a.X[row * a.ld + g];
}
__device__ inline void corrupt_put(const CorruptArgs& a, long row, int g, float v) {
  if (a.u16) reinterpret_cast<uint16_t*>(a.X)[row * a.ld + g] = (uint16_t)v;
  else a.X[row * a.ld + g] = v;
}

__device__ inline uint64_t corrupt_key(uint32_t k0, uint32_t k1, uint32_t cell, uint32_t gene) {
  const U4 w = philox4x32_10(gene, cell, 0u, (uint32_t)ST_CORRUPT_SELECT, k0, k1);
  return ((uint64_t)w.x << 32) | (uint64_t)w.y;
}

__global__ __launch_bounds__(256) void corrupt_hist_kernel(CorruptArgs a, int pass) {
  __shared__ unsigned int h[256];
  h[threadIdx.x] = 0u;
  __syncthreads();
  const int shift = 56 - 8 * pass;
  for (long row = blockIdx.x; row < a.N; row += gridDim.x) {
    const uint32_t cell = a.cell_base + (uint32_t)row;
    for (int g = threadIdx.x; g < a.G; g += 256) {
      if (!(corrupt_get(a, row, g) > 0.f)) continue;
      const uint64_t key = corrupt_key(a.k0, a.k1, cell, (uint32_t)g);
      if (pass > 0 && (key >> (shift + 8)) != (a.prefix >> (shift + 8))) continue;
      atomicAdd(&h[(unsigned)((key >> shift) & 0xFFu)], 1u);
    }
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&a.hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// x <- Binomial(n = x, p): trial t succeeds iff word (t % 4) of
// philox(counter = (gene, cell_id, 0, BINOMIAL | (t / 4) << 8)) < thr_binom (= floor(p * 2^32)).
__global__ __launch_bounds__(256) void corrupt_apply_kernel(CorruptArgs a) {
  unsigned long long mine = 0ull;
  for (long row = blockIdx.x; row < a.N; row += gridDim.x) {
    const uint32_t cell = a.cell_base + (uint32_t)row;
    for (int g = threadIdx.x; g < a.G; g += 256) {
      const float x = corrupt_get(a, row, g);
      if (!(x > 0.f)) continue;
      if (corrupt_key(a.k0, a.k1, cell, (uint32_t)g) > a.prefix) continue;
      const long n = (long)x;
      long got = 0;
      for (long blk = 0; 4 * blk < n; ++blk) {
        const U4 w = philox4x32_10((uint32_t)g, cell, 0u, (uint32_t)ST_CORRUPT_BINOMIAL | ((uint32_t)blk << 8), a.k0, a.k1);
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (4 * blk + q < n && (uint64_t)ww[q] < a.thr_binom) ++got;
      }
      corrupt_put(a, row, g, (float)got);
      ++mine;
    }
  }
  if (mine) atomicAdd(a.hist, mine);
}

int launch_corrupt_hist(hipStream_t st, const CorruptArgs& a, int pass) {
  const unsigned blocks = (unsigned)(a.N < 4096 ? a.N : 4096);
  hipLaunchKernelGGL(corrupt_hist_kernel, dim3(blocks), dim3(256), 0, st, a, pass);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

int launch_corrupt_apply(hipStream_t st, const CorruptArgs& a) {
  const unsigned blocks = (unsigned)(a.N < 4096 ? a.N : 4096);
  hipLaunchKernelGGL(corrupt_apply_kernel, dim3(blocks), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}

}  // namespace smx
