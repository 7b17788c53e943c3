// smx_kapi.hip -- kernel-level entry points (parity tests of single kernels): smx_k_count_llk, smx_k_adam, smx_k_gemm, smx_k_head_fused, smx_k_noise.
#include "smx_model.h"
#include <hiprand/hiprand_kernel.h>

namespace smx {
// The library's noise function against hiprand's own Philox generator (row N-1, "sampling from a hiprand state per wavefront"): thread i
// builds a hiprandStatePhilox4_32_10_t with hiprand_init(seed, subsequence = (c2, c3) = (step, stream | sample << 8), offset =
// 4 * (c0, c1) = 4 * (column block, cell id)) and draws ONE hiprand4 -- and evaluates philox4x32_10(c0, c1, c2, c3, seed) as the kernels do.
__global__ void hiprand_probe_kernel(uint64_t seed, int n, const uint32_t* c, uint32_t* ours, uint32_t* theirs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t c0 = c[4 * i], c1 = c[4 * i + 1], c2 = c[4 * i + 2], c3 = c[4 * i + 3];
  const U4 w = philox4x32_10(c0, c1, c2, c3, (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32));
  ours[4 * i] = w.x; ours[4 * i + 1] = w.y; ours[4 * i + 2] = w.z; ours[4 * i + 3] = w.w;
  hiprandStatePhilox4_32_10_t st;
  hiprand_init(seed, (unsigned long long)c2 | ((unsigned long long)c3 << 32), 4ull * ((unsigned long long)c0 | ((unsigned long long)c1 << 32)), &st);
  const uint4 r = hiprand4(&st);
  theirs[4 * i] = r.x; theirs[4 * i + 1] = r.y; theirs[4 * i + 2] = r.z; theirs[4 * i + 3] = r.w;
}
// words of a and b that differ, added to *n (the stress entry point below)
__global__ void count_diff_kernel(const uint32_t* a, const uint32_t* b, long n, unsigned* out) {
  unsigned mine = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) mine += a[i] != b[i];
  if (mine) atomicAdd(out, mine);
}
}  // namespace smx

extern "C" {

int smx_k_hiprand(uint64_t seed, int32_t n, const uint32_t* counters, uint32_t* ours, uint32_t* hiprand_words) {
  SMX_REQUIRE(n > 0 && counters && ours && hiprand_words, "bad arguments");
  uint32_t *dC = nullptr, *dO = nullptr, *dH = nullptr;
  int rc;
  if ((rc = dmalloc(&dC, (size_t)4 * n)) || (rc = dmalloc(&dO, (size_t)4 * n)) || (rc = dmalloc(&dH, (size_t)4 * n))) return rc;
  SMX_HIP(hipMemcpy(dC, counters, (size_t)16 * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(smx::hiprand_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, seed, (int)n, dC, dO, dH);
  SMX_HIP(hipGetLastError());
  SMX_HIP(hipDeviceSynchronize());
  SMX_HIP(hipMemcpy(ours, dO, (size_t)16 * n, hipMemcpyDeviceToHost));
  SMX_HIP(hipMemcpy(hiprand_words, dH, (size_t)16 * n, hipMemcpyDeviceToHost));
  hipFree(dC); hipFree(dO); hipFree(dH);
  return SMX_OK;
}

// ---- kernel-level entry points ------------------------------------------------
int smx_k_count_llk(int likelihood, int direct, const float* x, const float* planes, int32_t B, int32_t G, float* llk,
                    float* grads) {
  SMX_REQUIRE(x && planes && llk && B > 0 && G > 0, "bad arguments");
  const int k = llk_planes(likelihood);
  const int Gp = round_up(G, 32);
  const int nch = loss_chunks(Gp, B);
  float *dX = nullptr, *dPl = nullptr, *dG = nullptr, *dPart = nullptr;
  int rc;
  if ((rc = dmalloc(&dX, (size_t)B * Gp)) || (rc = dmalloc(&dPl, (size_t)B * k * Gp)) || (rc = dmalloc(&dG, (size_t)B * k * Gp)) ||
      (rc = dmalloc(&dPart, (size_t)B * nch)))
    return rc;
  SMX_HIP(hipMemcpy2D(dX, (size_t)Gp * 4, x, (size_t)G * 4, (size_t)G * 4, (size_t)B, hipMemcpyHostToDevice));
  for (int c = 0; c < k; ++c)
    SMX_HIP(hipMemcpy2D(dPl + (size_t)c * Gp, (size_t)k * Gp * 4, planes + (size_t)c * B * G, (size_t)G * 4, (size_t)G * 4,
                        (size_t)B, hipMemcpyHostToDevice));
  LossArgs lo;
  lo.likelihood = likelihood; lo.direct = direct; lo.backward = grads != nullptr;
  lo.X = dX; lo.ldx = Gp; lo.P = dPl; lo.ldp = (long)k * Gp; lo.plane_stride = Gp; lo.dP = dG; lo.llk_part = dPart;
  lo.B = B; lo.G = G; lo.Gp = Gp; lo.grad_scale = 1.f;
  rc = launch_count_loss(nullptr, lo);
  if (rc == SMX_OK) {
    std::vector<float> part((size_t)B * nch);
    SMX_HIP(hipDeviceSynchronize());
    SMX_HIP(hipMemcpy(part.data(), dPart, part.size() * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b) {
      double s = 0.0;
      for (int c = 0; c < nch; ++c) s += part[(size_t)b * nch + c];
      for (int g = 0; g < G; ++g) { const float v = x[(size_t)b * G + g]; if (v > 0.f) s -= lgamma((double)v + 1.0); }
      llk[b] = (float)s;
    }
    if (grads)
      for (int c = 0; c < k; ++c)
        SMX_HIP(hipMemcpy2D(grads + (size_t)c * B * G, (size_t)G * 4, dG + (size_t)c * Gp, (size_t)k * Gp * 4, (size_t)G * 4,
                            (size_t)B, hipMemcpyDeviceToHost));
  }
  hipFree(dX); hipFree(dPl); hipFree(dG); hipFree(dPart);
  return rc;
}

int smx_k_adam(int32_t n_tensors, const int32_t* sizes, float* params, const float* grads, float* mom, float* vel,
               int32_t step, float lr, float beta1, float beta2, float eps, float clipnorm, float* norms) {
  SMX_REQUIRE(n_tensors > 0 && n_tensors <= SMX_MAX_TENSORS && sizes && params && grads && mom && vel && step >= 1, "bad arguments");
  // the model's own layout: every tensor padded to a multiple of 64 floats, 4096-float optimiser chunks
  std::vector<size_t> off((size_t)n_tensors), pad((size_t)n_tensors);
  std::vector<OptChunk> chunks;
  size_t total = 0;
  const int CH = 4096;
  for (int t = 0; t < n_tensors; ++t) {
    SMX_REQUIRE(sizes[t] > 0, "empty tensor");
    off[t] = total; pad[t] = ((size_t)sizes[t] + 63) / 64 * 64;
    const int first = (int)chunks.size(), n = (int)((pad[t] + CH - 1) / CH);
    for (int i = 0; i < n; ++i) {
      OptChunk c;
      memset(&c, 0, sizeof(c));
      c.tensor = t; c.offset = (int)(off[t] + (size_t)i * CH);
      c.count = (int)((size_t)(i + 1) * CH <= pad[t] ? CH : pad[t] - (size_t)i * CH);
      c.first_chunk = first; c.n_chunks = n; c.tensor_count = (int32_t)pad[t];
      chunks.push_back(c);
    }
    total += pad[t];
  }
  float *dP = nullptr, *dG = nullptr, *dM = nullptr, *dV = nullptr, *dPart = nullptr, *dNorm = nullptr;
  OptChunk* dCh = nullptr; StepState* dSt = nullptr;
  int rc;
  if ((rc = dmalloc(&dP, total)) || (rc = dmalloc(&dG, total)) || (rc = dmalloc(&dM, total)) || (rc = dmalloc(&dV, total)) ||
      (rc = dmalloc(&dPart, chunks.size())) || (rc = dmalloc(&dNorm, (size_t)n_tensors)) || (rc = dmalloc(&dCh, chunks.size())) ||
      (rc = dmalloc(&dSt, (size_t)3)))
    return rc;
  auto put = [&](float* dst, const float* src) -> int {
    size_t lo = 0;
    for (int t = 0; t < n_tensors; ++t) {
      SMX_HIP(hipMemcpy(dst + off[t], src + lo, (size_t)sizes[t] * sizeof(float), hipMemcpyHostToDevice));
      lo += (size_t)sizes[t];
    }
    return SMX_OK;
  };
  auto get = [&](float* dst, const float* src) -> int {
    size_t lo = 0;
    for (int t = 0; t < n_tensors; ++t) {
      SMX_HIP(hipMemcpy(dst + lo, src + off[t], (size_t)sizes[t] * sizeof(float), hipMemcpyDeviceToHost));
      lo += (size_t)sizes[t];
    }
    return SMX_OK;
  };
  rc = put(dP, params); if (rc == SMX_OK) rc = put(dG, grads); if (rc == SMX_OK) rc = put(dM, mom); if (rc == SMX_OK) rc = put(dV, vel);
  if (rc == SMX_OK && hipMemcpy(dCh, chunks.data(), chunks.size() * sizeof(OptChunk), hipMemcpyHostToDevice) != hipSuccess) rc = SMX_ERR_HIP;
  StepState st3[3];
  memset(st3, 0, sizeof(st3));
  st3[2].next = (uint32_t)(step - 1);   // optimiser steps completed so far
  if (rc == SMX_OK && hipMemcpy(dSt, st3, sizeof(st3), hipMemcpyHostToDevice) != hipSuccess) rc = SMX_ERR_HIP;
  // the step's scalars exactly as a training step prepares them (bias-corrected step size on the device)
  if (rc == SMX_OK) rc = launch_step_begin(nullptr, dSt + 2, dSt, nullptr, nullptr, 0, 0, 0u, lr, beta1, beta2);
  if (rc == SMX_OK) {
    AdamArgs a;
    a.params = dP; a.grads = dG; a.m = dM; a.v = dV; a.chunks = dCh; a.n_chunks = (int)chunks.size(); a.n_launch = a.n_chunks; a.gap_from = a.n_chunks; a.gap_len = 0;
    a.partial = dPart; a.tensor_norm = dNorm; a.use_sq = 0; a.state = dSt;
    a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.clipnorm = clipnorm; a.grad_scale = 1.f; a.lr = lr;
    rc = launch_adam(nullptr, a);
  }
  if (rc == SMX_OK && hipDeviceSynchronize() != hipSuccess) { set_error("k_adam: device synchronize failed"); rc = SMX_ERR_HIP; }
  if (rc == SMX_OK) rc = get(params, dP);
  if (rc == SMX_OK) rc = get(mom, dM);
  if (rc == SMX_OK) rc = get(vel, dV);
  if (rc == SMX_OK && norms && hipMemcpy(norms, dNorm, (size_t)n_tensors * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = SMX_ERR_HIP;
  hipFree(dP); hipFree(dG); hipFree(dM); hipFree(dV); hipFree(dPart); hipFree(dNorm); hipFree(dCh); hipFree(dSt);
  return rc;
}

int smx_k_gemm(int transA, int transB, const float* A, const float* B, int32_t M, int32_t N, int32_t K, int32_t split_k,
               int32_t tile_cfg, float* C) {
  SMX_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "bad arguments");
  // tile_cfg 100 / 101 / 102: the bf16 x 3 forms outside the LDS-tiled kernel -- smx_dgemm.hip (A [M][K]; K padded to 32 with
  // zeros), the 32 x 32-tile weight-gradient kernel and the panel form of smx_panel.h (both: A [K][M], Bm [K][N], M padded to 32)
  const bool direct = tile_cfg == 100, wg = tile_cfg == 101 || tile_cfg == 102;
  SMX_REQUIRE(!direct || !transA, "tile 100 (dgemm) takes A as [M][K]");
  SMX_REQUIRE(!wg || (transA && !transB), "tiles 101 / 102 (weight-gradient forms) take A as [K][M] and B as [K][N]");
  SMX_REQUIRE(tile_cfg != 102 || N <= 128, "tile 102 (panel form): N <= 128");
  // pad to the library's internal conventions: feature axes to 32, batch axes free
  const int Np = round_up(N, 32);
  const int Kp = direct ? round_up(K, 32) : round_up(K, 4), Mp = wg ? round_up(M, 32) : round_up(M, 4);
  const int lda = transA ? Mp : Kp, a_rows = transA ? K : M, a_cols = transA ? M : K;
  const int ldb = transB ? Kp : Np, b_rows = transB ? N : K, b_cols = transB ? K : N;
  float *dA = nullptr, *dB = nullptr, *dC = nullptr;
  int rc;
  const int S = split_k < 1 ? 1 : split_k;
  if ((rc = dmalloc(&dA, (size_t)a_rows * lda)) || (rc = dmalloc(&dB, (size_t)round_up(b_rows, 32) * ldb)) ||
      (rc = dmalloc(&dC, (size_t)S * M * Np)))
    return rc;
  SMX_HIP(hipMemcpy2D(dA, (size_t)lda * 4, A, (size_t)a_cols * 4, (size_t)a_cols * 4, (size_t)a_rows, hipMemcpyHostToDevice));
  SMX_HIP(hipMemcpy2D(dB, (size_t)ldb * 4, B, (size_t)b_cols * 4, (size_t)b_cols * 4, (size_t)b_rows, hipMemcpyHostToDevice));
  GemmArgs g;
  g.A = dA; g.lda = lda; g.a_kmajor = transA; g.B = dB; g.ldb = ldb; g.b_nmajor = transB;
  g.C = dC; g.ldc = Np; g.slab_stride = (long)M * Np; g.M = transA ? Mp : M; g.N = Np; g.K = (transA) ? K : Kp;
  if (transA) g.M = Mp;
  g.split_k = S; g.tile = tile_cfg;
  int eff = 1;
  // rows of C beyond M (when M was padded for k-major A) are never stored: allocate for Mp
  if (transA && Mp != M) { hipFree(dC); dC = nullptr; if ((rc = dmalloc(&dC, (size_t)S * Mp * Np))) return rc; g.C = dC; g.slab_stride = (long)Mp * Np; }
  if (direct) {
    g.K = Kp;   // (the padding of A and B is zero: dmalloc clears)
    SMX_REQUIRE(dgemm_supported(g), "tile 100 (dgemm): K >= 512 after padding to 32, split_k = 1");
    rc = launch_dgemm(nullptr, g);
  } else if (wg) {
    g.split_k = 1; g.panel_hint = tile_cfg == 102;
    SMX_REQUIRE(S == 1 && wgrad_supported(g, K), "tiles 101 / 102: split_k = 1");
    rc = launch_wgrad_group(nullptr, &g, 1, K, 1);
  } else
  rc = launch_gemm(nullptr, g, &eff);
  if (rc == SMX_OK && tuning("kgemm_reps", 0) > 0 && !direct && !wg) {  // diagnostic: average launch time of this shape / tile
    const int reps = (int)tuning("kgemm_reps", 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch_gemm(nullptr, g, nullptr);
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) launch_gemm(nullptr, g, nullptr);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    fprintf(stderr, "k_gemm tA=%d tB=%d M=%d N=%d K=%d split=%d tile=%d: %.2f us\n", transA, transB, M, N, K, eff, tile_cfg,
            1e3f * ms / reps);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  if (rc == SMX_OK) {
    SMX_HIP(hipDeviceSynchronize());
    const int rowsC = transA ? Mp : M;
    std::vector<float> h((size_t)eff * rowsC * Np);
    SMX_HIP(hipMemcpy(h.data(), dC, h.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < M; ++i)
      for (int j = 0; j < N; ++j) {
        float s = 0.f;
        for (int z = 0; z < eff; ++z) s += h[((size_t)z * rowsC + i) * Np + j];
        C[(size_t)i * N + j] = s;
      }
  }
  hipFree(dA); hipFree(dB); hipFree(dC);
  return rc;
}

// The fused output head of smx_headfused.hip over host arrays (H = 128 decoder columns, B <= 128 cells, G >= 4096 genes): W [128][k][G]
// (plane-major columns, as the model's head), bias [k][G], x [B][G] (u16 != 0: the counts travel through the uint16 store).  Out: llk [B], dW [128][k][G], db [k][G], dd [B][128], sumsq (of dW; may be NULL); us (may be
// NULL): average device time of `reps` launches of the fused kernel (HIP events; without the reduce launch).
int smx_k_head_fused(int likelihood, int u16, const float* x, const float* d, const float* W, const float* bias, int32_t B, int32_t G,
                     float grad_scale, int32_t reps, float* llk, float* dW, float* db, float* dd, float* sumsq, float* us) {
  SMX_REQUIRE(x && d && W && bias && llk && dW && db && dd && B > 0 && G > 0, "bad arguments");
  const int k = llk_planes(likelihood);
  const int Gp = round_up(G, 32), H = 128;
  SMX_REQUIRE(head_fused_supported(B, H, Gp, k), "head_fused: unsupported shape (B <= 256, G >= 4096 after padding, 2 or 3 planes)");
  const int grid = head_fused_grid(Gp), n_gt = head_fused_chunks(Gp);
  float *dX = nullptr, *dD = nullptr, *dWt = nullptr, *dBias = nullptr, *dGW = nullptr, *dGb = nullptr, *dPart = nullptr, *dLl = nullptr, *dSq = nullptr, *dDd = nullptr;
  uint16_t* dX16 = nullptr; float* dTab = nullptr;
  int rc;
  if ((rc = dmalloc(&dTab, (size_t)SMX_HEAD_FUSED_TAB_BYTES / 4)) || (rc = dmalloc(&dX, (size_t)B * Gp)) || (rc = dmalloc(&dD, (size_t)B * H)) || (rc = dmalloc(&dWt, (size_t)H * k * Gp)) || (rc = dmalloc(&dBias, (size_t)k * Gp)) ||
      (rc = dmalloc(&dGW, (size_t)H * k * Gp)) || (rc = dmalloc(&dGb, (size_t)k * Gp)) || (rc = dmalloc(&dPart, (size_t)grid * B * H)) ||
      (rc = dmalloc(&dLl, (size_t)B * n_gt)) || (rc = dmalloc(&dSq, (size_t)grid * 8)) || (rc = dmalloc(&dDd, (size_t)B * H)) || (rc = dmalloc(&dX16, (size_t)B * Gp)))
    return rc;
  SMX_HIP(hipMemcpy2D(dX, (size_t)Gp * 4, x, (size_t)G * 4, (size_t)G * 4, (size_t)B, hipMemcpyHostToDevice));
  if (u16) {
    std::vector<uint16_t> h16((size_t)B * Gp, 0);
    for (int b = 0; b < B; ++b)
      for (int g = 0; g < G; ++g) h16[(size_t)b * Gp + g] = (uint16_t)x[(size_t)b * G + g];
    SMX_HIP(hipMemcpy(dX16, h16.data(), h16.size() * 2, hipMemcpyHostToDevice));
  }
  SMX_HIP(hipMemcpy(dD, d, (size_t)B * H * 4, hipMemcpyHostToDevice));
  for (int c = 0; c < k; ++c) {
    SMX_HIP(hipMemcpy2D(dWt + (size_t)c * Gp, (size_t)k * Gp * 4, W + (size_t)c * G, (size_t)k * G * 4, (size_t)G * 4, (size_t)H, hipMemcpyHostToDevice));
    SMX_HIP(hipMemcpy(dBias + (size_t)c * Gp, bias + (size_t)c * G, (size_t)G * 4, hipMemcpyHostToDevice));
  }
  HeadFusedArgs a;
  a.D = dD; a.ldd = H; a.W = dWt; a.ldw = (long)k * Gp; a.bias = dBias;
  a.X = u16 ? (const void*)dX16 : (const void*)dX; a.ldx = Gp; a.x_u16 = u16 ? 1 : 0;
  a.dW = dGW; a.db = dGb; a.part = dPart; a.slab_stride = (long)B * H; a.llk_part = dLl; a.sq_part = dSq; a.dtab = dTab;
  a.B = B; a.G = G; a.Gp = Gp; a.likelihood = likelihood; a.grad_scale = grad_scale;
  int n_sq = 0;
  long long* dDbg = nullptr;
  if (tuning("hf_dbg", 0) > 0) { if ((rc = dmalloc(&dDbg, (size_t)128))) return rc; a.dbg = dDbg; }
  int n_slabs = 0;
  rc = launch_head_fused(nullptr, a, &n_slabs, &n_sq);
  if (rc == SMX_OK) rc = launch_head_fused_reduce(nullptr, a, n_slabs, dDd);
  if (rc == SMX_OK && reps > 0 && us) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps && rc == SMX_OK; ++i) rc = launch_head_fused(nullptr, a, &n_slabs, &n_sq);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    *us = 1e3f * ms / (float)reps;
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  if (rc == SMX_OK && hipDeviceSynchronize() != hipSuccess) { set_error("k_head_fused: device synchronize failed"); rc = SMX_ERR_HIP; }
  if (rc == SMX_OK && dDbg) {
    long long h[128];
    hipMemcpy(h, dDbg, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 2; ++b) {
      fprintf(stderr, "hf stamps block %d: (prologue %lld)", b ? 100 : 0, h[64 * b] - h[64 * b + 63]);
      for (int i = 1; i < 62 && h[64 * b + i]; ++i) fprintf(stderr, " %lld", h[64 * b + i] - h[64 * b + i - 1]);
      fprintf(stderr, "\n");
    }

    hipFree(dDbg);
  }
  if (rc == SMX_OK) {
    std::vector<float> part((size_t)B * n_gt), sq((size_t)n_sq);
    SMX_HIP(hipMemcpy(part.data(), dLl, part.size() * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b) {
      double s = 0.0;
      for (int c = 0; c < n_gt; ++c) s += part[(size_t)b * n_gt + c];
      for (int g = 0; g < G; ++g) { const float v = x[(size_t)b * G + g]; if (v > 0.f) s -= lgamma((double)v + 1.0); }
      llk[b] = (float)s;
    }
    for (int c = 0; c < k; ++c) {
      SMX_HIP(hipMemcpy2D(dW + (size_t)c * G, (size_t)k * G * 4, dGW + (size_t)c * Gp, (size_t)k * Gp * 4, (size_t)G * 4, (size_t)H, hipMemcpyDeviceToHost));
      SMX_HIP(hipMemcpy(db + (size_t)c * G, dGb + (size_t)c * Gp, (size_t)G * 4, hipMemcpyDeviceToHost));
    }
    SMX_HIP(hipMemcpy(dd, dDd, (size_t)B * H * 4, hipMemcpyDeviceToHost));
    if (sumsq) {
      SMX_HIP(hipMemcpy(sq.data(), dSq, sq.size() * 4, hipMemcpyDeviceToHost));
      double s = 0.0;
      for (float v : sq) s += v;
      *sumsq = (float)s;
    }
  }
  hipFree(dX); hipFree(dD); hipFree(dWt); hipFree(dBias); hipFree(dGW); hipFree(dGb); hipFree(dPart); hipFree(dLl); hipFree(dSq); hipFree(dDd); hipFree(dX16); hipFree(dTab);
  return rc;
}

// The fused head launched `launches` times on the same inputs; every launch after the first is compared on the device, bit for bit, with
// what the first one left (dW, db, the d d slabs, the likelihood partials).  *n_differ = launches that differed anywhere, *first_word = index
// of the first differing word of the first differing launch within [dW | db | slabs | partials] (-1 if none).  Two timing-dependent wrong
// results of this kernel were hardware behaviour that no single launch shows reliably (tools/isa_lint.py); this is their regression test.
int smx_k_head_fused_stress(int likelihood, int u16, const float* x, const float* d, const float* W, const float* bias, int32_t B, int32_t G,
                            float grad_scale, int32_t launches, int32_t* n_differ, int64_t* first_word) {
  SMX_REQUIRE(x && d && W && bias && n_differ && B > 0 && B <= 128 && G > 0 && launches >= 2, "bad arguments");
  const int k = llk_planes(likelihood);
  const int Gp = round_up(G, 32), H = 128;
  SMX_REQUIRE(head_fused_supported(B, H, Gp, k), "head_fused: unsupported shape");
  const int grid = head_fused_grid(Gp), n_gt = head_fused_chunks(Gp);
  const size_t nW = (size_t)H * k * Gp, nb = (size_t)k * Gp, nP = (size_t)grid * B * H, nL = (size_t)B * n_gt, nOut = nW + nb + nP + nL;
  float *dX = nullptr, *dD = nullptr, *dWt = nullptr, *dBias = nullptr, *dOut = nullptr, *dRef = nullptr, *dSq = nullptr, *dTab = nullptr;
  uint16_t* dX16 = nullptr; unsigned* dN = nullptr;
  int rc;
  if ((rc = dmalloc(&dTab, (size_t)SMX_HEAD_FUSED_TAB_BYTES / 4)) || (rc = dmalloc(&dX, (size_t)B * Gp)) || (rc = dmalloc(&dD, (size_t)B * H)) || (rc = dmalloc(&dWt, nW)) ||
      (rc = dmalloc(&dBias, nb)) || (rc = dmalloc(&dOut, nOut)) || (rc = dmalloc(&dRef, nOut)) || (rc = dmalloc(&dSq, (size_t)grid * 8)) || (rc = dmalloc(&dX16, (size_t)B * Gp)) ||
      (rc = dmalloc(&dN, (size_t)1)))
    return rc;
  SMX_HIP(hipMemset(dX, 0, (size_t)B * Gp * 4)); SMX_HIP(hipMemset(dWt, 0, nW * 4)); SMX_HIP(hipMemset(dBias, 0, nb * 4));
  SMX_HIP(hipMemcpy2D(dX, (size_t)Gp * 4, x, (size_t)G * 4, (size_t)G * 4, (size_t)B, hipMemcpyHostToDevice));
  if (u16) {
    std::vector<uint16_t> h16((size_t)B * Gp, 0);
    for (int b = 0; b < B; ++b)
      for (int g = 0; g < G; ++g) h16[(size_t)b * Gp + g] = (uint16_t)x[(size_t)b * G + g];
    SMX_HIP(hipMemcpy(dX16, h16.data(), h16.size() * 2, hipMemcpyHostToDevice));
  }
  SMX_HIP(hipMemcpy(dD, d, (size_t)B * H * 4, hipMemcpyHostToDevice));
  for (int c = 0; c < k; ++c) {
    SMX_HIP(hipMemcpy2D(dWt + (size_t)c * Gp, (size_t)k * Gp * 4, W + (size_t)c * G, (size_t)k * G * 4, (size_t)G * 4, (size_t)H, hipMemcpyHostToDevice));
    SMX_HIP(hipMemcpy(dBias + (size_t)c * Gp, bias + (size_t)c * G, (size_t)G * 4, hipMemcpyHostToDevice));
  }
  HeadFusedArgs a;
  a.D = dD; a.ldd = H; a.W = dWt; a.ldw = (long)k * Gp; a.bias = dBias;
  a.X = u16 ? (const void*)dX16 : (const void*)dX; a.ldx = Gp; a.x_u16 = u16 ? 1 : 0;
  a.dW = dOut; a.db = dOut + nW; a.part = dOut + nW + nb; a.slab_stride = (long)B * H; a.llk_part = dOut + nW + nb + nP; a.sq_part = dSq; a.dtab = dTab;
  a.B = B; a.G = G; a.Gp = Gp; a.likelihood = likelihood; a.grad_scale = grad_scale;
  int n_sq = 0, n_slabs = 0, differ = 0;
  long long first = -1;
  std::vector<uint32_t> ho, hr;
  for (int it = 0; it < launches && rc == SMX_OK; ++it) {
    SMX_HIP(hipMemsetAsync(dOut, 0xFF, nOut * 4, nullptr));
    rc = launch_head_fused(nullptr, a, &n_slabs, &n_sq);
    if (rc != SMX_OK) break;
    if (it == 0) { SMX_HIP(hipMemcpyAsync(dRef, dOut, nOut * 4, hipMemcpyDeviceToDevice, nullptr)); continue; }
    SMX_HIP(hipMemsetAsync(dN, 0, 4, nullptr));
    hipLaunchKernelGGL(smx::count_diff_kernel, dim3(1024), dim3(256), 0, nullptr, reinterpret_cast<const uint32_t*>(dOut), reinterpret_cast<const uint32_t*>(dRef), (long)nOut, dN);
    unsigned n = 0;
    SMX_HIP(hipMemcpy(&n, dN, 4, hipMemcpyDeviceToHost));
    if (n) {
      ++differ;
      if (first < 0) {
        ho.resize(nOut); hr.resize(nOut);
        SMX_HIP(hipMemcpy(ho.data(), dOut, nOut * 4, hipMemcpyDeviceToHost)); SMX_HIP(hipMemcpy(hr.data(), dRef, nOut * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < nOut; ++i) if (ho[i] != hr[i]) { first = (long long)i; break; }
      }
    }
  }
  if (rc == SMX_OK && hipDeviceSynchronize() != hipSuccess) { set_error("k_head_fused_stress: device synchronize failed"); rc = SMX_ERR_HIP; }
  *n_differ = differ;
  if (first_word) *first_word = first;
  hipFree(dX); hipFree(dD); hipFree(dWt); hipFree(dBias); hipFree(dOut); hipFree(dRef); hipFree(dSq); hipFree(dX16); hipFree(dTab); hipFree(dN);
  return rc;
}

int smx_k_noise(uint64_t seed, int32_t stream, int32_t step, int32_t sample, const int64_t* cell_ids, int32_t B, int32_t width,
                float dropout_p, float* dropout_mult, float* normal) {
  SMX_REQUIRE(cell_ids && B > 0 && width > 0, "bad arguments");
  int64_t* dIds = nullptr; float *dM = nullptr, *dN = nullptr;
  int rc;
  if ((rc = dmalloc(&dIds, (size_t)B)) || (rc = dmalloc(&dM, (size_t)B * width)) || (rc = dmalloc(&dN, (size_t)B * width))) return rc;
  SMX_HIP(hipMemcpy(dIds, cell_ids, (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice));
  NoiseKey nk;
  nk.k0 = (uint32_t)(seed & 0xFFFFFFFFu); nk.k1 = (uint32_t)(seed >> 32); nk.step = (uint32_t)step;
  nk.stream = (uint32_t)((stream & 0xFF) | ((sample & 0xFFFFFF) << 8)); nk.step_ptr = nullptr;
  rc = launch_noise_probe(nullptr, nk, dIds, B, width, dropout_p, dM, dN);
  if (rc == SMX_OK) {
    SMX_HIP(hipDeviceSynchronize());
    if (dropout_mult) SMX_HIP(hipMemcpy(dropout_mult, dM, (size_t)B * width * 4, hipMemcpyDeviceToHost));
    if (normal) SMX_HIP(hipMemcpy(normal, dN, (size_t)B * width * 4, hipMemcpyDeviceToHost));
  }
  hipFree(dIds); hipFree(dM); hipFree(dN);
  return rc;
}

}  // extern "C"
