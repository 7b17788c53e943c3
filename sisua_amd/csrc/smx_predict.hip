// smx_predict.hip -- evaluation-mode forward passes handed back to the host: smx_forward, smx_forward_samples, smx_predict, smx_decode.
#include "smx_model.h"
#include "smx_loss.h"

namespace smx {

// The k parameter planes of a batch, device [B][k * Gp] -> caller's [k][B][G]: ONE contiguous copy into a pinned
// staging buffer (a pitched copy into pageable memory runs at ~1.6 GB/s here), then row copies on the host.
static int fetch_planes(smx_model* m, int B, float* x_params) {
  if (!x_params) return SMX_OK;
  const size_t n = (size_t)B * m->k * m->Gp;
  if (n > m->pinned_floats) {
    if (m->pinned) hipHostFree(m->pinned);
    m->pinned = nullptr; m->pinned_floats = 0;
    const size_t cap = (size_t)m->Bmax * m->k * m->Gp;
    SMX_HIP(hipHostMalloc((void**)&m->pinned, (cap > n ? cap : n) * sizeof(float), hipHostMallocDefault));
    m->pinned_floats = cap > n ? cap : n;
  }
  SMX_HIP(hipMemcpyAsync(m->pinned, m->P, n * sizeof(float), hipMemcpyDeviceToHost, m->st));
  SMX_HIP(hipStreamSynchronize(m->st));
  const size_t G = (size_t)m->G, ldp = (size_t)m->k * m->Gp;
  for (int ch = 0; ch < m->k; ++ch)
    for (int b = 0; b < B; ++b)
      memcpy(x_params + ((size_t)ch * B + b) * G, m->pinned + (size_t)b * ldp + (size_t)ch * m->Gp, G * sizeof(float));
  return SMX_OK;
}

// copy the results of the forward pass in flight back to the caller's arrays (any pointer may be NULL);
// y_off: element offset into every y_params[j] (draw index * batch * width)
static int fetch_forward(smx_model* m, int B, float* z_mean, float* z_scale, float* z_sample, float* l_mean, float* l_scale,
                         float* l_sample, float* x_params, float* const* y_params, size_t y_draw) {
  SMX_HIP(hipStreamSynchronize(m->st));
  const int D = m->D, Dp = m->Dp;
  const int lat_ld = m->lat_planes * Dp;
  std::vector<float> tmp;
  auto fetch2d = [&](float* dst, const float* src, int ld, int w) -> int {
    if (!dst) return SMX_OK;
    SMX_HIP(hipMemcpy2D(dst, (size_t)w * sizeof(float), src, (size_t)ld * sizeof(float), (size_t)w * sizeof(float), (size_t)B,
                        hipMemcpyDeviceToHost));
    return SMX_OK;
  };
  SMX_CHECK(fetch2d(z_mean, m->mixpost ? m->zmean : m->latbuf, m->mixpost ? Dp : lat_ld, D));   // (mixture-density posterior: the mixture's mean)
  if (m->stochastic) SMX_CHECK(fetch2d(z_scale, m->sig, Dp, D));
  SMX_CHECK(fetch2d(z_sample, m->z, Dp, D));
  if (m->scvi) {
    SMX_CHECK(fetch2d(l_mean, m->latlbuf, 32, 1));
    SMX_CHECK(fetch2d(l_scale, m->lsig, 1, 1));
    SMX_CHECK(fetch2d(l_sample, m->lsmp, 1, 1));
  }
  SMX_CHECK(fetch_planes(m, B, x_params));
  if (y_params) {
    for (int j = 0; j < m->n_heads; ++j) {
      if (!y_params[j]) continue;
      const int P = m->cfg.label_dim[j], Pp = m->lab_Pp[j], ld = m->tensors[m->t_labW[j]].ld;
      float* dst = y_params[j] + y_draw * (size_t)B * m->lab_ky[j] * P;
      tmp.resize((size_t)B * ld);
      SMX_HIP(hipMemcpy(tmp.data(), m->laby_raw[j], tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
      for (int b = 0; b < B; ++b)
        for (int c = 0; c < m->lab_ky[j]; ++c)
          memcpy(dst + ((size_t)b * m->lab_ky[j] + c) * P, &tmp[(size_t)b * ld + (size_t)c * Pp], sizeof(float) * P);
    }
  }
  return SMX_OK;
}

// SingleCellModel.predict over a whole host matrix in ONE call.  The batch loop runs here; after every forward pass one
// small launch packs what the caller asked for (parameter planes, latent moments, draws, label outputs) into device
// staging laid out like the caller's arrays for a CHUNK of cells (up to 128 MB), and each chunk leaves the device as a
// few large contiguous copies straight into its final place (48 GB/s into pageable memory as into pinned,
// tools/pcie_probe.hip).  No per-batch result arrays, no host re-packing, no concatenation afterwards -- and no swarm
// of small pitched copies (each a synchronous call: at batch 8 x 10 draws they cost 4x the whole old path).
// (n_rep repetitions of a job, e.g. the draws of a stacked pass: repetition q reads src + q src_rep, writes dst + q dst_rep)
struct PackJob { float* dst; long dpitch; const float* src; long spitch; int width; int height; int n_rep; long dst_rep; long src_rep; };
// A job list travels as a kernel argument (2 KB); a pass that needs more jobs than fit (MISA with four components: 3 latent +
// 3 planes + 12 label planes) launches the full list and starts the next one -- jobs are independent of each other.
#define SMX_PACK_MAX 32
struct PackJobs { int n; PackJob j[SMX_PACK_MAX]; };
__global__ __launch_bounds__(256) void pack_kernel(PackJobs jobs_by_value) {
  const PackJobs& J = *(const PackJobs*)__builtin_amdgcn_kernarg_segment_ptr();   // (run-time job index: no scratch copy)
  const PackJob& j = J.j[blockIdx.y];
  if ((int)blockIdx.z >= j.n_rep) return;
  const long total = (long)j.width * j.height;
  float* dst = j.dst + (long)blockIdx.z * j.dst_rep;
  const float* src = j.src + (long)blockIdx.z * j.src_rep;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / j.width, c = i % j.width;
    dst[r * j.dpitch + c] = src[r * j.spitch + c];
  }
}

// ---- statistics of the gene output on the device (smx_predict_stat): what the reference's callers ask of predict()'s result --
// y.mean() / .variance() / .log_prob(x), posterior.py:187-255 -- computed from the parameter planes of a pass WITHOUT the planes leaving
// the device (24 KB per cell and draw at C2: predict's D2H traffic).  Formulas = sisua_amd/distributions.py (float32 here).
struct StatArgs {
  const float* P; long ldp; long plane_stride;   // planes of `rows` = Sn * B stacked rows (row s * B + b = draw s of cell b)
  int B, Sn, G, lk, direct, count_only, stat;    // stat: 0 mean, 1 variance (per draw), 2 mean averaged over the draws, 3 log_prob
  float inv_S; int accumulate;                   // stat 2: this pass adds (sum over its draws) * inv_S to what earlier passes left
  float* dst; long dst_draw;                     // stat 0 / 1: dst[s * dst_draw + b * G + g]; stat 2: dst[b * G + g]; stat 3: dst[s * dst_draw + b]
  const float* T; long ldt; const int32_t* trows; int t_u16;   // stat 3: targets [B][ldt] (or the resident rows trows[b])
};
__device__ inline void plane_moments(int lk, int direct, int count_only, float p0, float p1, float p2, float& mean, float& var) {
  float mc, vc;
  if (lk == SMX_LLK_MSE) { mean = p0; var = 0.f; return; }
  if (lk == SMX_LLK_NB || lk == SMX_LLK_ZINB) { const float el = expf(p1); mc = expf(p0) * el; vc = mc * (1.f + el); }
  else {
    const float mu = direct ? p0 : softplus_sigmoid(p0).sp, th = direct ? p1 : softplus_sigmoid(p1 + SMX_SOFTPLUS_INV_1).sp;
    mc = mu; vc = mu + mu * mu / th;
  }
  if ((lk == SMX_LLK_ZINB || lk == SMX_LLK_ZINBD) && !count_only) {
    const float pi = 1.f / (1.f + expf(-p2));
    mean = (1.f - pi) * mc; var = (1.f - pi) * (vc + pi * mc * mc);
  } else { mean = mc; var = vc; }
}
__global__ __launch_bounds__(256) void plane_stat_kernel(StatArgs a) {
  const int g = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (g >= a.G) return;
  const bool zi = a.lk == SMX_LLK_ZINB || a.lk == SMX_LLK_ZINBD;
  float acc = 0.f;
  for (int s = 0; s < a.Sn; ++s) {   // (draw order: the average over the draws is deterministic)
    const float* p = a.P + ((long)s * a.B + b) * a.ldp + g;
    float mean, var;
    plane_moments(a.lk, a.direct, a.count_only, p[0], a.lk == SMX_LLK_MSE ? 0.f : p[a.plane_stride], zi ? p[2 * a.plane_stride] : 0.f, mean, var);
    if (a.stat == 2) acc += mean;
    else a.dst[(long)s * a.dst_draw + (long)b * a.G + g] = a.stat == 0 ? mean : var;
  }
  if (a.stat == 2) { float* d = a.dst + (long)b * a.G + g; *d = (a.accumulate ? *d : 0.f) + acc * a.inv_S; }
}
template <int LK, int DIRECT>
__device__ inline float row_llk_elem(float x, float p0, float p1, float p2) {
  float llk, d0, d1, d2;
  count_elem<LK, DIRECT>(x, p0, p1, p2, llk, d0, d1, d2);
  return llk - lgammaf(x + 1.f);
}
// stat 3: one workgroup per stacked row: log p(target row | the row's planes), summed over the genes (Independent(..., 1).log_prob)
__global__ __launch_bounds__(256) void plane_logprob_kernel(StatArgs a) {
  __shared__ float sh[4];
  const int row = blockIdx.x, b = row % a.B, s = row / a.B;
  const float* p = a.P + (long)row * a.ldp;
  const long trow = a.trows ? a.trows[b] : b;
  int lk = a.lk;
  if (a.count_only && lk == SMX_LLK_ZINB) lk = SMX_LLK_NB;
  if (a.count_only && lk == SMX_LLK_ZINBD) lk = SMX_LLK_NBD;
  float acc = 0.f;
  for (int g = threadIdx.x; g < a.G; g += 256) {
    const float x = a.t_u16 ? (float)reinterpret_cast<const uint16_t*>(a.T)[trow * a.ldt + g] : a.T[trow * a.ldt + g];
    const float p0 = p[g], p1 = a.lk == SMX_LLK_MSE ? 0.f : p[a.plane_stride + g], p2 = (a.lk == SMX_LLK_ZINB || a.lk == SMX_LLK_ZINBD) ? p[2 * a.plane_stride + g] : 0.f;
    float v;
    switch (lk) {
      case SMX_LLK_NB: v = row_llk_elem<SMX_LLK_NB, 0>(x, p0, p1, p2); break;
      case SMX_LLK_ZINB: v = row_llk_elem<SMX_LLK_ZINB, 0>(x, p0, p1, p2); break;
      case SMX_LLK_NBD: v = a.direct ? row_llk_elem<SMX_LLK_NBD, 1>(x, p0, p1, p2) : row_llk_elem<SMX_LLK_NBD, 0>(x, p0, p1, p2); break;
      case SMX_LLK_ZINBD: v = a.direct ? row_llk_elem<SMX_LLK_ZINBD, 1>(x, p0, p1, p2) : row_llk_elem<SMX_LLK_ZINBD, 0>(x, p0, p1, p2); break;
      default: { const float df = x - p0; v = -(df * df) / (float)a.G; }   // 'mse': -log_prob == tf.losses.mse
    }
    acc += v;
  }
  acc = wave_sum(acc);   // the four waves' sums meet in LDS, added in wave order
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) a.dst[(long)s * a.dst_draw + b] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
static int launch_plane_stat(hipStream_t st, const StatArgs& a) {
  if (a.stat == 3) hipLaunchKernelGGL(plane_logprob_kernel, dim3((unsigned)(a.Sn * a.B)), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(plane_stat_kernel, dim3((unsigned)((a.G + 255) / 256), (unsigned)a.B), dim3(256), 0, st, a);
  SMX_HIP(hipGetLastError());
  return SMX_OK;
}
// what smx_predict_stat asks of the batch loop below in place of the parameter planes
struct StatReq { int stat = 0, count_only = 0; const float* target = nullptr; float* out = nullptr; };

// Decoder layers over `rows` stacked rows (evaluation mode: moving statistics, no dropout; smx_score.hip).  The last
// layer's output: last_form 0 row-major f32 in place, 1 k-major f32 in ht [Hp][rows], 2 its three-way bf16 split in ht.
int stacked_decoder(smx_model* m, const float* z, long rows, float* const* hb, int last_form, float* ht, const float** out, int* out_ld) {
  const float* in = z;
  int ld = m->Dp;
  for (size_t i = 0; i < m->dec.size(); ++i) {
    MlpLayer& L = m->dec[i];
    GemmArgs g;
    g.A = in; g.lda = ld; g.B = P_(m, L.tW); g.ldb = m->tensors[L.tW].ld;
    g.M = (int)rows; g.N = L.out_p; g.K = L.in_p; g.C = hb[i & 1]; g.ldc = L.out_p; g.split_k = 1;
    if (L.bn < 0) { g.bias = P_(m, L.tBias); g.act = 1; g.leak = L.leak; }
    SMX_CHECK(launch_gemm(m->st, g));
    const bool last = (i + 1 == m->dec.size());
    if (L.bn >= 0 || (last && last_form != 0)) {
      ScoreBnArgs b;
      b.h = hb[i & 1]; b.R = rows; b.H = L.out; b.Hp = L.out_p; b.eps = m->cfg.bn_eps; b.leak = L.leak;
      if (L.bn >= 0) {
        b.gamma = P_(m, L.tGamma); b.beta = P_(m, L.tBeta);
        b.moving_mean = m->bn_moving + m->bn_off[L.bn]; b.moving_var = b.moving_mean + L.out_p;
      }
      if (last && last_form == 1) { b.out_t = ht; b.ldt = rows; }
      else if (last && last_form == 2) b.out3 = reinterpret_cast<__bf16*>(ht);
      SMX_CHECK(launch_score_bn_act(m->st, b));
    }
    in = hb[i & 1]; ld = L.out_p;
  }
  *out = in; *out_ld = ld;
  return SMX_OK;
}

// Stacked form (smx_score.hip): the encoder runs once, then the S draws of the B cells go through the decoder and the
// output head as S * B rows at a time (scvi: its library latent drawn per row as well, the raw planes materialised and a
// row-local softmax + likelihood launch, since its rate is normalised over all genes of a row; SCALE: its mixture prior
// in the latent part of log w).
bool stacked_scoring_ok(const smx_model* m) {
  if (!m->flags.stacked_scoring || !m->stochastic || m->use_injected || m->dec.empty()) return false;
  if (m->mixpost) return false;   // (mixture-density posterior: every draw picks its component -- the draw-by-draw form)
  if (m->scale && (m->Dp > 64 || m->cfg.n_components > 32 || m->scale_tril)) return false;   // (full-covariance components: the draw-by-draw form, whose prior term is scale_prior_fwd's)
  if (m->scvi && !scvi_score_supported(m->Gp)) return false;
  if (!head_loss_supported(1, m->dec.back().out_p, m->Gp) || (m->dec.back().out_p % 4)) return false;
  for (const MlpLayer& L : m->dec)
    if ((L.in_p % 4) || (L.out_p % 32)) return false;
  return m->dec[0].in_p == m->Dp;
}

}  // namespace smx

extern "C" {

int smx_forward(smx_model* m, const int32_t* row_ids, const float* host_x, const float* host_library, int32_t batch,
                int32_t sample_index, int32_t training, float* z_mean, float* z_scale, float* z_sample, float* l_mean,
                float* l_scale, float* l_sample, float* x_params, float* const* y_params) {
  SMX_REQUIRE(m, "null model");
  Pass ps;
  SMX_CHECK(setup_pass(m, ps, row_ids, host_x, host_library, batch, training, sample_index));
  SMX_REQUIRE(!(training && !row_ids), "training-mode forward needs resident rows");
  SMX_CHECK(forward_pass(m, ps, false, false));
  return fetch_forward(m, batch, z_mean, z_scale, z_sample, l_mean, l_scale, l_sample, x_params, y_params, 0);
}

int smx_forward_samples(smx_model* m, const int32_t* row_ids, const float* host_x, const float* host_library, int32_t batch,
                        int32_t n_samples, float* z_mean, float* z_scale, float* z_samples, float* l_mean, float* l_scale,
                        float* l_samples, float* x_params, float* const* y_params) {
  SMX_REQUIRE(m && n_samples > 0, "bad arguments");
  // several draws of a host batch: smx_predict over this one batch (same cell ids, same draws, same output layouts) decodes
  // them as rows of one pass instead of one decoder pass per draw
  if (!row_ids && host_x && n_samples > 1 && batch > 0 && batch <= m->Bmax && stacked_scoring_ok(m) && !m->scvi)
    return smx_predict(m, host_x, host_library, batch, batch, n_samples, z_mean, z_scale, z_samples, l_mean, l_scale, l_samples, x_params, y_params);
  Pass ps;
  SMX_CHECK(setup_pass(m, ps, row_ids, host_x, host_library, batch, 0, 0));
  const size_t B = (size_t)batch;
  for (int s = 0; s < n_samples; ++s) {
    ps.sample = s;
    // the encoders run once (eval mode: no noise in them); later draws re-sample the latents and decode
    SMX_CHECK(forward_pass(m, ps, false, false, s == 0 ? 0 : 2));
    SMX_CHECK(fetch_forward(m, batch, s == 0 ? z_mean : nullptr, s == 0 ? z_scale : nullptr,
                            z_samples ? z_samples + (size_t)s * B * m->D : nullptr, s == 0 ? l_mean : nullptr,
                            s == 0 ? l_scale : nullptr, l_samples ? l_samples + (size_t)s * B : nullptr,
                            x_params ? x_params + (size_t)s * m->k * B * m->G : nullptr, y_params, (size_t)s));
  }
  return SMX_OK;
}

static int predict_core(smx_model* m, const float* host_x, const float* host_library, int64_t n_cells, int32_t batch, int32_t n_samples,
                        float* z_mean, float* z_scale, float* z_samples, float* l_mean, float* l_scale, float* l_samples,
                        float* x_params, float* const* y_params, const StatReq* sr) {
  SMX_REQUIRE(m && host_x && n_cells > 0 && n_samples > 0, "bad arguments");
  SMX_REQUIRE(batch > 0 && batch <= m->Bmax, "batch must be in 1..max_batch");
  const size_t N = (size_t)n_cells, G = (size_t)m->G, D = (size_t)m->D, k = (size_t)m->k, S = (size_t)n_samples;
  const int Dp = m->Dp, lat_ld = m->lat_planes * Dp;
  if (!m->stochastic) z_scale = nullptr;
  if (!m->scvi) l_mean = l_scale = l_samples = nullptr;
  // ---- staging layout for a chunk of C cells (segments in floats; per-cell widths) ----
  size_t wy[SMX_MAX_LABELS] = {0, 0, 0, 0};
  size_t per_cell = 0;
  if (z_mean) per_cell += D;
  if (z_scale) per_cell += D;
  if (l_mean) per_cell += 1;
  if (l_scale) per_cell += 1;
  if (z_samples) per_cell += S * D;
  if (l_samples) per_cell += S;
  if (x_params) per_cell += S * k * G;
  // the requested statistic of the gene output, per cell: S G (mean / variance per draw), G (mean over the draws), S (log_prob)
  const size_t w_stat = !sr ? 0 : sr->stat == 2 ? G : sr->stat == 3 ? S : S * G;
  per_cell += w_stat;
  for (int j = 0; j < m->n_heads; ++j)
    if (y_params && y_params[j]) { wy[j] = (size_t)m->lab_ky[j] * (size_t)m->cfg.label_dim[j]; per_cell += S * wy[j]; }
  SMX_REQUIRE(per_cell > 0, "no output requested");
  // the INPUT rows of a chunk travel as ONE contiguous copy too (raw [C][G] -> a device re-pitch to [C][Gp]; library prior; the rows'
  // likelihood constants from one launch): a host-to-device copy per minibatch from the caller's pageable array is staged
  // synchronously by the runtime -- ~60 us per batch, at batch 8 (Posterior's default) most of the call
  const size_t Gp = (size_t)m->Gp;
  per_cell += G + Gp + 3;
  if (sr && sr->stat == 3 && sr->target && (size_t)batch * m->Gp > m->pred_target_floats) {   // device copy of a batch of target rows
    if (m->pred_target) hipFree(m->pred_target);
    m->pred_target = nullptr; m->pred_target_floats = 0;
    SMX_CHECK(dmalloc(&m->pred_target, (size_t)m->Bmax * m->Gp));
    m->pred_target_floats = (size_t)m->Bmax * m->Gp;
  }
  // 128 MB of staging (knob predict_stage_floats: tests force several chunks on small problems)
  const size_t cap_floats = (size_t)std::max(1.0, tuning("predict_stage_floats", (double)((size_t)32 << 20)));
  size_t C = std::max<size_t>((size_t)batch, cap_floats / per_cell / (size_t)batch * (size_t)batch);   // whole batches per chunk
  C = std::min(C, (N + (size_t)batch - 1) / (size_t)batch * (size_t)batch);
  if (C * per_cell > m->pred_floats) {
    if (m->pred_stage) hipFree(m->pred_stage);
    m->pred_stage = nullptr; m->pred_floats = 0;
    SMX_CHECK(dmalloc(&m->pred_stage, C * per_cell));
    m->pred_floats = C * per_cell;
  }
  float* st = m->pred_stage;
  float *s_zm = nullptr, *s_zs = nullptr, *s_lm = nullptr, *s_ls = nullptr, *s_zd = nullptr, *s_ld = nullptr, *s_xp = nullptr, *s_y[SMX_MAX_LABELS] = {nullptr, nullptr, nullptr, nullptr};
  if (z_mean) { s_zm = st; st += C * D; }
  if (z_scale) { s_zs = st; st += C * D; }
  if (l_mean) { s_lm = st; st += C; }
  if (l_scale) { s_ls = st; st += C; }
  if (z_samples) { s_zd = st; st += S * C * D; }
  if (l_samples) { s_ld = st; st += S * C; }
  if (x_params) { s_xp = st; st += S * k * C * G; }
  float* s_st = nullptr;
  if (w_stat) { s_st = st; st += C * w_stat; }
  float* in_raw = st; st += C * G;
  float* in_x = st; st += C * Gp;
  float* in_lib = st; st += C * 2;
  float* in_lgx1 = st; st += C;
  for (int j = 0; j < m->n_heads; ++j)
    if (wy[j]) { s_y[j] = st; st += S * C * wy[j]; }
  // the statistic of the planes of Sn draws (stacked rows) of the batch at b0 into the chunk's staging
  auto stat_of = [&](const float* P, long ldp_, int B, int Sn, size_t s0, size_t Cn, size_t b0, const Pass& ps) -> int {
    StatArgs a;
    a.P = P; a.ldp = ldp_; a.plane_stride = m->Gp; a.B = B; a.Sn = Sn; a.G = m->G; a.lk = m->cfg.likelihood; a.direct = m->scvi ? 1 : 0;
    a.count_only = sr->count_only; a.stat = sr->stat; a.inv_S = 1.f / (float)S; a.accumulate = s0 > 0 ? 1 : 0;
    a.T = nullptr; a.ldt = 0; a.trows = nullptr; a.t_u16 = 0;
    if (sr->stat == 2) { a.dst = s_st + b0 * G; a.dst_draw = 0; }
    else if (sr->stat == 3) {
      a.dst = s_st + s0 * Cn + b0; a.dst_draw = (long)Cn;
      if (sr->target) { a.T = m->pred_target; a.ldt = m->Gp; }
      else { a.T = ps.Xsrc; a.ldt = m->Gp; a.trows = ps.xrows; a.t_u16 = ps.x_u16; }   // the input rows themselves
    } else { a.dst = s_st + (s0 * Cn + b0) * G; a.dst_draw = (long)(Cn * G); }
    return launch_plane_stat(m->st, a);
  };
  auto out = [&](float* dst, const float* src, size_t count) -> int {   // one contiguous device -> host copy
    SMX_HIP(hipMemcpyAsync(dst, src, count * sizeof(float), hipMemcpyDeviceToHost, m->st));
    return SMX_OK;
  };
  // the stacked form: with several draws, and for a statistic request at any draw count (smx_predict itself keeps the draw-by-draw
  // kernels at one draw: bit-identical with smx_forward batch by batch)
  const bool stack = (S > 1 || sr != nullptr) && stacked_scoring_ok(m) && !m->scvi;
  // SUPER-BATCHES: in evaluation mode nothing couples the rows of a minibatch (moving statistics, no dropout) but the noise ids -- a
  // cell's id is its index within ITS minibatch -- so the stacked form takes k minibatches per pass (k batch <= max_batch) and hands the
  // draw kernel the ids (row % batch): same numbers, 1 / k of the launches (at Posterior's batch 8 a pass is ~12 launches per 8 cells)
  size_t step = (size_t)batch;
  if (stack && (size_t)m->Bmax >= 2 * (size_t)batch) {
    step = (size_t)batch * ((size_t)m->Bmax / (size_t)batch);
    if (m->pred_ids_batch != batch) {
      if (!m->pred_ids) SMX_HIP(hipMalloc((void**)&m->pred_ids, (size_t)m->Bmax * sizeof(int32_t)));
      std::vector<int32_t> ids((size_t)m->Bmax);
      for (int i = 0; i < m->Bmax; ++i) ids[(size_t)i] = i % batch;
      SMX_HIP(hipMemcpy(m->pred_ids, ids.data(), ids.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      m->pred_ids_batch = batch;
    }
  }
  SMX_REQUIRE(!m->scvi || host_library, "scvi needs host_library with host_x");
  bool any_y = false;
  for (int j = 0; j < m->n_heads; ++j) any_y = any_y || s_y[j] != nullptr;
  const bool need_dec = s_xp || s_st || any_y;   // (latents only: the decoder and the heads are not run at all)
  for (size_t c0 = 0; c0 < N; c0 += C) {
    const size_t Cn = std::min(C, N - c0);   // cells of this chunk
    {
      SMX_HIP(hipMemcpyAsync(in_raw, host_x + c0 * G, Cn * G * sizeof(float), hipMemcpyHostToDevice, m->st));
      if (Gp != G) SMX_HIP(hipMemsetAsync(in_x, 0, Cn * Gp * sizeof(float), m->st));
      PackJobs J; J.n = 1;
      PackJob& q = J.j[0];
      q.dst = in_x; q.dpitch = (long)Gp; q.src = in_raw; q.spitch = (long)G; q.width = (int)G; q.height = (int)Cn; q.n_rep = 1; q.dst_rep = 0; q.src_rep = 0;
      hipLaunchKernelGGL(pack_kernel, dim3((unsigned)std::min<size_t>(1024, (Cn * G + 255) / 256), 1, 1), dim3(256), 0, m->st, J);
      SMX_HIP(hipGetLastError());
      SMX_CHECK(launch_row_stats(m->st, in_x, 0, (long)Gp, (long)Cn, m->G, in_lgx1, nullptr));
      if (host_library) SMX_HIP(hipMemcpyAsync(in_lib, host_library + c0 * 2, Cn * 2 * sizeof(float), hipMemcpyHostToDevice, m->st));
    }
    for (size_t b0 = 0; b0 < Cn; b0 += step) {
      const int B = (int)std::min<size_t>(step, Cn - b0);
      const size_t g0 = c0 + b0;
      Pass ps;   // (what setup_pass leaves for a host batch, on the chunk's resident copy)
      ps.B = B; ps.training = 0; ps.sample = 0; ps.global_batch = B;
      ps.rows = nullptr; ps.Xsrc = in_x + b0 * Gp; ps.lib = in_lib + b0 * 2; ps.lgx1 = in_lgx1 + b0; ps.cell_base = 0;
      if (sr && sr->stat == 3 && sr->target)
        SMX_HIP(hipMemcpy2DAsync(m->pred_target, (size_t)m->Gp * sizeof(float), sr->target + g0 * G, G * sizeof(float), G * sizeof(float), (size_t)B,
                                 hipMemcpyHostToDevice, m->st));
      if (stack) {
        // ---- several draws: the encoder once, then the draws of this batch as rows of one decoder pass (as the scoring
        // paths, smx_score.hip) -- at batch 8 x 10 draws (Posterior's defaults, posterior.py:114-115) the draw-by-draw form
        // is 50 launches per 8 cells ----
        ps.sample = 0;
        SMX_CHECK(forward_pass(m, ps, false, false, 3));   // encoder + latent moments only
        int Hmax = 0, lab_floats = 0;
        for (const MlpLayer& L : m->dec) Hmax = std::max(Hmax, L.out_p);
        for (int j = 0; j < m->n_heads; ++j) lab_floats += s_y[j] ? m->tensors[m->t_labW[j]].ld : 0;
        const size_t ldp = k * (size_t)m->Gp;
        const int Sc = (int)std::min<size_t>(S, std::max<size_t>(1, (size_t)4096 / (size_t)B));   // draws per pass
        const size_t R = (size_t)Sc * B;
        const size_t need = R * ((size_t)Dp + 1 + 2 * (size_t)Hmax + ldp + (size_t)lab_floats);
        if (need > m->score_floats) {
          if (m->score_buf) { SMX_HIP(hipStreamSynchronize(m->st)); hipFree(m->score_buf); }
          m->score_buf = nullptr; m->score_floats = 0;
          SMX_CHECK(dmalloc(&m->score_buf, need));
          m->score_floats = need;
        }
        float* zst = m->score_buf;
        float* lwst = zst + R * Dp;
        float* hb[2] = {lwst + R, lwst + R + R * Hmax};
        float* Pst = hb[1] + R * Hmax;
        float* yst = Pst + R * ldp;
        {
          PackJobs J; J.n = 0;
          auto add1 = [&](float* dst, size_t dpitch, const float* src, size_t spitch, size_t width) {
            if (!dst) return;
            PackJob& q = J.j[J.n++];
            q.dst = dst; q.dpitch = (long)dpitch; q.src = src; q.spitch = (long)spitch; q.width = (int)width; q.height = B; q.n_rep = 1; q.dst_rep = 0; q.src_rep = 0;
          };
          add1(s_zm ? s_zm + b0 * D : nullptr, D, m->mixpost ? m->zmean : m->latbuf, (size_t)(m->mixpost ? Dp : lat_ld), D);
          add1(s_zs ? s_zs + b0 * D : nullptr, D, m->sig, (size_t)Dp, D);
          if (J.n) { hipLaunchKernelGGL(pack_kernel, dim3(8, (unsigned)J.n, 1), dim3(256), 0, m->st, J); SMX_HIP(hipGetLastError()); }
        }
        for (size_t s0 = 0; s0 < S; s0 += (size_t)Sc) {
          const int Sn = (int)std::min<size_t>((size_t)Sc, S - s0);
          const long rows = (long)Sn * B;
          ScoreDrawArgs d;
          d.lat = m->latbuf; d.ld = 2 * Dp; d.B = B; d.D = m->D; d.Dp = Dp; d.S = Sn; d.s0 = (int)s0;
          d.nk = make_key(m, ST_EPS_Z, 0, false); d.rows = step > (size_t)batch ? m->pred_ids : ps.rows; d.cell_base = ps.cell_base; d.z = zst; d.lw = lwst;
          SMX_CHECK(launch_score_draws(m->st, d));
          const float* hl = nullptr; int hld = 0;
          if (need_dec) SMX_CHECK(stacked_decoder(m, zst, rows, hb, 0, nullptr, &hl, &hld));
          PackJobs J; J.n = 0;
          int pack_err = SMX_OK;
          auto flush = [&]() {
            if (!J.n || pack_err != SMX_OK) return;
            const unsigned gx = (unsigned)std::min<size_t>(64, ((size_t)B * std::max(G, D) + 255) / 256);
            hipLaunchKernelGGL(pack_kernel, dim3(gx, (unsigned)J.n, (unsigned)Sn), dim3(256), 0, m->st, J);
            if (hipGetLastError() != hipSuccess) { set_error("pack_kernel launch failed"); pack_err = SMX_ERR_HIP; }
            J.n = 0;
          };
          auto addr = [&](float* dst, size_t dpitch, size_t dst_rep, const float* src, size_t spitch, size_t src_rep, size_t width) {
            if (!dst) return;
            if (J.n == SMX_PACK_MAX) flush();
            PackJob& q = J.j[J.n++];
            q.dst = dst; q.dpitch = (long)dpitch; q.src = src; q.spitch = (long)spitch; q.width = (int)width; q.height = B;
            q.n_rep = Sn; q.dst_rep = (long)dst_rep; q.src_rep = (long)src_rep;
          };
          addr(s_zd ? s_zd + (s0 * Cn + b0) * D : nullptr, D, Cn * D, zst, (size_t)Dp, (size_t)B * Dp, D);
          if (s_xp || s_st) {
            GemmArgs g;
            g.A = hl; g.lda = hld; g.B = P_(m, m->t_outW[0]); g.ldb = m->tensors[m->t_outW[0]].ld;
            g.C = Pst; g.ldc = (int)ldp; g.M = (int)rows; g.N = (int)ldp; g.K = hld; g.bias = P_(m, m->t_outb[0]); g.split_k = 1;
            SMX_CHECK(launch_gemm(m->st, g));
            for (size_t c = 0; c < k && s_xp; ++c)
              addr(s_xp + ((s0 * k + c) * Cn + b0) * G, G, k * Cn * G, Pst + c * (size_t)m->Gp, ldp, (size_t)B * ldp, G);
            if (s_st) SMX_CHECK(stat_of(Pst, (long)ldp, B, Sn, s0, Cn, b0, ps));
          }
          float* ycur = yst;
          for (int j = 0; j < m->n_heads; ++j) {
            if (!s_y[j]) continue;
            const TensorInfo& tw = m->tensors[m->t_labW[j]];
            GemmArgs g;
            g.A = hl; g.lda = hld; g.B = P_(m, m->t_labW[j]); g.ldb = tw.ld;
            g.C = ycur; g.ldc = tw.ld; g.M = (int)rows; g.N = tw.ld; g.K = hld; g.bias = P_(m, m->t_labb[j]); g.split_k = 1;
            SMX_CHECK(launch_gemm(m->st, g));
            const size_t P = (size_t)m->cfg.label_dim[j], Pp = (size_t)m->lab_Pp[j], ld = (size_t)tw.ld;
            for (size_t c = 0; c < (size_t)m->lab_ky[j]; ++c)
              addr(s_y[j] + (s0 * Cn + b0) * wy[j] + c * P, wy[j], Cn * wy[j], ycur + c * Pp, ld, (size_t)B * ld, P);
            ycur += R * ld;
          }
          flush();
          SMX_CHECK(pack_err);
        }
        continue;
      }
      for (size_t s = 0; s < S; ++s) {
        ps.sample = (int)s;
        // the encoders run once per batch (eval mode: no noise in them); later draws re-sample the latents and decode
        SMX_CHECK(forward_pass(m, ps, false, false, s == 0 ? ((need_dec || S > 1) ? 0 : 3) : 2));
        PackJobs J; J.n = 0;
        int pack_err = SMX_OK;
        auto flush = [&]() {
          if (!J.n || pack_err != SMX_OK) return;
          const unsigned gx = (unsigned)std::min<size_t>(256, ((size_t)B * std::max(G, D) + 255) / 256);
          hipLaunchKernelGGL(pack_kernel, dim3(gx, (unsigned)J.n), dim3(256), 0, m->st, J);
          if (hipGetLastError() != hipSuccess) { set_error("pack_kernel launch failed"); pack_err = SMX_ERR_HIP; }
          J.n = 0;
        };
        auto add = [&](float* dst, size_t dpitch, const float* src, size_t spitch, size_t width) {
          if (!dst) return;
          if (J.n == SMX_PACK_MAX) flush();
          PackJob& q = J.j[J.n++];
          q.dst = dst; q.dpitch = (long)dpitch; q.src = src; q.spitch = (long)spitch; q.width = (int)width; q.height = B;
          q.n_rep = 1; q.dst_rep = 0; q.src_rep = 0;
        };
        if (s == 0) {
          add(s_zm ? s_zm + b0 * D : nullptr, D, m->mixpost ? m->zmean : m->latbuf, (size_t)(m->mixpost ? Dp : lat_ld), D);
          add(s_zs ? s_zs + b0 * D : nullptr, D, m->sig, (size_t)Dp, D);
          add(s_lm ? s_lm + b0 : nullptr, 1, m->latlbuf, 32, 1);
          add(s_ls ? s_ls + b0 : nullptr, 1, m->lsig, 1, 1);
        }
        add(s_zd ? s_zd + (s * Cn + b0) * D : nullptr, D, m->z, (size_t)Dp, D);
        add(s_ld ? s_ld + s * Cn + b0 : nullptr, 1, m->lsmp, 1, 1);
        if (s_xp)
          for (size_t c = 0; c < k; ++c) add(s_xp + ((s * k + c) * Cn + b0) * G, G, m->P + c * (size_t)m->Gp, k * (size_t)m->Gp, G);
        if (s_st) SMX_CHECK(stat_of(m->P, (long)(k * (size_t)m->Gp), B, 1, s, Cn, b0, ps));
        for (int j = 0; j < m->n_heads; ++j) {
          if (!s_y[j]) continue;
          const size_t P = (size_t)m->cfg.label_dim[j], Pp = (size_t)m->lab_Pp[j], ld = (size_t)m->tensors[m->t_labW[j]].ld;
          for (size_t c = 0; c < (size_t)m->lab_ky[j]; ++c) add(s_y[j] + (s * Cn + b0) * wy[j] + c * P, wy[j], m->laby_raw[j] + c * Pp, ld, P);
        }
        flush();
        SMX_CHECK(pack_err);
      }
    }
    // ---- the chunk leaves the device: every segment's rows are contiguous here and in the caller's arrays ----
    if (s_zm) SMX_CHECK(out(z_mean + c0 * D, s_zm, Cn * D));
    if (s_zs) SMX_CHECK(out(z_scale + c0 * D, s_zs, Cn * D));
    if (s_lm) SMX_CHECK(out(l_mean + c0, s_lm, Cn));
    if (s_ls) SMX_CHECK(out(l_scale + c0, s_ls, Cn));
    if (s_st && sr->stat == 2) SMX_CHECK(out(sr->out + c0 * G, s_st, Cn * G));
    for (size_t s = 0; s < S && s_st && sr->stat != 2; ++s) {
      if (sr->stat == 3) SMX_CHECK(out(sr->out + s * N + c0, s_st + s * Cn, Cn));
      else SMX_CHECK(out(sr->out + (s * N + c0) * G, s_st + s * Cn * G, Cn * G));
    }
    for (size_t s = 0; s < S; ++s) {
      if (s_zd) SMX_CHECK(out(z_samples + (s * N + c0) * D, s_zd + s * Cn * D, Cn * D));
      if (s_ld) SMX_CHECK(out(l_samples + s * N + c0, s_ld + s * Cn, Cn));
      if (s_xp)
        for (size_t c = 0; c < k; ++c) SMX_CHECK(out(x_params + ((s * k + c) * N + c0) * G, s_xp + (s * k + c) * Cn * G, Cn * G));
      for (int j = 0; j < m->n_heads; ++j)
        if (s_y[j]) SMX_CHECK(out(y_params[j] + (s * N + c0) * wy[j], s_y[j] + s * Cn * wy[j], Cn * wy[j]));
    }
    SMX_HIP(hipStreamSynchronize(m->st));
  }
  return SMX_OK;
}

int smx_predict(smx_model* m, const float* host_x, const float* host_library, int64_t n_cells, int32_t batch, int32_t n_samples,
                float* z_mean, float* z_scale, float* z_samples, float* l_mean, float* l_scale, float* l_samples,
                float* x_params, float* const* y_params) {
  return predict_core(m, host_x, host_library, n_cells, batch, n_samples, z_mean, z_scale, z_samples, l_mean, l_scale, l_samples, x_params, y_params, nullptr);
}

int smx_predict_stat(smx_model* m, const float* host_x, const float* host_library, int64_t n_cells, int32_t batch, int32_t n_samples,
                     int32_t stat, int32_t count_only, const float* target, float* out) {
  SMX_REQUIRE(m && out && stat >= 0 && stat <= 3, "bad arguments (stat: 0 mean, 1 variance, 2 mean over the draws, 3 log_prob)");
  SMX_REQUIRE(!(count_only && m->cfg.likelihood == SMX_LLK_MSE), "the deterministic 'mse' output has no count distribution");
  StatReq sr;
  sr.stat = stat; sr.count_only = count_only ? 1 : 0; sr.target = target; sr.out = out;
  return predict_core(m, host_x, host_library, n_cells, batch, n_samples, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &sr);
}

int smx_decode(smx_model* m, const float* z, const float* l, int32_t batch, float* x_params, float* const* y_params) {
  SMX_REQUIRE(m && z, "null argument");
  SMX_REQUIRE(batch > 0 && batch <= m->Bmax, "batch must be in 1..max_batch");
  SMX_REQUIRE(!m->scvi || l, "scvi decode needs the library latent");
  Pass ps;
  ps.B = batch; ps.training = 0; ps.sample = 0; ps.global_batch = batch; ps.rows = nullptr; ps.Xsrc = m->hostX;
  ps.lib = m->hostLib; ps.lgx1 = m->hostLgx1; ps.cell_base = 0;
  SMX_HIP(hipMemsetAsync(m->z, 0, (size_t)batch * m->Dp * sizeof(float), m->st));
  SMX_HIP(hipMemcpy2DAsync(m->z, (size_t)m->Dp * sizeof(float), z, (size_t)m->D * sizeof(float), (size_t)m->D * sizeof(float),
                           (size_t)batch, hipMemcpyHostToDevice, m->st));
  if (m->scvi) SMX_HIP(hipMemcpyAsync(m->lsmp, l, (size_t)batch * sizeof(float), hipMemcpyHostToDevice, m->st));
  SMX_CHECK(forward_pass(m, ps, false, false, 1));
  SMX_HIP(hipStreamSynchronize(m->st));
  const int B = batch;
  SMX_CHECK(fetch_planes(m, B, x_params));
  if (y_params) {
    std::vector<float> tmp;
    for (int j = 0; j < m->n_heads; ++j) {
      if (!y_params[j]) continue;
      const int P = m->cfg.label_dim[j], Pp = m->lab_Pp[j], ld = m->tensors[m->t_labW[j]].ld;
      tmp.resize((size_t)B * ld);
      SMX_HIP(hipMemcpy(tmp.data(), m->laby_raw[j], tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
      for (int b = 0; b < B; ++b)
        for (int c = 0; c < m->lab_ky[j]; ++c)
          memcpy(y_params[j] + ((size_t)b * m->lab_ky[j] + c) * P, &tmp[(size_t)b * ld + (size_t)c * Pp], sizeof(float) * P);
    }
  }
  return SMX_OK;
}

}  // extern "C"
