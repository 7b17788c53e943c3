// smx_scoring.hip -- importance-weighted marginal likelihood and posterior-predictive scores (host side of smx_score.hip).
#include "smx_model.h"

namespace smx {

// importance weights of one posterior draw, folded into a running log-sum-exp per cell:
//   log w = log p(x|z) + log N(z;0,I) - log N(z;mu,sigma) [+ the library latent's terms, scvi]
struct IwArgs {
  const float* llk_part; int n_chunks; const float* lgx1; const int32_t* rows;
  const float* z; const float* sig; const float* eps; int D, Dp, stochastic;
  const float* l; const float* lsig; const float* leps; const float* library;  // scvi (library indexed like lgx1)
  float* run_max; float* run_sum; float* llk_sum; int B, first;
  const float* klmc;   // scale: log q(z|x) - log p_mixture(z) of this draw (replaces the N(0, I) prior terms)
};
// one wave per cell: lanes over the loss kernel's partial sums and over the latent dims
__global__ __launch_bounds__(256) void iw_accum_kernel(IwArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  float llk = 0.f;
  for (int c = lane; c < a.n_chunks; c += 64) llk += a.llk_part[(long)b * a.n_chunks + c];
  llk = wave_sum(llk);
  const long src = a.rows ? a.rows[b] : b;
  llk -= a.lgx1[src];
  float lw = 0.f;
  if (a.klmc) lw = (lane == 0) ? -a.klmc[b] : 0.f;
  else if (a.stochastic)
    for (int d = lane; d < a.D; d += 64) {
      const float z = a.z[(long)b * a.Dp + d], e = a.eps[(long)b * a.Dp + d], s = a.sig[(long)b * a.Dp + d];
      lw += -0.5f * z * z + 0.5f * e * e + logf(s);
    }
  lw = wave_sum(lw) + llk;
  if (lane != 0) return;
  if (a.l) {
    const float mp = a.library[src * 2], vp = a.library[src * 2 + 1];
    const float l = a.l[b], e = a.leps[b], s = a.lsig[b];
    lw += -0.5f * (l - mp) * (l - mp) / vp - 0.5f * logf(vp) + 0.5f * e * e + logf(s);
  }
  if (a.first) { a.run_max[b] = lw; a.run_sum[b] = 1.f; if (a.llk_sum) a.llk_sum[b] = llk; }
  else {
    const float mx = a.run_max[b], nm = fmaxf(mx, lw);
    a.run_sum[b] = a.run_sum[b] * expf(mx - nm) + expf(lw - nm);
    a.run_max[b] = nm;
    if (a.llk_sum) a.llk_sum[b] += llk;
  }
}

// one score over the stacked draws: the likelihood of `X` under the decoded parameters, folded per cell into a running
// log-sum-exp (with the latent terms of the importance weight: marginal_log_prob; without: Posterior.cal_llk's scores)
struct ScoreJob {
  const float* X = nullptr; int x_u16 = 0; const int32_t* xrows = nullptr;   // counts to score, [.. or B][Gp]
  const float* lgx1 = nullptr; const int32_t* lgrows = nullptr;              // their sum lgamma(x + 1) per cell
  int likelihood = 0;                                                        // the model's, or its count part without the zero-inflation gate
  int with_lw = 0;
  float* run_max = nullptr; float* run_sum = nullptr; float* llk_sum = nullptr;   // [B] each (llk_sum may be null)
};

static int stacked_scores(smx_model* m, const Pass& ps, int n_samples, const ScoreJob* jobs, int n_jobs) {
  const int B = ps.B, n_gt = head_loss_chunks(m->Gp);
  int Hmax = 0;
  for (const MlpLayer& L : m->dec) Hmax = std::max(Hmax, L.out_p);
  // rows per stacked pass: whole draws, up to 16 384 rows (SMX_SCORE_ROWS: the tests force several chunks)
  // (scvi: 4 096 rows -- their raw planes are 100 MB at 2 000 genes)
  const long cap_rows = std::max(1L, (long)tuning("score_rows", m->scvi ? 4096.0 : 16384.0));
  const int Sc = (int)std::min<long>(std::min<long>(n_samples, SMX_SCORE_MAX_DRAWS), std::max<long>(1, cap_rows / B));
  const size_t R = (size_t)Sc * B;
  const size_t raw_ld = (size_t)m->k * m->Gp;
  const size_t need = R * ((size_t)m->Dp + 2 + 4 * (size_t)Hmax + (size_t)n_gt + (m->scvi ? raw_ld : 0));
  if (need > m->score_floats) {
    if (m->score_buf) hipFree(m->score_buf);
    m->score_buf = nullptr; m->score_floats = 0;
    SMX_CHECK(dmalloc(&m->score_buf, need));
    m->score_floats = need;
  }
  float* z = m->score_buf;
  float* lw = z + R * m->Dp;
  float* hb[2] = {lw + R, lw + R + R * Hmax};
  float* ht = hb[1] + R * Hmax;          // the last layer's output: bf16 three-way split [3][R][Hp], or k-major f32 [Hp][R]
  float* part = ht + 2 * R * Hmax;
  float* lsmp = part + R * n_gt;         // scvi: the library latent of every row ...
  float* raw = lsmp + R;                 // ... and the k raw planes [R][k * Gp]
  const bool wide_head = m->scvi || tuning_on("score_head_wide") || !score_head_supported(m->dec.back().out_p, m->Gp);   // the training kernel's direct-operand form (A/B)
  const int nslab = m->dec.back().out_p / 32;
  // W as bf16 slab images, one set per plane count in use (3: zero-inflated likelihoods; 2: the others and the
  // count part of a zero-inflated one) -- once per call, W does not change meanwhile
  const __bf16* wimg[4] = {nullptr, nullptr, nullptr, nullptr};
  if (!wide_head && !m->scvi) {
    bool use_np[4] = {false, false, false, false};
    for (int j = 0; j < n_jobs; ++j) use_np[(jobs[j].likelihood == SMX_LLK_ZINB || jobs[j].likelihood == SMX_LLK_ZINBD) ? 3 : 2] = true;
    const size_t per_plane = (size_t)n_gt * nslab * 3 * 1024;   // bf16 elements per plane of an image set
    const size_t wneed = (per_plane * ((use_np[2] ? 2 : 0) + (use_np[3] ? 3 : 0)) + 1) / 2;   // ... as floats
    if (wneed > m->score_wimg_floats) {
      if (m->score_wimg) hipFree(m->score_wimg);
      m->score_wimg = nullptr; m->score_wimg_floats = 0; m->wimg_epoch = 0;
      SMX_CHECK(dmalloc(&m->score_wimg, wneed));
      m->score_wimg_floats = wneed;
    }
    __bf16* at = reinterpret_cast<__bf16*>(m->score_wimg);
    // (the images stand while the parameters do: a scoring sweep over a dataset splits W once, not once per batch)
    const int key = n_gt * 64 + nslab * 4 + (use_np[2] ? 1 : 0) + (use_np[3] ? 2 : 0);
    // (... and while the knobs do: ADVICE r05 -- the key used to ignore the head's form knobs)
    const bool fresh = m->wimg_epoch == m->params_epoch && m->wimg_key == key && m->wimg_tuning == tuning_epoch() && !tuning_on("no_wimg_cache");
    for (int np = 2; np <= 3; ++np) {
      if (!use_np[np]) continue;
      ScoreSplitWArgs sw;
      sw.W = P_(m, m->t_outW[0]); sw.ldw = m->tensors[m->t_outW[0]].ld; sw.Gp = m->Gp; sw.n_gt = n_gt; sw.nslab = nslab; sw.NP = np;
      sw.img = at;
      if (!fresh) SMX_CHECK(launch_score_split_w(m->st, sw));
      wimg[np] = at;
      at += per_plane * np;
    }
    m->wimg_epoch = m->params_epoch; m->wimg_key = key; m->wimg_tuning = tuning_epoch();
  }
  // (knob no_score_dec1: the three launches)
  const bool dec1 = !wide_head && !m->scvi && !m->scale && m->dec.size() == 1 && m->dec[0].bn >= 0 && m->dec[0].in_p == m->Dp &&
                    score_decoder1_supported(m->Dp, m->dec[0].out_p) && !tuning_on("no_score_dec1");
  // the encoders and the latent heads
  SMX_CHECK(forward_pass(m, ps, false, false, (m->scale || m->mixpost) ? 3 : 4));   // (4: without the latent moments' launch -- the draws below read `latbuf`)
  for (int s0 = 0; s0 < n_samples; s0 += Sc) {
    const int S = std::min(Sc, n_samples - s0);
    const long rows = (long)S * B;
    ScoreDrawArgs d;
    d.lat = m->latbuf; d.ld = 2 * m->Dp; d.B = B; d.D = m->D; d.Dp = m->Dp; d.S = S; d.s0 = s0;
    d.nk = make_key(m, ST_EPS_Z, 0, false); d.rows = ps.rows; d.cell_base = ps.cell_base; d.z = z; d.lw = lw;
    if (m->scale) {
      d.pr_logits = P_(m, m->t_prLogits); d.pr_loc = P_(m, m->t_prLoc); d.pr_scale_raw = P_(m, m->t_prScale); d.C = m->cfg.n_components;
    }
    if (m->scvi) {
      d.latl = m->latlbuf; d.ld_l = 32; d.library = ps.lib; d.lib_rows = ps.rows; d.nk_l = make_key(m, ST_EPS_L, 0, false); d.l = lsmp;
    }
    const float* in = nullptr;
    int ld = 0;
    if (dec1) {   // draws, product, BatchNorm, activation and split of a one-layer decoder in one launch (z is not stored)
      const MlpLayer& L = m->dec[0];
      ScoreDec1Args f;
      f.d = d; f.W = P_(m, L.tW); f.ldw = m->tensors[L.tW].ld; f.H = L.out; f.Hp = L.out_p;
      f.gamma = P_(m, L.tGamma); f.beta = P_(m, L.tBeta); f.moving_mean = m->bn_moving + m->bn_off[L.bn]; f.moving_var = f.moving_mean + L.out_p;
      f.eps = m->cfg.bn_eps; f.leak = L.leak; f.out3 = reinterpret_cast<__bf16*>(ht);
      SMX_CHECK(launch_score_decoder1(m->st, f));
    } else {
      SMX_CHECK(launch_score_draws(m->st, d));
      SMX_CHECK(stacked_decoder(m, z, rows, hb, m->scvi ? 0 : (wide_head ? 1 : 2), ht, &in, &ld));
    }
    if (m->scvi) {
      for (int ch = 0; ch < m->k; ++ch) {
        if (!m->out_has_W[ch]) {   // (dispersion / inflation = 'share': the per-gene vector in every row)
          SMX_CHECK(launch_plane_fill(m->st, raw + (size_t)ch * m->Gp, (long)raw_ld, P_(m, m->t_outb[ch]), (int)rows, m->Gp, m->out_single[ch] ? 1 : 0));
          continue;
        }
        GemmArgs g;
        g.A = in; g.lda = ld; g.B = P_(m, m->t_outW[ch]); g.ldb = m->tensors[m->t_outW[ch]].ld;
        g.C = raw + (size_t)ch * m->Gp; g.ldc = (int)raw_ld; g.M = (int)rows; g.N = m->Gp; g.K = ld; g.bias = P_(m, m->t_outb[ch]); g.split_k = 1;
        SMX_CHECK(launch_gemm(m->st, g));
      }
    }
    for (int j = 0; j < n_jobs; ++j) {
      const ScoreJob& q = jobs[j];
      if (m->scvi) {
        ScviScoreArgs sa;
        sa.raw = raw; sa.ld = (long)raw_ld; sa.plane_stride = m->Gp; sa.R = (int)rows; sa.G = m->G; sa.Gp = m->Gp; sa.k = m->k;
        sa.likelihood = q.likelihood; sa.row_mod = B; sa.l = lsmp; sa.clip_library = m->cfg.clip_library;
        sa.X = q.X; sa.ldx = m->Gp; sa.rows = q.xrows; sa.x_u16 = q.x_u16; sa.llk = part;
        SMX_CHECK(launch_scvi_score_rows(m->st, sa));
      } else if (!wide_head) {
        ScoreHeadArgs sh;
        sh.A3 = reinterpret_cast<const __bf16*>(ht);
        sh.Wimg = wimg[(q.likelihood == SMX_LLK_ZINB || q.likelihood == SMX_LLK_ZINBD) ? 3 : 2]; sh.bias = P_(m, m->t_outb[0]);
        sh.X = q.X; sh.x_u16 = q.x_u16; sh.ldx = m->Gp; sh.rows = q.xrows; sh.llk_part = part;
        sh.R = (int)rows; sh.row_mod = B; sh.G = m->G; sh.Gp = m->Gp; sh.Hp = m->dec.back().out_p; sh.likelihood = q.likelihood;
        SMX_CHECK(launch_score_head(m->st, sh));
      } else {
        HeadLossArgs hl;
        hl.H = ht; hl.ldh = (int)rows; hl.W = P_(m, m->t_outW[0]); hl.ldw = m->tensors[m->t_outW[0]].ld; hl.bias = P_(m, m->t_outb[0]);
        hl.X = q.X; hl.x_u16 = q.x_u16; hl.ldx = m->Gp; hl.rows = q.xrows; hl.llk_part = part;
        hl.B = (int)rows; hl.G = m->G; hl.Gp = m->Gp; hl.Hp = m->dec.back().out_p; hl.likelihood = q.likelihood; hl.grad_scale = 0.f;
        hl.llk_only = 1; hl.row_mod = B;
        SMX_CHECK(launch_out_head_loss(m->st, hl));
      }
      IwStackArgs w;
      w.llk_part = part; w.n_chunks = m->scvi ? 1 : n_gt; w.lw = q.with_lw ? lw : nullptr; w.lgx1 = q.lgx1; w.rows = q.lgrows;
      w.run_max = q.run_max; w.run_sum = q.run_sum; w.llk_sum = q.llk_sum; w.B = B; w.S = S; w.first = (s0 == 0);
      SMX_CHECK(launch_iw_stack(m->st, w));
    }
  }
  return SMX_OK;
}

static int marginal_llk_stacked(smx_model* m, const Pass& ps, int n_samples, float* run) {
  ScoreJob q;
  q.X = ps.Xsrc; q.x_u16 = ps.x_u16; q.xrows = ps.xrows; q.lgx1 = ps.lgx1; q.lgrows = ps.rows; q.likelihood = m->cfg.likelihood;
  q.with_lw = 1; q.run_max = run; q.run_sum = run + ps.B; q.llk_sum = run + 2 * ps.B;
  return stacked_scores(m, ps, n_samples, &q, 1);
}

// scratch of the scoring entry points, kept across calls (hipMalloc + hipFree per call cost more than the stacked pass)
static int score_aux(smx_model* m, size_t floats, float** out) {
  if (floats > m->score_aux_floats) {
    if (m->score_aux) { SMX_HIP(hipStreamSynchronize(m->st)); hipFree(m->score_aux); }
    m->score_aux = nullptr; m->score_aux_floats = 0;
    SMX_CHECK(dmalloc(&m->score_aux, floats));
    m->score_aux_floats = floats;
  }
  *out = m->score_aux;
  return SMX_OK;
}

// the results' landing area on the host, pinned and kept across calls: into a pageable array the runtime's copy waits for the stream, stages, and the
// hipStreamSynchronize behind it is a second trip through the runtime for nothing
static int score_landing(smx_model* m, size_t floats, float** out) {
  if (floats > m->score_pin_floats) {
    if (m->score_pin) { SMX_HIP(hipStreamSynchronize(m->st)); hipHostFree(m->score_pin); }
    m->score_pin = nullptr; m->score_pin_floats = 0;
    SMX_HIP(hipHostMalloc((void**)&m->score_pin, floats * sizeof(float), hipHostMallocDefault));
    m->score_pin_floats = floats;
  }
  *out = m->score_pin;
  return SMX_OK;
}

}  // namespace smx

extern "C" {

int smx_marginal_llk(smx_model* m, const int32_t* row_ids, const float* host_x, const float* host_library, int32_t batch,
                     int32_t n_samples, float* mllk, float* llk_mean) {
  SMX_REQUIRE(m && mllk && n_samples > 0, "bad arguments");
  SMX_REQUIRE(m->cfg.likelihood != SMX_LLK_MSE, "the 'mse' output is not a normalised density: no marginal likelihood");
  Pass ps;
  SMX_CHECK(setup_pass(m, ps, row_ids, host_x, host_library, batch, 0, 0));
  const bool stacked = stacked_scoring_ok(m);
  // a deterministic latent (DCA) decodes to the same parameters in every draw: one pass is the whole estimate
  if (!m->stochastic) n_samples = 1;
  float* run = nullptr;   // [3][B]: running max, running sum, sum of log p(x|z)
  SMX_CHECK(score_aux(m, (size_t)3 * batch, &run));
  int rc = SMX_OK;
  if (stacked) rc = marginal_llk_stacked(m, ps, n_samples, run);
  for (int s = 0; !stacked && s < n_samples && rc == SMX_OK; ++s) {
    ps.sample = s;
    rc = forward_pass(m, ps, false, false, s == 0 ? 0 : 2);
    if (rc != SMX_OK) break;
    LossArgs lo;
    lo.likelihood = m->cfg.likelihood; lo.direct = m->scvi; lo.backward = 0;
    lo.X = ps.Xsrc; lo.x_u16 = ps.x_u16; lo.ldx = m->Gp; lo.rows = ps.xrows;
    lo.P = m->P; lo.ldp = (long)m->k * m->Gp; lo.plane_stride = m->Gp; lo.dP = m->dP; lo.llk_part = m->llk_part;
    lo.B = ps.B; lo.G = m->G; lo.Gp = m->Gp; lo.grad_scale = 0.f;
    rc = launch_count_loss(m->st, lo);
    if (rc != SMX_OK) break;
    IwArgs a;
    a.llk_part = m->llk_part; a.n_chunks = loss_chunks(m->Gp, ps.B); a.lgx1 = ps.lgx1; a.rows = ps.rows;
    a.z = m->z; a.sig = m->sig; a.eps = m->eps; a.D = m->D; a.Dp = m->Dp; a.stochastic = m->stochastic;
    a.l = m->scvi ? m->lsmp : nullptr; a.lsig = m->lsig; a.leps = m->leps; a.library = ps.lib;
    a.run_max = run; a.run_sum = run + batch; a.llk_sum = run + 2 * batch; a.B = batch; a.first = (s == 0);
    a.klmc = (m->scale || m->mixpost) ? m->kl : nullptr;   // (Monte-Carlo KL models: log p(z) - log q(z|x) of the draw is minus that term)
    hipLaunchKernelGGL(iw_accum_kernel, dim3((batch + 3) / 4), dim3(256), 0, m->st, a);
  }
  if (rc == SMX_OK) {
    float* h = nullptr;
    SMX_CHECK(score_landing(m, (size_t)3 * batch, &h));
    hipError_t e = hipMemcpyAsync(h, run, (size_t)3 * batch * sizeof(float), hipMemcpyDeviceToHost, m->st);
    if (e == hipSuccess) e = hipStreamSynchronize(m->st);
    if (e != hipSuccess) { set_error(std::string("marginal_llk readback failed: ") + hipGetErrorString(e)); rc = SMX_ERR_HIP; }
    else
      for (int b = 0; b < batch; ++b) {
        mllk[b] = h[b] + logf(h[batch + b]) - logf((float)n_samples);
        if (llk_mean) llk_mean[b] = h[2 * batch + b] / (float)n_samples;
      }
  } else {
    hipStreamSynchronize(m->st);
  }
  return rc;
}

int smx_score_llk(smx_model* m, const int32_t* row_ids, const float* host_x, const float* host_library,
                  const float* const* targets, int32_t n_targets, int32_t batch, int32_t n_samples, float* out) {
  SMX_REQUIRE(m && out && n_samples > 0 && n_targets >= 1 && n_targets <= 4, "bad arguments");
  SMX_REQUIRE(m->cfg.likelihood != SMX_LLK_MSE, "the 'mse' output is not a normalised density: no posterior-predictive scores");
  if (!m->stochastic) n_samples = 1;   // (deterministic latent: every draw decodes to the same parameters)
  Pass ps;
  SMX_CHECK(setup_pass(m, ps, row_ids, host_x, host_library, batch, 0, 0));
  const int lk = m->cfg.likelihood;
  const bool zi = (lk == SMX_LLK_ZINB || lk == SMX_LLK_ZINBD);
  const int n_dist = zi ? 2 : 1;
  const size_t plane = (size_t)batch * m->Gp;
  float *tX = nullptr, *tLg = nullptr, *run = nullptr;   // run: [n_targets][2]{max, sum}[batch]
  int rc = SMX_OK;
  {
    float* aux = nullptr;
    SMX_CHECK(score_aux(m, plane * n_targets + (size_t)batch * n_targets + (size_t)n_targets * 2 * 2 * batch, &aux));
    tX = aux; tLg = tX + plane * n_targets; run = tLg + (size_t)batch * n_targets;
  }
  hipError_t e = hipMemsetAsync(tX, 0, plane * n_targets * sizeof(float), m->st);
  for (int t = 0; t < n_targets && e == hipSuccess && rc == SMX_OK; ++t) {
    const float* src = targets ? targets[t] : nullptr;
    if (!src) continue;   // NULL target: score against the input cells themselves
    e = hipMemcpy2DAsync(tX + plane * t, (size_t)m->Gp * sizeof(float), src, (size_t)m->G * sizeof(float),
                         (size_t)m->G * sizeof(float), (size_t)batch, hipMemcpyHostToDevice, m->st);
    // sum lgamma(x + 1) per cell of the target, on the device (the kernel the resident matrix's constants come from;
    // on the host it was ~0.5 ms of lgamma() calls per call)
    if (e == hipSuccess) rc = launch_row_stats(m->st, tX + plane * t, 0, m->Gp, batch, m->G, tLg + (size_t)batch * t, nullptr);
  }
  if (e != hipSuccess) { set_error(std::string("score_llk upload failed: ") + hipGetErrorString(e)); rc = SMX_ERR_HIP; }
  const bool stacked = stacked_scoring_ok(m);
  if (stacked && rc == SMX_OK) {
    // all draws as rows of one decoder pass; one likelihood-only head launch per (target, distribution)
    ScoreJob jobs[8];
    int nj = 0;
    for (int t = 0; t < n_targets; ++t) {
      const bool own = !(targets && targets[t]);
      for (int j = 0; j < n_dist; ++j) {
        ScoreJob& q = jobs[nj++];
        q.likelihood = (j == 0) ? lk : (lk == SMX_LLK_ZINB ? SMX_LLK_NB : SMX_LLK_NBD);
        q.X = own ? ps.Xsrc : tX + plane * t; q.x_u16 = own ? ps.x_u16 : 0; q.xrows = own ? ps.xrows : nullptr;
        q.lgx1 = own ? ps.lgx1 : tLg + (size_t)batch * t; q.lgrows = own ? ps.rows : nullptr;
        float* r = run + ((size_t)t * 2 + j) * 2 * batch;
        q.with_lw = 0; q.run_max = r; q.run_sum = r + batch; q.llk_sum = nullptr;
      }
    }
    rc = stacked_scores(m, ps, n_samples, jobs, nj);
  }
  for (int s = 0; !stacked && s < n_samples && rc == SMX_OK; ++s) {
    ps.sample = s;
    rc = forward_pass(m, ps, false, false, s == 0 ? 0 : 2);
    for (int t = 0; t < n_targets && rc == SMX_OK; ++t) {
      const bool own = !(targets && targets[t]);
      for (int j = 0; j < n_dist && rc == SMX_OK; ++j) {
        LossArgs lo;
        // j == 1: the count distribution under the zero-inflation wrapper (first two planes, no gate)
        lo.likelihood = (j == 0) ? lk : (lk == SMX_LLK_ZINB ? SMX_LLK_NB : SMX_LLK_NBD);
        lo.direct = m->scvi; lo.backward = 0;
        lo.X = own ? ps.Xsrc : tX + plane * t; lo.x_u16 = own ? ps.x_u16 : 0; lo.ldx = m->Gp; lo.rows = own ? ps.xrows : nullptr;
        lo.P = m->P; lo.ldp = (long)m->k * m->Gp; lo.plane_stride = m->Gp; lo.dP = m->dP; lo.llk_part = m->llk_part;
        lo.B = ps.B; lo.G = m->G; lo.Gp = m->Gp; lo.grad_scale = 0.f;
        rc = launch_count_loss(m->st, lo);
        if (rc != SMX_OK) break;
        IwArgs a;
        memset(&a, 0, sizeof(a));
        a.llk_part = m->llk_part; a.n_chunks = loss_chunks(m->Gp, ps.B);
        a.lgx1 = own ? ps.lgx1 : tLg + (size_t)batch * t; a.rows = own ? ps.rows : nullptr;
        a.D = m->D; a.Dp = m->Dp; a.stochastic = 0; a.l = nullptr;
        float* r = run + ((size_t)t * 2 + j) * 2 * batch;
        a.run_max = r; a.run_sum = r + batch; a.llk_sum = nullptr; a.B = batch; a.first = (s == 0);
        hipLaunchKernelGGL(iw_accum_kernel, dim3((batch + 3) / 4), dim3(256), 0, m->st, a);
      }
    }
  }
  if (rc == SMX_OK) {
    float* h = nullptr;
    SMX_CHECK(score_landing(m, (size_t)n_targets * 2 * 2 * batch, &h));
    e = hipMemcpyAsync(h, run, (size_t)n_targets * 2 * 2 * batch * sizeof(float), hipMemcpyDeviceToHost, m->st);
    if (e == hipSuccess) e = hipStreamSynchronize(m->st);
    if (e != hipSuccess) { set_error(std::string("score_llk readback failed: ") + hipGetErrorString(e)); rc = SMX_ERR_HIP; }
    else
      for (int t = 0; t < n_targets; ++t)
        for (int j = 0; j < 2; ++j) {
          const float* r = h + ((size_t)t * 2 + (j < n_dist ? j : 0)) * 2 * batch;
          for (int b = 0; b < batch; ++b)
            out[((size_t)t * 2 + j) * batch + b] = r[b] + logf(r[batch + b]) - logf((float)n_samples);
        }
  } else {
    hipStreamSynchronize(m->st);
  }
  return rc;
}

}  // extern "C"
