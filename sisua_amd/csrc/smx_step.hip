// smx_step.hip -- one training / evaluation step as a launch sequence: forward, ELBO, backward, (all-reduce), optimiser;
// eager and captured-graph execution.
#include "smx_model.h"

namespace smx {

__global__ void bn_moving_update_kernel(float* moving, const float* batch_sum, int n, float inv_world, float momentum) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) moving[i] = moving[i] * momentum + batch_sum[i] * inv_world * (1.f - momentum);
}

void fill_adam_args(smx_model* m, AdamArgs& a);
// SyncBatchNorm applies to training passes of a data-parallel job only (eval mode uses the moving statistics)
bool sync_bn_on(const smx_model* m, int training) { return m->sync_bn && training && m->cfg.batchnorm && dp_active(m); }
BnSyncArgs sync_args(smx_model* m) { BnSyncArgs y; y.gather = m->sync_buf; y.rank = m->rank; y.world = m->world; return y; }

// shapes / modes under which the decoder's first BatchNorm launch takes the latent sample and its product along
// (forward_pass adds what depends on injected noise)
static bool front_shapes_ok(smx_model* m, const Pass& ps) {
  const int lat_ld = m->lat_planes * m->Dp;
  return m->flags.front && !m->scale && !m->mixpost && !sync_bn_on(m, ps.training) && bn_front_supported(ps.B, m->Dp) &&
         (m->Dp == 32 || m->Dp == 64) && m->dec[0].in_p == m->Dp && m->dec[0].out_p % 8 == 0 && (lat_ld % 4) == 0;
}


// the model's products run from bf16 MFMAs on three-way split operands (flag "bf16x3": -1 = by size, SMX_BF16X3_MIN_WORK)
static bool b3_on(const smx_model* m, const Pass& ps) {
  return m->flags.bf16x3 < 0 ? use_bf16x3((long)ps.B * m->Gp * m->k) : m->flags.bf16x3 != 0;
}

// twin: another MLP whose FIRST layer consumes the same input (scvi: the library encoder beside the encoder).  When the
// shapes allow, both first layers run as ONE product launch and ONE BatchNorm launch (side by side along the output
// columns); *twin_done tells the caller, who then continues the twin from its second layer (first_layer = 1).
int mlp_forward(smx_model* m, std::vector<MlpLayer>& mlp, const Pass& ps, const float* in0, int ld0, bool in_is_x,
                const char* label0, int n_layers = -1, const LatentArgs* front = nullptr, int first_layer = 0,
                std::vector<MlpLayer>* twin = nullptr, bool* twin_done = nullptr) {
  const float* in = in0;
  int ld = ld0;
  const size_t nl = n_layers < 0 ? mlp.size() : (size_t)n_layers;
  auto make_gemm = [&](MlpLayer& L, const float* a_in, int a_ld, bool first_x, float* slab) {
    const TensorInfo& tw = m->tensors[L.tW];
    GemmArgs g;
    g.A = a_in; g.lda = a_ld; g.B = P_(m, L.tW); g.ldb = tw.ld;
    g.M = ps.B; g.N = L.out_p; g.K = L.in_p;
    g.C = slab; g.ldc = L.out_p; g.slab_stride = (long)ps.B * L.out_p;
    g.split_k = suggest_split_k(ps.B, L.out_p, L.in_p);
    if (first_x) {
      g.use_xform = 1;
      g.xf.rows = ps.xrows; g.xf.u16 = ps.x_u16; g.xf.log1p = m->cfg.log_norm; g.xf.cell_base = ps.cell_base;
      if (ps.training && m->cfg.input_dropout > 0.f) {
        g.xf.drop_p = m->cfg.input_dropout; g.xf.drop_scale = 1.f / (1.f - m->cfg.input_dropout);
        g.xf.nk = make_key(m, ST_INPUT_DROPOUT, ps.sample, true);
        if (const Injected* ij = inj(m, ST_INPUT_DROPOUT)) { g.xf.inj_mask = ij->d; g.xf.inj_ld = ij->ld; }
      }
    }
    return g;
  };
  auto make_bn = [&](MlpLayer& L, const float* slab, int eff, long slab_stride) {
    BnFwdArgs b;
    b.pre = slab; b.n_slabs = eff; b.slab_stride = slab_stride; b.ld = L.out_p;
    b.B = ps.B; b.H = L.out; b.Hp = L.out_p; b.batchnorm = L.bn >= 0; b.training = ps.training; b.leak = L.leak;
    if (L.bn >= 0) {
      b.gamma = P_(m, L.tGamma); b.beta = P_(m, L.tBeta);
      b.moving_mean = m->bn_moving + m->bn_off[L.bn]; b.moving_var = b.moving_mean + L.out_p;
      b.batch_mean = m->grads + m->tail_off_bn + m->bn_off[L.bn]; b.batch_var = b.batch_mean + L.out_p;
      b.update_moving = (m->world == 1);
      b.momentum = m->cfg.bn_momentum; b.eps = m->cfg.bn_eps;
    } else {
      b.bias = P_(m, L.tBias);
    }
    b.xhat = L.xhat; b.inv_std = L.inv_std; b.out = L.out_buf;
    b.drop_p = ps.training ? L.drop_p : 0.f;
    b.nk = make_key(m, L.stream, ps.sample, true);
    b.rows = ps.rows; b.cell_base = ps.cell_base;
    if (const Injected* ij = inj(m, L.stream)) { b.inj_mask = ij->d; b.inj_ld = ij->ld; }
    return b;
  };
  if (twin_done) *twin_done = false;
  for (size_t i = (size_t)first_layer; i < nl; ++i) {
    if (i > 0 && i == (size_t)first_layer) { in = mlp[i - 1].out_buf; ld = mlp[i - 1].out_p; }
    MlpLayer& L = mlp[i];
    const TensorInfo& tw = m->tensors[L.tW];
    GemmArgs g = make_gemm(L, in, ld, i == 0 && in_is_x, m->slab);
    static const bool no_ahead = tuning_on("no_noise_ahead");
    const bool no_twin = !m->flags.twin;
    const bool sync = sync_bn_on(m, ps.training) && L.bn >= 0;
    // hidden -> hidden layers 32 / 64 / 128 wide: the BatchNorm launch stages the layer's INPUT tile [B][K] in LDS and forms its
    // own columns as dot products -- the same front the first decoder layer uses for the latent sample, here as a plain
    // copy (no product launch; the reference's default networks are [64, 64], configs/base.yaml:10-17)
    LatentArgs dense_la;
    const bool dense_front = m->flags.front && !sync && !(front != nullptr && i == 0) && !(i == 0 && in_is_x) && L.leak == 0.f &&
                             (L.in_p == 32 || L.in_p == 64 || (L.in_p == 128 && ps.B <= 128)) && bn_front_supported(ps.B, L.in_p) && L.out_p % 8 == 0 &&
                             (ld % 4) == 0;
    if (dense_front) {
      dense_la.stochastic = 0; dense_la.relu = 0; dense_la.training = ps.training;
      dense_la.lat = in; dense_la.ld = ld; dense_la.B = ps.B; dense_la.D = L.in; dense_la.Dp = L.in_p;
    }
    const LatentArgs* front_i = (front != nullptr && i == 0) ? front : (dense_front ? &dense_la : nullptr);
    const bool with_front = front_i != nullptr;   // the BatchNorm launch produces its own input (latent sample / input tile + product)
    int eff = 1;
    SMX_REQUIRE((size_t)std::max(g.split_k, 1) * (size_t)g.slab_stride <= m->slab_cap, "split-K slabs exceed the slab buffer");
    // ---- the twin's first layer beside this one: one product launch, one BatchNorm launch ----
    bool dual = false;
    GemmArgs g2;
    if (twin && i == 0 && !with_front && !sync && !no_twin && in_is_x && !twin->empty() && bn_dual_supported(ps.B) &&
        !(ps.training && m->cfg.input_dropout > 0.f) && (*twin)[0].in_p == L.in_p && L.leak == 0.f && (*twin)[0].leak == 0.f) {
      MlpLayer& T = (*twin)[0];
      float* slab2 = m->slab + (size_t)std::max(g.split_k, 1) * (size_t)g.slab_stride;
      g2 = make_gemm(T, in, ld, true, slab2);
      dual = ((size_t)std::max(g.split_k, 1) * ((size_t)g.slab_stride + (size_t)g2.slab_stride) <= m->slab_cap);
    }
    // layers without BatchNorm and without dropout (the FactorVAE discriminator; plain autoencoders at evaluation): bias
    // and activation in the product's own store path -- no bias / activation launch
    const bool epi_act = m->flags.act_epilogue && !dual && !with_front && !sync && L.bn < 0 && !(ps.training && L.drop_p > 0.f) &&
                         g.split_k <= 1 && !m->use_injected;
    if (epi_act) {
      g.bias = P_(m, L.tBias); g.act = 1; g.leak = L.leak; g.C = L.out_buf; g.ldc = L.out_p; g.split_k = 1;
      Timed t(m, (i == 0 && in_is_x) ? label0 : "gemm_mlp_fwd");
      // a deep contraction (the discriminator's 1000-wide layers): the direct-operand bf16 x 3 form (smx_dgemm.hip)
      if (b3_on(m, ps) && dgemm_supported(g) && !tuning_on("no_dgemm")) SMX_CHECK(launch_dgemm(m->st, g));
      else SMX_CHECK(launch_gemm(m->st, g));
      in = L.out_buf; ld = L.out_p;
      continue;
    }
    // wide panel (the bf16 x 3 regime): the first layer's product as one workgroup per K slice + a reduce launch (smx_bigk.hip)
    bool bigk = false;
    BigKArgs bk;
    if (i == 0 && in_is_x && !dual && !with_front && m->bigk_part && !g.xf.drop_p && !g.xf.inj_mask && L.out_p <= 128 &&
        (m->flags.bf16x3 < 0 ? use_bf16x3((long)ps.B * m->Gp * m->k) : m->flags.bf16x3 != 0) && !tuning_on("no_bigk")) {
      bk.A = in; bk.lda = ld; bk.a_u16 = ps.x_u16; bk.log1p = m->cfg.log_norm; bk.rows = ps.xrows;
      bk.Bm = P_(m, L.tW); bk.ldb = tw.ld; bk.b_kmajor = 1;
      bk.M = ps.B; bk.N = L.out_p; bk.K = L.in_p; bk.ldc = L.out_p; bk.slab_stride = (long)ps.B * L.out_p;
      bk.part = m->bigk_part; bk.out = m->slab;
      bk.n_slices = bigk_slices(bk.K, SMX_BIGK_MAX_SLICES, &bk.k_chunk);
      bigk = bk.log1p && bigk_supported(bk) && (size_t)bk.n_slices * (size_t)bk.slab_stride <= m->bigk_floats;
      // ... and no reduce launch: column-major slabs, summed by the BatchNorm launch (bn_wide_fwd_kernel)
      if (bigk && !sync && bn_wide_supported(ps.B, L.out_p, bk.n_slices) && (size_t)bk.n_slices * 128 * 128 <= m->bigk_floats) {
        bk.colmajor = 1; bk.slab_stride = 128L * 128;
      }
    }
    if (bigk) {
      Timed t(m, label0);
      SMX_CHECK(launch_bigk(m->st, bk));
      eff = 1;
    } else if (dual) {
      Timed t(m, label0);
      SMX_CHECK(launch_gemm_dual(m->st, g, g2, &eff));
    } else if (!with_front) {
      // the gathered first layer at up to 128 cells: its (at most 16) split-K slabs column-major, summed by a BatchNorm launch of one
      // workgroup per column -- every load 16 bytes of a contiguous 512-byte column instead of 4 bytes of a 32-byte row piece, the additions
      // in the same order (slab 0, 1, ...): the same bits (bn_wide_fwd_kernel)
      if (i == 0 && in_is_x && !sync && g.use_xform && g.tile == TILE_AUTO && g.split_k <= 16 && bn_wide_supported(ps.B, L.out_p, g.split_k) &&
          (size_t)g.split_k * 128 * (size_t)L.out_p <= m->slab_cap && !tuning_on("xf_tile")) {
        g.c_colmajor = 1; g.slab_stride = 128L * L.out_p;
      }
      Timed t(m, (i == 0 && in_is_x) ? label0 : "gemm_mlp_fwd");
      SMX_CHECK(launch_gemm(m->st, g, &eff));
    }
    BnFwdArgs b = make_bn(L, m->slab, eff, g.slab_stride);
    if (bigk && bk.colmajor) { b.pre = bk.part; b.n_slabs = bk.n_slices; b.slab_stride = bk.slab_stride; b.wide = 1; }
    else if (!bigk && !dual && !with_front && g.c_colmajor) b.wide = 1;
    if (!no_ahead && !sync && i == 0 && in_is_x && &mlp == &m->enc && ps.training && front_shapes_ok(m, ps) &&
        !with_front && b.n_jobs == 0) {
      // the decoder's front launch (latent sample + first decoder layer) computes the whole latent tile in EVERY one of
      // its workgroups: its Philox draws (eps: ~1.2 us at batch 128, twice that at 256; dropout ~1 us) are made here
      // instead, once, by extra workgroups on CUs this launch leaves idle
      auto add = [&](float* dst, int ld, int width, int normal, float p, int stream) {
        NoiseJob& j = b.jobs[b.n_jobs++];
        j.dst = dst; j.ld = ld; j.width = width; j.normal = normal; j.p = p;
        j.stream = (uint32_t)((stream & 0xFF) | ((ps.sample & 0xFFFFFF) << 8));
      };
      const MlpLayer& d0 = m->dec[0];
      if (d0.drop_p > 0.f && !inj(m, d0.stream)) { add(d0.noise, d0.out_p, d0.out, 0, d0.drop_p, d0.stream); m->ahead_front_drop = true; }
      if (m->stochastic && !inj(m, ST_EPS_Z)) { add(m->noise_eps, m->Dp, m->D, 1, 0.f, ST_EPS_Z); m->ahead_front_eps = true; }
      if (b.n_jobs) b.nk.step_ptr = &cur_state(m)->step;
    }
    if (dual) {
      MlpLayer& T = (*twin)[0];
      const BnFwdArgs b2 = make_bn(T, g2.C, eff, g2.slab_stride);
      Timed t(m, "bn_fwd");
      SMX_CHECK(launch_bn_act_fwd_dual(m->st, b, b2));
      if (twin_done) *twin_done = true;
    } else if (with_front) {
      if (m->ahead_front_drop && front != nullptr && i == 0 && !b.inj_mask && b.drop_p > 0.f) { b.inj_mask = L.noise; b.inj_ld = L.out_p; }
      b.front = 1; b.lat = *front_i; b.W = P_(m, L.tW); b.ldw = tw.ld; b.n_jobs = 0;
      Timed t(m, "bn_fwd");
      SMX_CHECK(launch_bn_act_fwd(m->st, b));
    } else if (sync) {
      Timed t(m, "bn_fwd");
      b.n_jobs = 0;
      const BnSyncArgs y = sync_args(m);
      SMX_REQUIRE((size_t)y.world * 2 * L.out_p <= m->sync_cap, "SyncBatchNorm buffer too small");
      SMX_CHECK(launch_bn_sync_fwd(m->st, b, y, 0));
      SMX_CHECK(dp_allreduce_buf(m, m->sync_buf, (size_t)y.world * 2 * L.out_p, m->st));
      SMX_CHECK(launch_bn_sync_fwd(m->st, b, y, 1));
    } else {
      Timed t(m, "bn_fwd");
      SMX_CHECK(launch_bn_act_fwd(m->st, b));
    }
    in = L.out_buf; ld = L.out_p;
  }
  return SMX_OK;
}


// Single GPU: once the head products have written dW / db of the output and label heads (3/4 of the parameters),
// their clip + Adam update rides along with the next BatchNorm-backward launch, which leaves most CUs idle; the
// optimiser launch at the end of the step then covers only the encoder / latent / decoder chunks.
// optimiser arguments of a launch that carries chunks of the heads' update as riders (norms from the products' partials, or from the
// sums the reduce riders of an earlier launch left)
static void fill_rider_adam(smx_model* m, AdamArgs& a) {
  fill_adam_args(m, a);
  a.use_sq = 1;
  for (size_t t = 0; t < m->tensors.size(); ++t) {
    a.sq_first[t] = m->sq_first[t]; a.sq_count[t] = m->sq_count[t];
    if (m->sq_reduced[t]) { a.sq_first[t] = m->sq_total_first + (int)t * SMX_SQR_PER_TENSOR; a.sq_count[t] = m->sq_reduced[t]; }   // (summed by the launch before)
  }
  a.master = nullptr; a.with_metrics = 0;
}

// wide panels: the latent head's backward product (four workgroups of its own) takes the first adam_ride_b of the waiting chunks;
// `store` must stay alive until the product is launched
void take_adam_riders(smx_model* m, GemmArgs& h, AdamArgs& store) {
  const int n = std::min(m->adam_ride_b, m->adam_rest_to - m->adam_rest_from);
  m->adam_ride_b = 0;
  if (n <= 0) return;
  fill_rider_adam(m, store);
  h.ride_adam = &store; h.ride_first = m->adam_rest_from; h.ride_count = n;
  if (m->adam_early_from < 0) m->adam_early_from = m->adam_rest_from;
  m->adam_rest_from += n;
  m->adam_early_to = m->adam_rest_from;
}

void attach_early_adam(smx_model* m, BnBwdArgs& b) {
  if (!m->adam_early_pending) {
    // the second part of the heads' update: riders of the NEXT BatchNorm-backward launch of the step (see below)
    if (m->adam_ride_b > 0) { m->adam_rest_to = std::max(m->adam_rest_from, m->adam_rest_to - m->adam_ride_b); m->adam_ride_b = 0; }   // (no product took its share)
    if (m->adam_rest_to > m->adam_rest_from) {
      fill_rider_adam(m, b.adam);
      b.adam_first = m->adam_rest_from; b.adam_count = m->adam_rest_to - m->adam_rest_from;
      if (m->adam_early_from < 0) m->adam_early_from = m->adam_rest_from;
      m->adam_early_to = m->adam_rest_to;
      m->adam_rest_from = m->adam_rest_to = 0;
    }
    return;
  }
  m->adam_early_pending = false;
  static const bool off = tuning_on("no_adam_early");
  if (off || dp_active(m) || !m->sq_slots || m->chunk_first_head >= m->n_chunks || tuning_on("no_sq_partials")) return;
  for (size_t t = (size_t)m->t_outW[0]; t < m->tensors.size(); ++t)   // head tensors are the last ones of the manifest
    if (m->sq_count[t] == 0 && m->tensors[t].count > SMX_SQ_SMALL_TENSOR) return;
  // riders use half of a 512-thread BatchNorm workgroup: fine while the heads' update is a few MB (C2: 22 MB, hidden
  // under the launch), but at the 20 000-gene width it ran at 2.8 TB/s against 6.2 TB/s for the optimiser's own launch.
  // There only the heads' sum-of-squares slots are reduced here (one rider workgroup per tensor with many slots:
  // 30 000 for the output head at 20 000 genes), so that each of the optimiser's ~1900 workgroups for that tensor
  // reads ONE number instead of sweeping all of them (225 MB of L2 reads, 66 -> 5x us of the optimiser launch).
  if ((long)(m->n_chunks - m->chunk_first_head) * m->chunks_floats > 512L * 4096) {
    fill_adam_args(m, b.adam);
    for (size_t t = 0; t < m->tensors.size(); ++t) { b.adam.sq_first[t] = m->sq_first[t]; b.adam.sq_count[t] = m->sq_count[t]; }
    for (size_t t = (size_t)m->t_outW[0]; t < m->tensors.size(); ++t) {
      const int cnt = m->sq_count[t];
      const int R = std::min(SMX_SQR_PER_TENSOR, (cnt + SMX_SQR_MIN_SLOTS - 1) / SMX_SQR_MIN_SLOTS);
      if (cnt <= SMX_SQR_MIN_SLOTS || b.sqr_count + R > SMX_SQR_MAX) continue;
      const int seg = ((cnt + R - 1) / R + 255) / 256 * 256;
      int r = 0;
      for (int lo = 0; lo < cnt; lo += seg, ++r) {
        const int i = b.sqr_count++;
        b.sqr_first[i] = m->sq_first[t] + lo; b.sqr_n[i] = std::min(seg, cnt - lo); b.sqr_dst[i] = (int)t * SMX_SQR_PER_TENSOR + r;
      }
      m->sq_reduced[t] = (char)r;   // the optimiser reads r partial sums for this tensor
    }
    b.sq_total = m->sq_slots + m->sq_total_first;
    // ... and a share of the heads' chunks rides with the NEXT BatchNorm-backward launch as full 512-thread workgroups (the norms are
    // single numbers by then): SMX_ADAM_WIDE_SHARE of them, the optimiser launch keeps the rest
    static const float share = (float)tuning("adam_wide_share", 0.3f);
    // ... and SMX_ADAM_WIDE_SHARE_B of them before that with the latent head's backward product (take_adam_riders)
    static const float share_b = (float)tuning("adam_wide_share_b", 0.1f);
    const int early_to = m->lab_deferred ? m->chunk_first_label : m->n_chunks;
    const int n = (int)((early_to - m->chunk_first_head) * std::min(std::max(share, 0.f), 1.f));
    const int nb = std::min((int)((early_to - m->chunk_first_head) * std::min(std::max(share_b, 0.f), 1.f)), early_to - m->chunk_first_head - n);
    if (n + nb > 0) { m->adam_rest_from = m->chunk_first_head; m->adam_rest_to = m->chunk_first_head + n + nb; m->adam_ride_b = nb; }
    return;
  }
  fill_adam_args(m, b.adam);
  b.adam.use_sq = 1;
  for (size_t t = 0; t < m->tensors.size(); ++t) { b.adam.sq_first[t] = m->sq_first[t]; b.adam.sq_count[t] = m->sq_count[t]; }
  b.adam.master = nullptr; b.adam.with_metrics = 0;
  // (label heads whose weight gradients come with the grouped launch at the END of the backward pass stay with the
  // optimiser launch)
  const int early_to = m->lab_deferred ? m->chunk_first_label : m->n_chunks;
  // the riders (22 MB of optimiser traffic at the benchmark size) set the duration of the launch that carries them (10 us against
  // 6 for its own work) while the NEXT BatchNorm-backward launch of the step leaves the chip as idle: split them over the two
  // (SMX_ADAM_SPLIT = share of the first, default 0.5; a step with one such launch keeps them all, the final launch takes what
  // nobody carried)
  static const float split = (float)tuning("adam_split", 0.5f);
  const int total = early_to - m->chunk_first_head;
  const int first = std::max(1, std::min(total, (int)(total * std::min(std::max(split, 0.f), 1.f) + 0.5f)));
  b.adam_first = m->chunk_first_head;
  b.adam_count = first;
  m->adam_early_from = m->chunk_first_head; m->adam_early_to = m->chunk_first_head + first;
  m->adam_rest_from = m->chunk_first_head + first; m->adam_rest_to = early_to;
  // ... of which SMX_ADAM_SPLIT_B of the total go with the latent head's backward product between the two (take_adam_riders)
  static const float split_b = (float)tuning("adam_split_b", 0.f);
  m->adam_ride_b = std::min((int)(total * std::min(std::max(split_b, 0.f), 1.f)), m->adam_rest_to - m->adam_rest_from);
}

// ---- wide panels, one GPU, eager steps: the heads' update as a background sweep ----
// dW / db of the output head are final when smx_headfused.hip's launch ends, and nobody reads the head's parameters before the NEXT step's
// output head: clip + Adam for the head's chunks (3/4 of the parameters at the C5 width) run on a second stream as `wgs` persistent
// workgroups beside the backward chain, the optimiser launch (which skips them) and the next step's encoder and decoder -- launches that
// leave most of the memory system idle.  The next reader of the head (forward_pass, below the decoder) and the end of every
// smx_train_steps call wait for the sweep's event, so every other entry point finds the stream order it always had.
// What it reads stays put meanwhile: the head's gradients and sum-of-squares slots are rewritten by the next output head only, and the
// step state of parity p by the optimiser launch of the step after next.
// It pays from ~6 M head parameters: a second queue with work in it costs every launch of the main stream ~1.5 us (the step at 4096 genes:
// 99.8 -> 117.3 us with the sweep, at 12 000: 139.9 -> 145.3, at 20 000: 181.6 -> 177.7; tools/head_fused_width_ab.py), which the hidden update
// has to buy back.  Workgroups: one per SMX_HEAD_SWEEP_CHUNKS_PER_WG chunks -- the sweep should last most of the window between two output heads.
static int head_sweep_wgs(const smx_model* m) {
  if (!m->flags.head_sweep) return 0;
  const int chunks = m->n_chunks - m->chunk_first_head;
  if (chunks < (int)tuning("adam_sweep_min_chunks", SMX_HEAD_SWEEP_MIN_CHUNKS)) return 0;
  const int forced = (int)tuning("adam_sweep_wgs", 0);
  return forced > 0 ? forced : std::min(std::max(chunks / SMX_HEAD_SWEEP_CHUNKS_PER_WG, 64), 256);
}
// the main stream behind the sweep (its event)
int head_sweep_join(smx_model* m) {
  if (!m->sweep_pending) return SMX_OK;
  m->sweep_pending = false;
  SMX_HIP(hipStreamWaitEvent(m->st, m->ev_sweep, 0));
  return SMX_OK;
}
// decided where the output head is launched (forward_pass), acted on where its gradients are known to be final (backward_pass)
// (With label heads -- SISUA, SCALAR, observed outputs: round 6 -- the sweep covers the OUTPUT head's chunks only: the label heads' gradients come from the
// grouped launch of the backward pass, which the second stream does not wait for; their chunks stay with the optimiser launch.)
static int head_sweep_end(const smx_model* m) { return m->n_heads > 0 ? m->chunk_first_label : m->n_chunks; }
static bool head_sweep_ok(smx_model* m) {
  if (head_sweep_wgs(m) <= 0 || dp_active(m) || m->capturing || !m->timing_label.empty() || m->use_injected) return false;
  if (!m->sq_slots || m->chunk_first_head >= head_sweep_end(m) || tuning_on("no_sq_partials") || tuning_on("no_adam_early")) return false;
  const size_t t_end = m->n_heads > 0 ? (size_t)m->t_labW[0] : m->tensors.size();
  for (size_t t = (size_t)m->t_outW[0] + 1; t < t_end; ++t)   // (W_out's slots come from the output head's launch itself)
    if (m->sq_count[t] == 0 && m->tensors[t].count > SMX_SQ_SMALL_TENSOR) return false;
  return true;
}
static int head_sweep_prepare(smx_model* m) {
  // one GPU: both ends of these events are queues of THIS device -- the kernels' own agent-scope release / acquire orders their memory, and the
  // system-scope fence an event record carries by default costs ~1.5 us at either end of the output head (c5-shard 156 -> 153 us, same bits;
  // knob event_system_fence).  With a communicator attached (peers read what the chain's collective sends) the default stays.
  const int mode = (dp_active(m) || tuning_on("event_system_fence")) ? 1 : 0;
  if (m->st_side && m->ev_mode == mode) return SMX_OK;
  if (!m->st_side) {
    int lo = 0, hi = 0;
    SMX_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    SMX_HIP(hipStreamCreateWithPriority(&m->st_side, hipStreamNonBlocking, lo));
  }
  if (m->ev_sweep) { hipEventDestroy(m->ev_sweep); m->ev_sweep = nullptr; }
  if (m->ev_hf) { hipEventDestroy(m->ev_hf); m->ev_hf = nullptr; }
  const unsigned fl = hipEventDisableTiming | (mode ? 0u : (unsigned)hipEventDisableSystemFence);
  SMX_HIP(hipEventCreateWithFlags(&m->ev_sweep, fl));
  SMX_HIP(hipEventCreateWithFlags(&m->ev_hf, fl));
  m->ev_mode = mode;
  return SMX_OK;
}
// behind the output head's launch (ev_hf) on the second stream
static int head_sweep_start(smx_model* m) {
  AdamArgs a;
  fill_rider_adam(m, a);
  static const bool skip = tuning_on("skip_head_adam");   // timing only (WRONG results): the heads are never updated -- what their update costs the step
  if (!skip) {
    SMX_HIP(hipStreamWaitEvent(m->st_side, m->ev_hf, 0));
    SMX_CHECK(launch_adam_sweep(m->st_side, a, m->chunk_first_head, head_sweep_end(m) - m->chunk_first_head, head_sweep_wgs(m)));
    SMX_HIP(hipEventRecord(m->ev_sweep, m->st_side));
    m->sweep_pending = true;
  }
  // the optimiser launch skips these chunks; no launch of the backward chain carries any of them
  m->adam_early_from = m->chunk_first_head; m->adam_early_to = head_sweep_end(m);
  m->adam_early_pending = false; m->adam_rest_from = m->adam_rest_to = 0; m->adam_ride_b = 0;
  return SMX_OK;
}

// Data parallel, two buckets (smx_comm.hip: dp_chain_ok): the heads' part of the step as ONE chain on the communication stream, started
// where the heads' gradients are final (`after`: an event of the model's stream recorded there) --
//   all-reduce of the head bucket (its own communicator)  ->  the chunks' sums of squares of the REDUCED gradient  ->  clip + Adam of the
//   heads' chunks (the sweep above, norms from those sums)
// -- and joined where the background sweep is joined: in front of the next step's output head and at the end of every smx_train_steps
// call.  The model's stream all-reduces the front bucket [encoder / latent / decoder | BatchNorm statistics | ELBO scalars] itself and
// updates the front chunks; it never waits for the communication stream inside a step (round 4's two-bucket form did, twice: +32-35 us on
// one rank).  Same arithmetic as the one-bucket step element by element (sum over ranks in RCCL's order, norm of the reduced gradient,
// clip, Adam); the per-tensor norm is summed per chunk, then over the tensor's chunks, as the optimiser launch does.
int dp_chain_start(smx_model* m) {
  if (m->chain_started) return SMX_OK;
  SMX_CHECK(head_sweep_prepare(m));
  SMX_HIP(hipEventRecord(m->ev_hf, m->st));
  SMX_HIP(hipStreamWaitEvent(m->st_comm, m->ev_hf, 0));
  AdamArgs a;
  fill_adam_args(m, a);
  a.use_sq = 0; a.master = nullptr; a.with_metrics = 0;
  const int first = m->chunk_first_head, count = m->n_chunks - m->chunk_first_head;
  const int forced = (int)tuning("adam_sweep_wgs", 0);
  const int wgs = forced > 0 ? forced : std::min(std::max(count / SMX_HEAD_SWEEP_CHUNKS_PER_WG, 64), 256);
  if (m->flags.opt_shard && (m->world > 1 || m->dp_force) && dp_shard_available(m)) {
    // The heads' optimiser state SHARDED over the ranks (flag opt_shard): the bucket is world slices of equal length (cut
    // at a 64-float boundary, through chunks where it falls; the flat buffers end in SMX_SHARD_SLACK floats so that the last slice exists);
    //   reduce-scatter (rank r gets the sums of slice r)  ->  per chunk, the sum of squares of its part inside the slice  ->  all-reduce of
    //   those partials (a few KB: the chunks' sums of squares of the whole reduced gradient, hence every tensor's norm, on every rank)  ->
    //   clip + Adam of the slice  ->  all-gather of the updated parameters.
    // The same bytes on the wire as the all-reduce (a ring all-reduce IS these two halves); clip + Adam read and write 28 bytes per parameter
    // of 1 / world of the heads instead of all of them (BASELINE configs[4] on 8 GPUs: 215 -> 27 MB per GPU and step).  Every updated element
    // is computed by ONE rank and copied: the replicas stay bit-identical.  The moments outside the slice go stale: smx_opt_gather.
    const size_t slice = (((size_t)m->bucket1_count + m->world - 1) / m->world + 63) / 64 * 64;
    SMX_REQUIRE((size_t)m->world * slice <= (size_t)m->bucket1_count + SMX_SHARD_SLACK, "opt_shard: too many ranks for the buffers' slack");
    SMX_CHECK(dp_reduce_scatter(m, m->grads + m->bucket1_off, slice, m->st_comm, true));
    a.partial = m->shard_partial;
    a.shard_lo = (long)(m->bucket1_off + (size_t)m->rank * slice);
    a.shard_hi = (long)std::min(m->bucket1_off + ((size_t)m->rank + 1) * slice, m->bucket1_off + (size_t)m->bucket1_count);
    SMX_CHECK(launch_grad_sqsum_shard(m->st_comm, a, first, count));
    SMX_CHECK(dp_allreduce_buf(m, m->shard_partial + first, (size_t)count, m->st_comm, true));
    SMX_CHECK(launch_head_norms(m->st_comm, a, first, count));
    if (a.shard_hi > a.shard_lo) SMX_CHECK(launch_adam_shard(m->st_comm, a, first, count, wgs));
    SMX_CHECK(dp_all_gather(m, m->params + m->bucket1_off, slice, m->st_comm, true));
    m->opt_stale = true;
  } else {
    SMX_CHECK(dp_allreduce(m, m->bucket1_off, m->bucket1_count, m->st_comm, true));
    SMX_CHECK(launch_grad_sqsum_range(m->st_comm, a, first, count));
    SMX_CHECK(launch_adam_sweep(m->st_comm, a, first, count, wgs));
  }
  SMX_HIP(hipEventRecord(m->ev_sweep, m->st_comm));
  m->sweep_pending = true;
  m->chain_started = true;
  // the optimiser launch skips these chunks; no launch of the backward chain carries any of them
  m->adam_early_from = m->chunk_first_head; m->adam_early_to = m->n_chunks;
  m->adam_rest_from = m->adam_rest_to = 0; m->adam_ride_b = 0;
  return SMX_OK;
}

// ask the product that writes the gradient of tensor t for sum-of-squares partials
void want_sq(smx_model* m, GemmArgs& g, int t) {
  if (!m->sq_slots || tuning_on("no_sq_partials")) return;   // read per call: tests toggle it
  g.sq_part = m->sq_slots + m->sq_first[(size_t)t];
  g.sq_count = &m->sq_count[(size_t)t];
}

// backward through an MLP.  d(out of last layer) arrives as `n_slabs` slabs in m->slab.
// Leaves d(input of first layer) as slabs in m->slab unless skip_input_grad.
int mlp_backward(smx_model* m, std::vector<MlpLayer>& mlp, const Pass& ps, const float* in0, int ld0, bool in_is_x,
                 int n_slabs, bool skip_input_grad, int* out_slabs, const char* label_dw0,
                 const EpiLatentBwd* lat_epi = nullptr, GemmArgs* defer_dw0 = nullptr,
                 const BnBwdArgs* grad_front = nullptr, std::vector<GemmArgs>* defer = nullptr,
                 std::vector<MlpLayer>* twin = nullptr, const BnBwdArgs* twin_front = nullptr, bool* twin_done = nullptr,
                 bool last_bn_done = false) {
  // grad_front: the LAST layer's BatchNorm-backward launch computes its incoming gradient itself (fD fW^T as dot
  // products) instead of reading slabs.  defer: weight-gradient products that nothing later in the backward pass
  // reads are appended there instead of being launched (the caller runs them as ONE grouped launch at the end).
  // twin / twin_front: another MLP whose last layer's BatchNorm-backward (also with a gradient front) is independent of
  // this one's: both in ONE launch (*twin_done); the caller then walks the twin with last_bn_done = true.
  auto make_b = [&](MlpLayer& L, int slabs, const BnBwdArgs* front) {
    BnBwdArgs b;
    b.dout = m->slab; b.n_slabs = slabs; b.slab_stride = (long)ps.B * L.out_p; b.ld = L.out_p;
    b.out = L.out_buf; b.xhat = L.xhat; b.inv_std = L.inv_std;
    b.B = ps.B; b.H = L.out; b.Hp = L.out_p; b.batchnorm = L.bn >= 0; b.training = ps.training; b.leak = L.leak;
    b.drop_scale = (ps.training && L.drop_p > 0.f) ? 1.f / (1.f - L.drop_p) : 1.f;
    b.dpre = L.dpre;
    if (L.bn >= 0) { b.gamma = P_(m, L.tGamma); b.dgamma = G_(m, L.tGamma); b.dbeta = G_(m, L.tBeta); }
    else b.dbias = G_(m, L.tBias);
    if (front) {
      b.front = 1; b.fD = front->fD; b.fld = front->fld; b.fW = front->fW; b.fldw = front->fldw; b.fK = front->fK;
      b.fold_dz = front->fold_dz; b.zD = front->zD; b.zld = front->zld; b.zW = front->zW; b.zldw = front->zldw; b.zlb = front->zlb;
    }
    if (&mlp == &m->dec && &L == &mlp.back() && m->wide_dd_slabs > 0 && !front) {   // the one-launch head's slabs, column-major
      b.dout = m->wide_dd_src; b.n_slabs = m->wide_dd_slabs; b.slab_stride = m->wide_dd_stride; b.wide = 1;
    }
    return b;
  };
  if (twin_done) *twin_done = false;
  BnBwdArgs carried;            // gradient front handed from layer i + 1 to layer i (hidden layers up to 64 wide)
  bool have_carried = false;
  bool dpre_done = false;       // layer i's d pre-activation was written by the d in product of layer i + 1 (activation epilogue)
  for (int i = (int)mlp.size() - 1; i >= 0; --i) {
    MlpLayer& L = mlp[i];
    const TensorInfo& tw = m->tensors[L.tW];
    const bool last = (i == (int)mlp.size() - 1);
    BnBwdArgs b = make_b(L, n_slabs, (grad_front && last) ? grad_front : (have_carried ? &carried : nullptr));
    have_carried = false;
    const bool dpre_ready = dpre_done;
    dpre_done = false;
    if ((last && last_bn_done) || dpre_ready) {
      // (this layer's BatchNorm-backward ran beside the other MLP's / its d pre-activation came with the product above)
    } else if (sync_bn_on(m, ps.training) && L.bn >= 0) {   // the ELBO scalars then go with a launch of their own (optimizer_pass)
      Timed t(m, "bn_bwd");
      m->adam_early_pending = false;
      const BnSyncArgs y = sync_args(m);
      SMX_CHECK(launch_bn_sync_bwd(m->st, b, y, 0));
      SMX_CHECK(dp_allreduce_buf(m, m->sync_buf, (size_t)y.world * 2 * L.out_p, m->st));
      SMX_CHECK(launch_bn_sync_bwd(m->st, b, y, 1));
    } else {
      if (m->metrics_before_allreduce && m->have_pending_metrics) {
        b.metrics = m->pending_metrics; b.with_metrics = 1; m->have_pending_metrics = false;
      }
      attach_early_adam(m, b);
      Timed t(m, "bn_bwd");
      const bool dual = last && b.front && b.fK <= 64 && twin && twin_front && twin_front->fK <= 64 && !twin->empty() && m->flags.twin && bn_dual_supported(ps.B) &&
                        bn_bwd_front_supported(ps.B, twin_front->fK) && twin->back().out_p % 8 == 0 &&
                        !(sync_bn_on(m, ps.training) && twin->back().bn >= 0);
      if (dual) {
        const BnBwdArgs b2 = make_b(twin->back(), 0, twin_front);
        SMX_CHECK(launch_bn_act_bwd_dual(m->st, b, b2));
        if (twin_done) *twin_done = true;
      } else {
        SMX_CHECK(launch_bn_act_bwd(m->st, b));
      }
    }
    // dW = in^T * dpre
    const bool first_x = (i == 0 && in_is_x);
    GemmArgs g;
    g.A = (i == 0) ? in0 : mlp[i - 1].out_buf; g.lda = (i == 0) ? ld0 : mlp[i - 1].out_p; g.a_kmajor = 1;
    g.B = L.dpre; g.ldb = L.out_p;
    g.C = G_(m, L.tW); g.ldc = tw.ld;
    g.M = L.in_p; g.N = L.out_p; g.K = ps.B;
    want_sq(m, g, L.tW);
    if (dpre_ready) g.colsum = G_(m, L.tBias);   // (no bias / activation backward launch ran: the bias gradient is the column sum of d pre)
    if (first_x) {
      g.use_xform = 1;
      g.xf.rows = ps.xrows; g.xf.u16 = ps.x_u16; g.xf.log1p = m->cfg.log_norm; g.xf.cell_base = ps.cell_base;
      if (ps.training && m->cfg.input_dropout > 0.f) {
        g.xf.drop_p = m->cfg.input_dropout; g.xf.drop_scale = 1.f / (1.f - m->cfg.input_dropout);
        g.xf.nk = make_key(m, ST_INPUT_DROPOUT, ps.sample, true);
        if (const Injected* ij = inj(m, ST_INPUT_DROPOUT)) { g.xf.inj_mask = ij->d; g.xf.inj_ld = ij->ld; }
      }
    }
    if (i == 0 && skip_input_grad) {
      if (defer_dw0) *defer_dw0 = g;   // the caller launches it (possibly grouped with another first-layer gradient)
      else {
        Timed t(m, first_x ? label_dw0 : "gemm_mlp_dw");
        SMX_CHECK(launch_gemm(m->st, g));
      }
      n_slabs = 0;
      break;
    }
    // hidden layers 32 / 64 / 128 wide: the layer below takes d in = dpre W^T as the gradient front of its BatchNorm-backward
    // launch (dot products over K = this layer's width) and d W joins the grouped launch at the end -- no product launch
    if (defer && i > 0 && !(i == 0 && lat_epi) && m->flags.bwd_front && (L.out_p == 32 || L.out_p == 64 || L.out_p == 128) &&
        bn_bwd_front_supported(ps.B, L.out_p) && mlp[i - 1].out_p % 8 == 0 && (tw.ld % 4) == 0 && (L.out_p % 4) == 0 &&
        !(sync_bn_on(m, ps.training) && mlp[i - 1].bn >= 0)) {
      defer->push_back(g);
      carried = BnBwdArgs();
      carried.fD = L.dpre; carried.fld = L.out_p; carried.fW = P_(m, L.tW); carried.fldw = tw.ld; carried.fK = L.out_p;
      have_carried = true;
      n_slabs = 0;
      continue;
    }
    // d in = dpre * W^T  -> slabs; independent of dW: one grouped launch for both
    GemmArgs h;
    h.A = L.dpre; h.lda = L.out_p; h.B = P_(m, L.tW); h.ldb = tw.ld; h.b_nmajor = 1;
    h.M = ps.B; h.N = L.in_p; h.K = L.out_p;
    h.C = m->slab; h.ldc = L.in_p; h.slab_stride = (long)ps.B * L.in_p;
    h.split_k = suggest_split_k(ps.B, L.in_p, L.out_p);
    SMX_REQUIRE((size_t)h.split_k * (size_t)h.slab_stride <= m->slab_cap, "split-K slabs exceed the slab buffer");
    if (i == 0 && lat_epi) {  // d z feeds the latent head only: run its backward in the epilogue
      h.epi = 2; h.lb = *lat_epi; h.split_k = 1; h.tile = TILE_32x32_K4;
    }
    // the layer below has neither BatchNorm nor dropout: its activation's derivative goes into this product's store path
    // and the result IS its d pre-activation (its bias gradient: the column sums its weight-gradient product takes along)
    if (i > 0 && m->flags.act_epilogue && mlp[i - 1].bn < 0 && !(ps.training && mlp[i - 1].drop_p > 0.f) && h.split_k <= 1) {
      MlpLayer& Lo = mlp[i - 1];
      h.split_k = 1; h.act = 2; h.leak = Lo.leak; h.act_out = Lo.out_buf; h.act_ld = Lo.out_p;
      h.C = Lo.dpre; h.ldc = Lo.out_p; h.slab_stride = 0;
      dpre_done = true;
    }
    int effs[2] = {1, 1};
    if (defer && i == 0 && lat_epi && m->fold_dz_now) {
      // fold_dz: no launch here -- the encoder's last BatchNorm-backward launch computes d z and the latent head's backward itself
      // (BnBwdArgs::fold_dz, smx_kernels.hip: fold_dz_tile); d W joins the final grouped launch as before
      defer->push_back(g);
    } else if (defer && i == 0 && lat_epi) {   // d z (+ latent-head backward) alone; d W joins the final grouped launch
      defer->push_back(g);
      AdamArgs riders;
      if (m->adam_ride_b > 0) take_adam_riders(m, h, riders);
      Timed t(m, "gemm_mlp_bwd");
      SMX_CHECK(launch_gemm_group(m->st, &h, 1, effs + 1));
    } else {
      GemmArgs pair[2] = {g, h};
      Timed t(m, "gemm_mlp_bwd");
      SMX_CHECK(launch_gemm_group(m->st, pair, 2, effs));
    }
    const int eff = effs[1];
    n_slabs = eff;
  }
  if (out_slabs) *out_slabs = n_slabs;
  return SMX_OK;
}

// Training steps of the count heads with raw parameter planes (VAE / DCA / SISUA): output product + likelihood +
// dP in ONE wide kernel, P never materialised (smx_headloss.hip).  SMX_NO_HEAD_LOSS=1 keeps the product / loss
// kernel pair (what eval, predict and the scoring paths always use).
bool use_head_loss(const smx_model* m, int B) {
  if (!m->flags.head_loss || m->scvi || m->dec.empty() || m->k < 2) return false;   // (k = 1: the 'mse' output, product + loss kernel pair)
  return head_loss_supported(B, m->dec.back().out_p, m->Gp);
}

// Whether a training step of B cells takes the ONE-launch output head (smx_headfused.hip): the one predicate behind forward_pass's
// choice and smx_head_fused_bytes (two copies of it had drifted apart).  flags.head_bwd is part of it: the fused launch
// never stores dP, so the separate-launch backward forms (head_bwd = 0) cannot follow it.  Label heads do not stand in the way since
// round 5: their products run as the grouped launch of the backward pass (they cannot ride with a head launch that is not there).
// ... the launch's d d as column-major slabs that the decoder's BatchNorm-backward launch sums itself (no reduce launch, nothing in m->slab): at most 128
// cells, no label slabs beside them, no SyncBatchNorm on that layer
static bool head_fused_wide_dd(const smx_model* m, int B, bool training) {
  const MlpLayer& dL = m->dec.back();
  return m->n_heads == 0 && !(sync_bn_on(m, training) && dL.bn >= 0) && bn_wide_supported(B, dL.out_p, head_fused_grid(m->Gp)) &&
         (size_t)head_fused_grid(m->Gp) * 128 * 128 <= m->bigk_floats;
}
bool head_fused_ok(const smx_model* m, int B) {
  if (!use_head_loss(m, B) || !m->flags.head_fused || !m->flags.head_bwd || !m->hf_tab || !m->bigk_part) return false;
  // FactorVAE (round 6): the discriminator's passes use m->slab between the head's launch and the decoder's backward -- fine where the head's d d
  // does not live there (the column-major slab form); with an observed output beside the genes (its d d arrives as a further slab) the separate launches stay
  if (m->fvae && !head_fused_wide_dd(m, B, true)) return false;
  if (m->k > 3 || !m->out_has_W[1] || (m->k == 3 && !m->out_has_W[2])) return false;
  const MlpLayer& dL = m->dec.back();
  const TensorInfo& tw = m->tensors[m->t_outW[0]];
  const bool b3 = m->flags.bf16x3 < 0 ? use_bf16x3((long)B * m->Gp * m->k) : m->flags.bf16x3 != 0;
  return b3 && dL.out_p == 128 && tw.ld == (long)m->k * m->Gp && head_fused_supported(B, dL.out_p, m->Gp, m->k) && head_bwd_supported(B, dL.out_p, m->Gp) &&
         (size_t)head_fused_grid(m->Gp) * (size_t)B * 128 <= m->bigk_floats;
}

// arguments of the row-local scvi head launch of a training step; returns whether that launch applies
// (out == nullptr: only the test)
static bool scvi_train_args(smx_model* m, const Pass& ps, ScviTrainArgs* out) {
  const smx_config& c = m->cfg;
  if (!m->scvi || !m->flags.scvi_fused || m->encl.empty()) return false;
  const MlpLayer& lL = m->encl.back();
  const TensorInfo& twl = m->tensors[m->t_latlW];
  ScviTrainArgs a;
  const long ldp = (long)m->k * m->Gp;
  a.raw = m->raw; a.ld = ldp; a.plane_stride = m->Gp; a.B = ps.B; a.G = m->G; a.Gp = m->Gp; a.likelihood = c.likelihood;
  a.X = ps.Xsrc; a.ldx = m->Gp; a.x_u16 = ps.x_u16; a.rows = ps.rows; a.x_identity = (ps.rows != nullptr && ps.xrows == nullptr) ? 1 : 0;
  a.clip_library = c.clip_library; a.grad_scale = -1.f / (float)ps.global_batch;
  a.draw = m->draw; a.llk_part = m->llk_part;
  a.hl = lL.out_buf; a.ldh = lL.out_p; a.Kl = lL.out_p;
  a.Wl = P_(m, m->t_latlW); a.ldwl = twl.ld; a.bl = P_(m, m->t_latlb);
  a.library = ps.lib; a.cell_base = ps.cell_base;
  a.nk = make_key(m, ST_EPS_L, ps.sample, ps.training != 0);
  if (const Injected* ij = inj(m, ST_EPS_L)) { a.inj_eps = ij->d; a.inj_ld = ij->ld; }
  a.kl_scale = c.beta / (float)ps.global_batch;
  a.latl = m->latlbuf; a.ldl = 32; a.l = m->lsmp; a.sig = m->lsig; a.eps = m->leps; a.kl = m->kl_l;
  a.dlatl = m->dlatl; a.dl = m->dl;
  if (!scvi_head_train_supported(a)) return false;
  if (out) *out = a;
  return true;
}

// mode: 0 full forward; 1 decoder only (z given in m->z); 2 resample (encoder outputs m->latbuf / m->latlbuf kept,
// only the latent draw and everything after it run again)
int factor_forward(smx_model* m, const Pass& ps, bool backward);
int forward_pass(smx_model* m, const Pass& ps, bool with_loss, bool backward, int mode) {
  const bool decode_only = (mode == 1), resample = (mode == 2);
  const bool encode_only = (mode == 3 || mode == 4);   // encoders + latent heads + latent moments / draw 0, no decoder (the stacked-draw paths)
  const bool no_moments = (mode == 4);                  // ... and not even the moments / draw 0: the caller reads the latent head's raw output only
  const smx_config& c = m->cfg;
  const float inv_gb = 1.f / (float)ps.global_batch;
  m->head_loss = false; m->head_fused = false; m->ev_hf_fresh = false; m->wide_dd_slabs = 0;
  m->ahead_front_eps = m->ahead_front_drop = false;
  m->scvi_fused = false; m->encl_twinned = false;
  bool front_ok = false; LatentArgs front_la;
  if (!decode_only) {
  // ---- encoder ----
  bool twin_done = false;
  if (!resample) SMX_CHECK(mlp_forward(m, m->enc, ps, ps.Xsrc, m->Gp, true, "gemm_enc_fwd", -1, nullptr, 0, m->scvi ? &m->encl : nullptr, &twin_done));
  m->encl_twinned = twin_done;
  const MlpLayer& eL = m->enc.back();
  const int lat_ld = m->lat_planes * m->Dp;
  if (!resample) {
    const TensorInfo& tw = m->tensors[m->t_latW];
    GemmArgs g;
    g.A = eL.out_buf; g.lda = eL.out_p; g.B = P_(m, m->t_latW); g.ldb = tw.ld;
    g.C = m->latbuf; g.ldc = lat_ld; g.M = ps.B; g.N = lat_ld; g.K = eL.out_p; g.bias = P_(m, m->t_latb);
    Timed t(m, "gemm_lat_fwd");
    SMX_CHECK(launch_gemm(m->st, g));
  }
  if (m->mixpost) {   // SCALE read literally: the draw from the mixture-density posterior and its Monte-Carlo KL (never the fused front)
    MixLatArgs ma;
    ma.lat = m->latbuf; ma.ld = lat_ld; ma.B = ps.B; ma.D = m->D; ma.Dp = m->Dp; ma.C = c.n_components;
    ma.nk = make_key(m, ST_EPS_Z, ps.sample, ps.training != 0); ma.nk_pick = make_key(m, ST_MIX_PICK, ps.sample, ps.training != 0);
    ma.rows = ps.rows; ma.cell_base = ps.cell_base;
    if (const Injected* ij = inj(m, ST_EPS_Z)) { ma.inj_eps = ij->d; ma.inj_ld = ij->ld; }
    ma.z = m->z; ma.eps = m->eps; ma.zmean = m->zmean; ma.zstd = m->sig; ma.kl = m->kl; ma.resp = m->resp; ma.pick = m->zpick;
    Timed t(m, "latent_fwd");
    SMX_CHECK(launch_mixlat_fwd(m->st, ma));
  }
  LatentArgs la;
  la.stochastic = m->stochastic; la.relu = (c.latent_activation == SMX_ACT_RELU); la.training = ps.training;
  la.lat = m->latbuf; la.ld = lat_ld; la.B = ps.B; la.D = m->D; la.Dp = m->Dp;
  la.nk = make_key(m, ST_EPS_Z, ps.sample, ps.training != 0);
  la.rows = ps.rows; la.cell_base = ps.cell_base;
  if (const Injected* ij = inj(m, ST_EPS_Z)) { la.inj_eps = ij->d; la.inj_ld = ij->ld; }
  if (m->ahead_front_eps && !la.inj_eps) { la.inj_eps = m->noise_eps; la.inj_ld = m->Dp; }
  la.z = m->z; la.sig = m->sig; la.eps = m->eps; la.kl = m->kl;
  // The latent sample + KL and the first decoder product run INSIDE the decoder's first BatchNorm launch (two
  // launches fewer) when the shapes allow; SMX_NO_FRONT=1 keeps the three-launch form.
  front_ok = !encode_only && front_shapes_ok(m, ps) && (!la.inj_eps || (la.inj_ld % 4) == 0);
  front_la = la;
  if (front_ok || m->mixpost) {
    // (launched below with the decoder / drawn above)
  } else if (no_moments && !m->scale) {
    // (the stacked scoring pass draws from the head's raw output itself: sigma, z and the KL of draw 0 would be a launch nobody reads)
  } else {
    Timed t(m, "latent_fwd");
    SMX_CHECK(launch_latent_fwd(m->st, la));
  }
  if (m->scale) {   // Monte-Carlo KL against the mixture prior at the z just drawn (overwrites the analytic KL)
    ScalePriorArgs sp;
    sp.z = m->z; sp.sig = m->sig; sp.eps = m->eps; sp.B = ps.B; sp.D = m->D; sp.Dp = m->Dp; sp.C = c.n_components;
    sp.logits = P_(m, m->t_prLogits); sp.loc = P_(m, m->t_prLoc); sp.scale_raw = P_(m, m->t_prScale);
    sp.kl = m->kl; sp.resp = m->resp; sp.dklz = m->dklz; sp.tril = m->scale_tril;
    SMX_CHECK(launch_scale_prior_fwd(m->st, sp));
  }
  // ---- scvi library latent ----
  if (m->scvi) {
    if (!resample) SMX_CHECK(mlp_forward(m, m->encl, ps, ps.Xsrc, m->Gp, true, "gemm_encl_fwd", -1, nullptr, twin_done ? 1 : 0));
    // training step: the library latent (its head as dot products, the sample, KL_l) is part of the row-local head
    // launch below (smx_scvi.hip); otherwise the product + lib_latent_fwd pair
    m->scvi_fused = with_loss && backward && mode == 0 && scvi_train_args(m, ps, nullptr);
    if (!resample && !m->scvi_fused) {
      const MlpLayer& lL = m->encl.back();
      const TensorInfo& tw = m->tensors[m->t_latlW];
      GemmArgs g;
      g.A = lL.out_buf; g.lda = lL.out_p; g.B = P_(m, m->t_latlW); g.ldb = tw.ld;
      g.C = m->latlbuf; g.ldc = 32; g.M = ps.B; g.N = 32; g.K = lL.out_p; g.bias = P_(m, m->t_latlb);
      SMX_CHECK(launch_gemm(m->st, g));
    }
    if (!m->scvi_fused) {
      LibLatentArgs ll;
      ll.latl = m->latlbuf; ll.ld = 32; ll.B = ps.B; ll.library = ps.lib; ll.rows = ps.rows; ll.cell_base = ps.cell_base;
      ll.nk = make_key(m, ST_EPS_L, ps.sample, ps.training != 0);
      if (const Injected* ij = inj(m, ST_EPS_L)) { ll.inj_eps = ij->d; ll.inj_ld = ij->ld; }
      ll.clip_library = c.clip_library;
      ll.l = m->lsmp; ll.sig = m->lsig; ll.eps = m->leps; ll.kl = m->kl_l;
      SMX_CHECK(launch_lib_latent_fwd(m->st, ll));
    }
  }
  }  // !decode_only
  if (encode_only) return SMX_OK;
  // ---- decoder ----
  SMX_CHECK(mlp_forward(m, m->dec, ps, m->z, m->Dp, false, "", -1, front_ok ? &front_la : nullptr));
  SMX_CHECK(head_sweep_join(m));   // (the heads' update of the step before, if it is still under way on the second stream)
  const MlpLayer& dL = m->dec.back();
  const long ldp = (long)m->k * m->Gp;
  if (m->scvi) {
    GemmArgs hg[3];
    int n_hg = 0;
    for (int ch = 0; ch < m->k; ++ch) {
      if (!m->out_has_W[ch]) {   // dispersion / inflation = 'share' (scvi.py:66-86): the per-gene vector in every row of the raw plane
        SMX_CHECK(launch_plane_fill(m->st, m->raw + (long)ch * m->Gp, ldp, P_(m, m->t_outb[ch]), ps.B, m->Gp, m->out_single[ch] ? 1 : 0));
        continue;
      }
      const TensorInfo& tw = m->tensors[m->t_outW[ch]];
      GemmArgs& g = hg[n_hg++];
      g.A = dL.out_buf; g.lda = dL.out_p; g.B = P_(m, m->t_outW[ch]); g.ldb = tw.ld;
      g.C = m->raw + (long)ch * m->Gp; g.ldc = (int)ldp; g.M = ps.B; g.N = m->Gp; g.K = dL.out_p;
      g.bias = P_(m, m->t_outb[ch]);
    }
    {
      // the heads read the same decoder output: pairs of them side by side in one launch
      const bool no_twin = !m->flags.twin;
      Timed t(m, "gemm_out_fwd");
      int ch = 0;
      for (; !no_twin && ch + 1 < n_hg; ch += 2) SMX_CHECK(launch_gemm_dual(m->st, hg[ch], hg[ch + 1]));
      for (; ch < n_hg; ++ch) SMX_CHECK(launch_gemm(m->st, hg[ch]));
    }
  }
  if (m->scvi && m->scvi_fused) {
    ScviTrainArgs st;
    scvi_train_args(m, ps, &st);
    // (timing mode: the idempotent launch repeated inside one event pair, as for the other likelihood kernels)
    const int reps = (!m->capturing && m->timing_label == "loss") ? m->timing_reps : 1;
    Timed t(m, "loss");
    for (int r = 0; r < reps; ++r) SMX_CHECK(launch_scvi_head_train(m->st, st));
  } else if (m->scvi) {
    ScviHeadArgs sh;
    sh.raw = m->raw; sh.planes = m->P; sh.ld = ldp; sh.plane_stride = m->Gp; sh.B = ps.B; sh.G = m->G; sh.Gp = m->Gp;
    sh.k = m->k; sh.l = m->lsmp; sh.clip_library = c.clip_library; sh.rho_raw = m->rho;
    SMX_CHECK(launch_scvi_head_fwd(m->st, sh));
  } else if ((m->head_loss = (with_loss && backward && use_head_loss(m, ps.B)))) {
    // the product runs below, fused with the likelihood
  } else {
    const TensorInfo& tw = m->tensors[m->t_outW[0]];
    GemmArgs g;
    g.A = dL.out_buf; g.lda = dL.out_p; g.B = P_(m, m->t_outW[0]); g.ldb = tw.ld;
    g.C = m->P; g.ldc = (int)ldp; g.M = ps.B; g.N = (int)ldp; g.K = dL.out_p; g.bias = P_(m, m->t_outb[0]);
    Timed t(m, "gemm_out_fwd");
    SMX_CHECK(launch_gemm(m->st, g));
  }
  // ---- label heads (raw outputs) ----
  for (int j = 0; j < m->n_heads; ++j) {
    const TensorInfo& tw = m->tensors[m->t_labW[j]];
    GemmArgs g;
    g.A = dL.out_buf; g.lda = dL.out_p; g.B = P_(m, m->t_labW[j]); g.ldb = tw.ld;
    g.C = m->laby_raw[j]; g.ldc = tw.ld; g.M = ps.B; g.N = tw.ld; g.K = dL.out_p; g.bias = P_(m, m->t_labb[j]);
    SMX_CHECK(launch_gemm(m->st, g));
  }
  if (!with_loss) return SMX_OK;
  // ---- losses ----
  LossArgs lo;
  lo.likelihood = c.likelihood; lo.direct = m->scvi; lo.backward = backward;
  lo.X = ps.Xsrc; lo.x_u16 = ps.x_u16; lo.ldx = m->Gp; lo.rows = ps.xrows;
  lo.P = m->P; lo.ldp = ldp; lo.plane_stride = m->Gp; lo.dP = m->dP; lo.llk_part = m->llk_part;
  lo.B = ps.B; lo.G = m->G; lo.Gp = m->Gp; lo.grad_scale = -inv_gb;
  int n_llk_chunks = loss_chunks(m->Gp, ps.B);
  if (m->scvi && m->scvi_fused) {
    n_llk_chunks = 1;   // the row-local head launch above left one partial per cell
  } else if (m->head_loss) {
    const TensorInfo& tw = m->tensors[m->t_outW[0]];
    HeadLossArgs hl;
    hl.H = dL.out_buf; hl.ldh = dL.out_p; hl.W = P_(m, m->t_outW[0]); hl.ldw = tw.ld; hl.bias = P_(m, m->t_outb[0]);
    hl.X = ps.Xsrc; hl.x_u16 = ps.x_u16; hl.ldx = m->Gp; hl.rows = ps.xrows;
    hl.dP = m->dP; hl.ldp = ldp; hl.plane_stride = m->Gp; hl.llk_part = m->llk_part;
    hl.B = ps.B; hl.G = m->G; hl.Gp = m->Gp; hl.Hp = dL.out_p; hl.likelihood = c.likelihood; hl.grad_scale = -inv_gb;
    hl.bf16x3 = m->flags.bf16x3 < 0 ? (use_bf16x3((long)ps.B * m->Gp * m->k) ? 1 : 0) : m->flags.bf16x3;
    n_llk_chunks = head_loss_chunks(m->Gp);
    // a wide panel: the whole head -- product, likelihood, dW / db and the per-workgroup slabs of d d -- in ONE launch + the ordered
    // sum of the slabs (smx_headfused.hip); backward_pass then finds its head products done
    m->head_fused = false;
    m->wide_dd_slabs = 0;
    if (head_fused_ok(m, ps.B) && !(!m->capturing && m->timing_label == "out_head_product")) {
      HeadFusedArgs hf;
      hf.D = dL.out_buf; hf.ldd = dL.out_p; hf.W = hl.W; hf.ldw = tw.ld; hf.bias = hl.bias;
      hf.X = ps.Xsrc; hf.ldx = m->Gp; hf.rows = ps.xrows; hf.x_u16 = ps.x_u16;
      hf.dW = G_(m, m->t_outW[0]); hf.db = G_(m, m->t_outb[0]);
      hf.part = m->bigk_part; hf.slab_stride = (long)ps.B * dL.out_p; hf.llk_part = m->llk_part;
      // without label heads (their d d arrives as further slabs) the decoder's BatchNorm-backward launch sums the workgroups' slabs itself
      const bool wide_dd = head_fused_wide_dd(m, ps.B, ps.training != 0);
      if (wide_dd) { hf.part_colmajor = 1; hf.slab_stride = 128L * 128; }
      hf.sq_part = (m->sq_slots && !tuning_on("no_sq_partials")) ? m->sq_slots + m->sq_first[(size_t)m->t_outW[0]] : nullptr;
      hf.dtab = m->hf_tab;
      hf.B = ps.B; hf.G = m->G; hf.Gp = m->Gp; hf.likelihood = c.likelihood; hf.grad_scale = -inv_gb;
      const int reps = (!m->capturing && m->timing_label == "out_head") ? m->timing_reps : 1;   // idempotent
      int n_slabs = 0;
      // this step's update of the heads: a sweep on the second stream, started behind this launch -- its event is the launch's own completion
      // signal (no marker packet for the stream's next launch to wait behind; knob no_hf_ext_event: hipEventRecord)
      const bool sweep_ev = hf.sq_part && head_sweep_ok(m);
      const bool ext_ev = sweep_ev && reps == 1 && !tuning_on("no_hf_ext_event");
      if (sweep_ev) SMX_CHECK(head_sweep_prepare(m));
      {
        Timed t(m, "out_head");
        for (int r = 0; r < reps; ++r) SMX_CHECK(launch_head_fused(m->st, hf, &n_slabs, &m->head_fused_sq, ext_ev ? m->ev_hf : nullptr));
      }
      if (sweep_ev) {
        if (!ext_ev) SMX_HIP(hipEventRecord(m->ev_hf, m->st));
        m->ev_hf_fresh = true;
      }
      // data parallel, two buckets: without label heads every gradient of the head bucket is final HERE -- its chain (all-reduce, norms,
      // clip + Adam) runs beside the whole backward pass, the optimiser launch and the next step's encoder and decoder
      if (m->n_heads == 0 && dp_chain_ok(m)) SMX_CHECK(dp_chain_start(m));
      m->wide_dd_slabs = wide_dd ? n_slabs : 0; m->wide_dd_src = m->bigk_part; m->wide_dd_stride = hf.slab_stride;
      if (!wide_dd) SMX_CHECK(launch_head_fused_reduce(m->st, hf, n_slabs, m->slab));
      n_llk_chunks = head_fused_chunks(m->Gp);
      m->head_fused = true;
    } else {
    if (!m->capturing && m->timing_label == "out_head_product") {
      // timing mode: the product alone (P stored, no counts, no likelihood) -- what the fused kernel's time is
      // compared with to attribute the rest to the likelihood (bench.py, roofline)
      HeadLossArgs po = hl;
      po.product_only = 1; po.dP = m->P;
      Timed t(m, "out_head_product");
      for (int r = 0; r < m->timing_reps; ++r) SMX_CHECK(launch_out_head_loss(m->st, po));
    }
    const int reps = (!m->capturing && m->timing_label == "out_head") ? m->timing_reps : 1;   // idempotent
    Timed t(m, "out_head");
    for (int r = 0; r < reps; ++r) SMX_CHECK(launch_out_head_loss(m->st, hl));
    }
  } else {
    // timing mode: the (idempotent) kernel is launched SMX_LOSS_TIMING_REPEAT times inside one event pair so
    // the pair's own ~5 us overhead can be separated from the per-launch time (bench.py)
    const int reps = (!m->capturing && m->timing_label == "loss") ? m->timing_reps : 1;
    Timed t(m, "loss");
    for (int r = 0; r < reps; ++r) SMX_CHECK(launch_count_loss(m->st, lo));
  }
  for (int j = 0; j < m->n_heads; ++j) {
    const TensorInfo& tw = m->tensors[m->t_labW[j]];
    LabelArgs lb;
    lb.kind = c.label_llk[j]; lb.C = c.label_components[j]; lb.raw = m->laby_raw[j]; lb.ld = tw.ld; lb.Y = m->Y[j]; lb.ldy = m->lab_Pp[j];
    lb.rows = ps.rows; lb.mask = m->mask; lb.B = ps.B; lb.P = c.label_dim[j]; lb.Pp = m->lab_Pp[j];
    // an observed output variable (outputs[1:]): weight 1, every cell, its own accumulator; a label variable: alpha, the label mask
    const bool obs = j < m->n_observed;
    lb.observed = obs ? 1 : 0;
    lb.grad_scale = obs ? -inv_gb : -c.alpha * inv_gb; lb.draw = m->laby_draw[j];
    lb.llk = obs ? m->llk_o : m->llk_y; lb.add = obs ? (j > 0) : (j > m->n_observed);
    lb.backward = backward;
    SMX_CHECK(launch_label_loss(m->st, lb));
  }
  if (m->fvae) SMX_CHECK(factor_forward(m, ps, backward));
  MetricsArgs me;
  me.llk_part = m->llk_part; me.n_chunks = n_llk_chunks; me.rows = ps.rows;
  me.lgx1 = c.likelihood == SMX_LLK_MSE ? nullptr : ps.lgx1;   // (the count likelihoods' data-only constant sum_g lgamma(x + 1))
  me.llk_y = c.n_labels > m->n_observed ? m->llk_y : nullptr;
  me.llk_o = m->n_observed ? m->llk_o : nullptr;
  if (m->fvae) { me.tc = m->tc_cell; me.dl = m->dl_cell; me.gamma = c.gamma; }
  me.kl = m->stochastic ? m->kl : nullptr; me.kl_l = m->scvi ? m->kl_l : nullptr;
  me.B = ps.B; me.alpha = c.alpha; me.beta = c.beta; me.inv_global_batch = inv_gb;
  me.out = m->grads + m->tail_off_metrics;
  if (backward && !dp_active(m)) { me.hist = m->mhist; me.state = cur_state(m); }
  if (backward) {
    // training step: the scalars come from one extra workgroup of a later launch -- of the optimiser kernel, or,
    // under data parallelism (they must be in the flat buffer BEFORE the all-reduce), of the first
    // BatchNorm-backward launch
    m->pending_metrics = me;
    m->have_pending_metrics = true;
    m->metrics_before_allreduce = dp_active(m);
    return SMX_OK;
  }
  {
    Timed t(m, "metrics");
    SMX_CHECK(launch_metrics(m->st, me));
  }
  return SMX_OK;
}

// ---- FactorVAE discriminator (fvae.py:9-18; Kim & Mnih 2018, Algorithm 2) -------------------------------------------
// forward: stacked batch [z ; permute_dims(z)] through the discriminator, then the head (TC estimate, discriminator
// loss, SemiFVAE's cross-entropy, both upstream gradients).  Runs after the latent sample exists; m->slab is free then.
int factor_forward(smx_model* m, const Pass& ps, bool backward) {
  const smx_config& c = m->cfg;
  const int B = ps.B, B2 = 2 * ps.B;
  SMX_REQUIRE(B2 <= 2 * m->Bmax, "batch exceeds max_batch");
  PermuteArgs pa;
  pa.z = m->z; pa.ldz = m->Dp; pa.zz = m->zz; pa.ld = m->Dp; pa.B = B; pa.D = m->D;
  pa.nk = make_key(m, ST_PERMUTE, ps.sample, ps.training != 0);
  pa.rows = ps.rows; pa.cell_base = ps.cell_base;
  if (const Injected* ij = inj(m, ST_PERMUTE)) { pa.inj_u = ij->d; pa.inj_ld = ij->ld; }
  {
    Timed t(m, "disc_permute");
    SMX_CHECK(launch_permute_dims(m->st, pa));
  }
  Pass p2 = ps;
  p2.B = B2; p2.rows = nullptr; p2.training = 1;   // (no BatchNorm / dropout in the discriminator: the mode is immaterial)
  SMX_CHECK(mlp_forward(m, m->disc, p2, m->zz, m->Dp, false, "disc_fwd"));
  const MlpLayer& last = m->disc.back();
  const TensorInfo& tw = m->tensors[m->t_discoutW];
  GemmArgs g;
  g.A = last.out_buf; g.lda = last.out_p; g.B = P_(m, m->t_discoutW); g.ldb = tw.ld;
  g.M = B2; g.N = tw.ld; g.K = last.out_p;
  g.C = m->slab; g.ldc = tw.ld; g.slab_stride = (long)B2 * tw.ld;
  g.split_k = suggest_split_k(B2, tw.ld, last.out_p);
  SMX_REQUIRE((size_t)std::max(g.split_k, 1) * (size_t)g.slab_stride <= m->slab_cap, "split-K slabs exceed the slab buffer");
  int eff = 1;
  {
    Timed t(m, "disc_fwd");
    SMX_CHECK(launch_gemm(m->st, g, &eff));
  }
  DiscHeadArgs h;
  h.logits = m->slab; h.n_slabs = eff; h.slab_stride = g.slab_stride; h.ld = tw.ld;
  h.bias = P_(m, m->t_discoutb); h.n_out = tw.cols; h.B = B;
  h.gamma = c.gamma; h.alpha = c.alpha; h.inv_gb = 1.f / (float)ps.global_batch; h.backward = backward ? 1 : 0;
  const int jd = m->n_observed;   // SemiFVAE's label variable: behind the observed outputs in the target order
  const bool semi = c.n_labels > jd;
  if (semi && m->Y[jd] && ps.Xsrc == m->X) {   // (resident cells: their labels are; a pass over host data has none)
    for (int j = jd; j < c.n_labels; ++j) {
      h.Y[h.n_groups] = m->Y[j]; h.ldy[h.n_groups] = m->lab_Pp[j];
      h.gstart[h.n_groups + 1] = h.gstart[h.n_groups] + c.label_dim[j];
      ++h.n_groups;
    }
    h.rows = ps.rows; h.mask = m->mask;
  }
  h.u_tc = m->u_d + (size_t)2 * B * 32; h.u_d = m->u_d;   // (one buffer: rows [0, 2B) the discriminator's objective, [2B, 3B) the TC term)
  h.tc_cell = m->tc_cell; h.dl_cell = m->dl_cell; h.llk_y = semi ? m->llk_y : nullptr;
  Timed t(m, "disc_head");
  SMX_CHECK(launch_disc_head(m->st, h));
  return SMX_OK;
}

// One backward sweep of the discriminator over the first `rows` rows of the stacked batch with upstream `up`
// [rows][32] on the logits.  with_grads: the discriminator's own gradients (its objective; nothing flows into z);
// otherwise only d objective / d z, left in m->dz_tc (the VAE objective's TC term; the weights are constants of it).
int factor_sweep(smx_model* m, const Pass& ps, int rows, const float* up, bool with_grads) {
  const MlpLayer& last = m->disc.back();
  // flag bf16x3: the weight gradients (K = the stacked minibatch) through the direct-operand 32 x 32-tile kernel of
  // smx_headbwd.hip, the 1000-deep input gradients through smx_dgemm.hip -- both from bf16 MFMAs on split operands
  const bool b3 = b3_on(m, ps) && !tuning_on("no_dgemm");
  auto wgrad = [&](const GemmArgs& g) -> int {
    if (!(b3 && m->flags.wgrad && wgrad_supported(g, rows))) return launch_gemm(m->st, g);
    // a square 1000 x 1000 gradient: as up to 8 column groups of 128 in the panel form (smx_panel.h role 0: a workgroup takes 32
    // rows with the group's four column tiles -- the A tile is split once per group instead of once per 32 columns)
    if (g.M >= 512 && g.N % 128 == 0 && g.N / 128 <= SMX_GROUP_MAX && g.N > 128 && !tuning_on("no_panel")) {
      GemmArgs part[SMX_GROUP_MAX];
      int counts[SMX_GROUP_MAX];
      const int ng = g.N / 128, per = ((g.M + 31) / 32) * 8;
      for (int k = 0; k < ng; ++k) {
        part[k] = g;
        part[k].B = g.B + 128 * k; part[k].C = g.C + 128 * k; part[k].N = 128; part[k].panel_hint = 1;
        if (g.colsum) part[k].colsum = g.colsum + 128 * k;
        if (g.sq_part) { part[k].sq_part = g.sq_part + (long)per * k; part[k].sq_count = &counts[k]; }
      }
      const int rc = launch_wgrad_group(m->st, part, ng, rows, 1);
      if (rc == SMX_OK && g.sq_part && g.sq_count) *g.sq_count = per * ng;
      return rc;
    }
    return launch_wgrad_group(m->st, &g, 1, rows, 1);
  };
  const TensorInfo& two = m->tensors[m->t_discoutW];
  if (with_grads) {
    GemmArgs gw;
    gw.A = last.out_buf; gw.lda = last.out_p; gw.a_kmajor = 1; gw.B = up; gw.ldb = 32;
    gw.C = G_(m, m->t_discoutW); gw.ldc = two.ld; gw.M = last.out_p; gw.N = two.ld; gw.K = rows;
    gw.colsum = G_(m, m->t_discoutb);
    want_sq(m, gw, m->t_discoutW);
    Timed t(m, "disc_bwd");
    SMX_CHECK(wgrad(gw));
  }
  int n_slabs = 1;
  {
    GemmArgs gh;
    gh.A = up; gh.lda = 32; gh.B = P_(m, m->t_discoutW); gh.ldb = two.ld; gh.b_nmajor = 1;
    gh.M = rows; gh.N = last.out_p; gh.K = two.ld;
    gh.C = m->slab; gh.ldc = last.out_p; gh.slab_stride = (long)rows * last.out_p; gh.split_k = 1;
    Timed t(m, "disc_bwd");
    SMX_CHECK(launch_gemm(m->st, gh, &n_slabs));
  }
  // The discriminator's layers have neither BatchNorm nor dropout: below the top layer the activation's derivative runs
  // in the store path of the d-input product above (which then writes the layer's d pre-activation directly) and the bias
  // gradient is the column sum its weight-gradient product takes along -- no bias / activation backward launch per
  // layer (flag act_epilogue).  Without gradients (the TC sweep) the d pre-activations ping-pong between two scratch
  // buffers, as a product must not write the operand it reads.
  float* pong[2] = {m->disc_dpre, m->slab};
  int pp = 0;
  bool ready = false;
  float* dpre_i = nullptr;
  for (int i = (int)m->disc.size() - 1; i >= 0; --i) {
    MlpLayer& L = m->disc[i];
    const TensorInfo& tw = m->tensors[L.tW];
    if (!ready) {
      dpre_i = with_grads ? L.dpre : pong[pp];
      BnBwdArgs b;
      b.dout = m->slab; b.n_slabs = n_slabs; b.slab_stride = (long)rows * L.out_p; b.ld = L.out_p;
      b.out = L.out_buf; b.B = rows; b.H = L.out; b.Hp = L.out_p; b.batchnorm = 0; b.training = 1; b.drop_scale = 1.f; b.leak = L.leak;
      b.dpre = dpre_i;
      b.dbias = with_grads ? G_(m, L.tBias) : m->disc_db;
      Timed t(m, "disc_bwd");
      SMX_CHECK(launch_bn_act_bwd(m->st, b));
    }
    const float* in = (i == 0) ? m->zz : m->disc[i - 1].out_buf;
    const int ld_in = (i == 0) ? m->Dp : m->disc[i - 1].out_p;
    if (with_grads) {
      GemmArgs g;
      g.A = in; g.lda = ld_in; g.a_kmajor = 1; g.B = dpre_i; g.ldb = L.out_p;
      g.C = G_(m, L.tW); g.ldc = tw.ld; g.M = L.in_p; g.N = L.out_p; g.K = rows;
      if (ready) g.colsum = G_(m, L.tBias);
      want_sq(m, g, L.tW);
      Timed t(m, "disc_bwd");
      SMX_CHECK(wgrad(g));
      if (i == 0) break;   // z is a constant of the discriminator's objective
    }
    GemmArgs h;
    h.A = dpre_i; h.lda = L.out_p; h.B = P_(m, L.tW); h.ldb = tw.ld; h.b_nmajor = 1;
    h.M = rows; h.N = L.in_p; h.K = L.out_p;
    ready = false;
    if (i == 0) { h.C = m->dz_tc; h.ldc = m->Dp; h.split_k = 1; h.tile = TILE_32x32_K4; }
    else {
      h.C = m->slab; h.ldc = L.in_p; h.slab_stride = (long)rows * L.in_p;
      h.split_k = suggest_split_k(rows, L.in_p, L.out_p);
      if (b3) { GemmArgs probe = h; probe.split_k = 1; if (dgemm_supported(probe)) h.split_k = 1; }   // (the direct form splits K over its waves)
      SMX_REQUIRE((size_t)h.split_k * (size_t)h.slab_stride <= m->slab_cap, "split-K slabs exceed the slab buffer");
      if (m->flags.act_epilogue && h.split_k <= 1 && (with_grads || pong[pp ^ 1] != dpre_i)) {
        MlpLayer& Lo = m->disc[i - 1];
        float* next = with_grads ? Lo.dpre : pong[pp ^= 1];
        h.split_k = 1; h.act = 2; h.leak = Lo.leak; h.act_out = Lo.out_buf; h.act_ld = Lo.out_p;
        h.C = next; h.ldc = Lo.out_p; h.slab_stride = 0;
        ready = true;
      }
    }
    {
      Timed t(m, "disc_bwd");
      if (b3 && dgemm_supported(h)) { SMX_CHECK(launch_dgemm(m->st, h)); n_slabs = 1; }
      else SMX_CHECK(launch_gemm(m->st, h, &n_slabs));
    }
    if (ready) dpre_i = h.C;
  }
  return SMX_OK;
}

// Both backward sweeps of the discriminator as rows of ONE chain of products: rows [0, 2B) carry the discriminator's own objective
// (its weight gradients contract over exactly these rows), rows [2B, 3B) the VAE objective's TC term on the rows of z (they only
// pass through the weights: d objective / d z in the end).  The input gradients of a layer are then one launch of 3B rows
// instead of one of 2B and one of B (each a latency chain of ~8 us whatever its row count), the activation's derivative for
// the extra rows reads the forward output of row - 2B (GemmArgs::act_wrap), every bias gradient is the column sum of its
// weight-gradient product.  Needs the store-path activation (flag act_epilogue); without it: the two sweeps of factor_sweep.
static int factor_backward_stacked(smx_model* m, const Pass& ps) {
  const int B = ps.B, R2 = 2 * B, R3 = 3 * B;
  const MlpLayer& last = m->disc.back();
  const TensorInfo& two = m->tensors[m->t_discoutW];
  const bool b3 = b3_on(m, ps) && !tuning_on("no_dgemm");
  const float* up = m->u_d;   // [3B][32]
  static const bool beside_on = !tuning_on("fvae_no_beside");
  // beside: the layer's input gradient in the same launch (wgrad_dgemm_kernel) where both forms apply, right after it otherwise
  auto input_grad = [&](const GemmArgs& h) -> int {
    if (b3 && dgemm_supported(h)) return launch_dgemm(m->st, h);
    return launch_gemm(m->st, h);
  };
  auto wgrad = [&](const GemmArgs& g, const GemmArgs* beside = nullptr) -> int {
    if (!(b3 && m->flags.wgrad && wgrad_supported(g, R2))) return launch_gemm(m->st, g);
    if (g.M >= 512 && g.N % 128 == 0 && g.N / 128 <= SMX_GROUP_MAX && g.N > 128 && !tuning_on("no_panel")) {
      GemmArgs part[SMX_GROUP_MAX];
      int counts[SMX_GROUP_MAX];
      const int ng = g.N / 128, per = ((g.M + 31) / 32) * 8;
      for (int k = 0; k < ng; ++k) {
        part[k] = g;
        part[k].B = g.B + 128 * k; part[k].C = g.C + 128 * k; part[k].N = 128; part[k].panel_hint = 1;
        if (g.colsum) part[k].colsum = g.colsum + 128 * k;
        if (g.sq_part) { part[k].sq_part = g.sq_part + (long)per * k; part[k].sq_count = &counts[k]; }
      }
      const int rc = launch_wgrad_group(m->st, part, ng, R2, 1, beside);
      if (rc == SMX_OK && g.sq_part && g.sq_count) *g.sq_count = per * ng;
      return rc;
    }
    return launch_wgrad_group(m->st, &g, 1, R2, 1, beside);
  };
  Timed t(m, "disc_bwd");
  {   // the logit layer: weight gradient over the discriminator's rows, then the input gradient of all 3B rows -> d pre of the last hidden layer
    GemmArgs gw;
    gw.A = last.out_buf; gw.lda = last.out_p; gw.a_kmajor = 1; gw.B = up; gw.ldb = 32;
    gw.C = G_(m, m->t_discoutW); gw.ldc = two.ld; gw.M = last.out_p; gw.N = two.ld; gw.K = R2;
    gw.colsum = G_(m, m->t_discoutb);
    want_sq(m, gw, m->t_discoutW);
    SMX_CHECK(wgrad(gw));
    MlpLayer& Ll = m->disc.back();
    GemmArgs gh;
    gh.A = up; gh.lda = 32; gh.B = P_(m, m->t_discoutW); gh.ldb = two.ld; gh.b_nmajor = 1;
    gh.M = R3; gh.N = last.out_p; gh.K = two.ld; gh.split_k = 1;
    gh.act = 2; gh.leak = Ll.leak; gh.act_out = Ll.out_buf; gh.act_ld = Ll.out_p; gh.act_wrap = R2;
    gh.C = Ll.dpre; gh.ldc = Ll.out_p;
    SMX_CHECK(launch_gemm(m->st, gh));
  }
  for (int i = (int)m->disc.size() - 1; i >= 0; --i) {
    MlpLayer& L = m->disc[i];
    const TensorInfo& tw = m->tensors[L.tW];
    const float* in = (i == 0) ? m->zz : m->disc[i - 1].out_buf;
    const int ld_in = (i == 0) ? m->Dp : m->disc[i - 1].out_p;
    GemmArgs g;   // weight gradient (+ bias gradient as its column sum) over the discriminator's 2B rows
    g.A = in; g.lda = ld_in; g.a_kmajor = 1; g.B = L.dpre; g.ldb = L.out_p;
    g.C = G_(m, L.tW); g.ldc = tw.ld; g.M = L.in_p; g.N = L.out_p; g.K = R2;
    g.colsum = G_(m, L.tBias);
    want_sq(m, g, L.tW);
    GemmArgs h;   // input gradient
    h.B = P_(m, L.tW); h.ldb = tw.ld; h.b_nmajor = 1; h.K = L.out_p; h.split_k = 1;
    if (i == 0) {   // into z: only the TC rows (z is a constant of the discriminator's objective)
      h.A = L.dpre + (size_t)R2 * L.out_p; h.lda = L.out_p; h.M = B; h.N = L.in_p;
      h.C = m->dz_tc; h.ldc = m->Dp; h.tile = TILE_32x32_K4;
    } else {
      MlpLayer& Lo = m->disc[i - 1];
      h.A = L.dpre; h.lda = L.out_p; h.M = R3; h.N = L.in_p;
      h.act = 2; h.leak = Lo.leak; h.act_out = Lo.out_buf; h.act_ld = Lo.out_p; h.act_wrap = R2;
      h.C = Lo.dpre; h.ldc = Lo.out_p;
    }
    if (beside_on && b3 && m->flags.wgrad && wgrad_supported(g, R2) && dgemm_supported(h) && h.b_nmajor) SMX_CHECK(wgrad(g, &h));
    else { SMX_CHECK(wgrad(g)); SMX_CHECK(input_grad(h)); }
  }
  return SMX_OK;
}

int factor_backward(smx_model* m, const Pass& ps) {
  if (m->flags.act_epilogue && !tuning_on("fvae_two_sweeps")) return factor_backward_stacked(m, ps);
  SMX_CHECK(factor_sweep(m, ps, 2 * ps.B, m->u_d, true));                          // discriminator objective -> the discriminator's tensors
  SMX_CHECK(factor_sweep(m, ps, ps.B, m->u_d + (size_t)2 * ps.B * 32, false));     // gamma TC (+ alpha CE) -> d z
  return SMX_OK;
}

int backward_pass(smx_model* m, const Pass& ps) {
  const smx_config& c = m->cfg;
  std::fill(m->sq_count.begin(), m->sq_count.end(), 0);   // the products of this step report what they wrote
  std::fill(m->sq_reduced.begin(), m->sq_reduced.end(), 0);
  m->adam_early_pending = false; m->adam_rest_from = m->adam_rest_to = 0; m->adam_ride_b = 0;
  if (!m->chain_started) m->adam_early_from = -1;   // (forward_pass may have sent the heads' chunks down the data-parallel chain already)
  const float inv_gb = 1.f / (float)ps.global_batch;
  if (m->fvae) SMX_CHECK(factor_backward(m, ps));   // first: it uses the slab buffer the head's backward fills next
  const MlpLayer& dL = m->dec.back();
  const long ldp = (long)m->k * m->Gp;
  int n_slabs = 0;
  const long dd_stride = (long)ps.B * dL.out_p;
  const float* dparams = m->dP;
  std::vector<GemmArgs> lab_dw;
  m->lab_deferred = false;
  if (m->scvi) {
    ScviHeadArgs sh;
    sh.raw = m->raw; sh.planes = m->P; sh.ld = ldp; sh.plane_stride = m->Gp; sh.B = ps.B; sh.G = m->G; sh.Gp = m->Gp;
    sh.k = m->k; sh.l = m->lsmp; sh.clip_library = c.clip_library; sh.rho_raw = m->rho;
    sh.dplanes = m->dP; sh.draw = m->draw; sh.dl = m->dl;
    if (!m->scvi_fused) SMX_CHECK(launch_scvi_head_bwd(m->st, sh));   // (the row-local head launch of the forward pass left d raw and d l)
    dparams = m->draw;
  }
  const int n_heads = m->scvi ? m->k : 1;
  // count heads with raw planes: both products of the output head in one launch of the wide direct-operand kernel
  // (smx_headbwd.hip); SMX_NO_HEAD_BWD=1 or scvi: the grouped LDS-tiled products below
  // (scvi: the planes are separate head tensors -- the kernel's SEP form)
  const bool all_heads = m->out_has_W[1] && (m->k < 3 || m->out_has_W[2]);   // (scvi with a 'share'd plane: the grouped products below)
  const bool hbwd = m->flags.head_bwd && head_bwd_supported(ps.B, dL.out_p, m->Gp) && m->k >= 2 && m->k <= 3 && all_heads;
  if (hbwd) {
    const TensorInfo& tw = m->tensors[m->t_outW[0]];
    HeadBwdArgs hb;
    hb.D = dL.out_buf; hb.ldd = dL.out_p; hb.dP = dparams; hb.ldp = ldp; hb.W = P_(m, m->t_outW[0]); hb.ldw = tw.ld;
    hb.dW = G_(m, m->t_outW[0]); hb.db = G_(m, m->t_outb[0]);
    if (m->scvi) {
      hb.sep = 1;
      for (int ch = 0; ch < m->k; ++ch) {
        hb.Wp[ch] = P_(m, m->t_outW[ch]); hb.dWp[ch] = G_(m, m->t_outW[ch]); hb.dbp[ch] = G_(m, m->t_outb[ch]);
        if (m->sq_slots && !tuning_on("no_sq_partials")) {
          hb.sqp[ch] = m->sq_slots + m->sq_first[(size_t)m->t_outW[ch]]; hb.sq_countp[ch] = &m->sq_count[(size_t)m->t_outW[ch]];
        }
      }
    }
    hb.B = ps.B; hb.Hp = dL.out_p; hb.Gp = m->Gp; hb.n_planes = m->k;
    hb.bf16x3 = m->flags.bf16x3 < 0 ? (use_bf16x3((long)ps.B * m->Gp * m->k) ? 1 : 0) : m->flags.bf16x3;
    hb.n_slices = head_bwd_slices(ldp, ldp <= 8192 ? 16 : 32, &hb.k_chunk);
    hb.slab = m->slab; hb.slab_stride = dd_stride;
    SMX_REQUIRE((size_t)hb.n_slices * (size_t)dd_stride <= m->slab_cap, "split-K slabs exceed the slab buffer");
    if (!m->scvi && m->sq_slots && !tuning_on("no_sq_partials")) {
      hb.sq_part = m->sq_slots + m->sq_first[(size_t)m->t_outW[0]]; hb.sq_count = &m->sq_count[(size_t)m->t_outW[0]];
    }
    n_slabs = hb.n_slices;
    // label heads (SISUA / MISA): d d += d Y W_lab^T as extra slabs of this launch, the head's weight gradient with the
    // grouped launch at the end of the backward pass -- instead of a grouped launch of their own here (8.6 us at C4)
    if (m->n_heads > 0 && m->flags.label_ride && !m->fvae && !m->scvi && !(m->head_fused && m->head_loss)) {
      bool ok = true;
      for (int j = 0; j < m->n_heads; ++j) ok = ok && (m->tensors[m->t_labW[j]].ld % 32) == 0;
      ok = ok && (size_t)(hb.n_slices + m->n_heads) * (size_t)dd_stride <= m->slab_cap;
      if (ok) {
        for (int j = 0; j < m->n_heads; ++j) {
          const TensorInfo& tl = m->tensors[m->t_labW[j]];
          hb.xA[j] = m->laby_draw[j]; hb.xlda[j] = tl.ld; hb.xW[j] = P_(m, m->t_labW[j]); hb.xldw[j] = tl.ld; hb.xK[j] = tl.ld;
        }
        hb.n_extra = m->n_heads;
        n_slabs += m->n_heads;
        m->lab_deferred = true;
      }
    }
    // at most 128 cells, no label slabs: d d's slabs column-major, summed by the decoder's BatchNorm-backward launch as one workgroup per
    // column (bn_wide_bwd_kernel; <= 16 slabs: the additions in the order of the 8-column launch, the same bits)
    const bool dd_wide = !(m->head_fused && m->head_loss) && m->n_heads == 0 && hb.n_extra == 0 && !hb.sep && hb.n_slices <= 16 && !(sync_bn_on(m, ps.training) && dL.bn >= 0) &&
                         bn_wide_supported(ps.B, dL.out_p, hb.n_slices) && (size_t)hb.n_slices * 128 * (size_t)dL.out_p <= m->slab_cap;
    Timed t(m, "gemm_out_bwd");
    const bool fused_done = m->head_fused && m->head_loss && !hb.sep && hb.n_extra == 0;   // (forward_pass ran smx_headfused.hip: dW, db, the sum of squares and d d are there)
    if (fused_done) {
      m->head_fused_bwd_done = true;
      n_slabs = 1;
      if (hb.sq_count) *hb.sq_count = m->head_fused_sq;
    }
    // a wide head (the bf16 x 3 regime): d d = dP W^T (K = every gene of every plane) as one workgroup per K slice + a reduce
    // launch (smx_bigk.hip) -- ONE slab for the BatchNorm-backward launch; d W / d b stay with the 32 x 32-tile kernel
    bool dd_bigk = false;
    if (!fused_done && hb.bf16x3 && !hb.sep && hb.n_extra == 0 && m->bigk_part && dL.out_p <= 128 && !tuning_on("no_bigk")) {
      BigKArgs bk;
      bk.A = dparams; bk.lda = ldp; bk.Bm = P_(m, m->t_outW[0]); bk.ldb = tw.ld; bk.b_kmajor = 0;
      bk.M = ps.B; bk.N = dL.out_p; bk.K = (int)ldp; bk.ldc = dL.out_p; bk.slab_stride = dd_stride;
      bk.part = m->bigk_part; bk.out = m->slab;
      bk.n_slices = bigk_slices(bk.K, SMX_BIGK_MAX_SLICES, &bk.k_chunk);
      if (bigk_supported(bk) && (size_t)bk.n_slices * (size_t)bk.slab_stride <= m->bigk_floats) {
        if (m->n_heads == 0 && !(sync_bn_on(m, ps.training) && dL.bn >= 0) && bn_wide_supported(ps.B, dL.out_p, bk.n_slices) && (size_t)bk.n_slices * 128 * 128 <= m->bigk_floats) {
          bk.colmajor = 1; bk.slab_stride = 128L * 128;   // (no reduce launch: as behind the one-launch head)
          m->wide_dd_slabs = bk.n_slices; m->wide_dd_src = m->bigk_part; m->wide_dd_stride = bk.slab_stride;
        }
        SMX_CHECK(launch_bigk(m->st, bk));
        dd_bigk = true;
        n_slabs = 1;
      }
    }
    hb.skip_dd = dd_bigk ? 1 : 0;
    if (dd_wide && !fused_done && !dd_bigk) {
      hb.dd_colmajor = 1; hb.slab_stride = 128L * dL.out_p;
      m->wide_dd_slabs = hb.n_slices; m->wide_dd_src = m->slab; m->wide_dd_stride = hb.slab_stride;
    }
    // ... and then d W / d b with one workgroup per (gene tile, plane) that holds every row of H (smx_panel.h): the panel is
    // transformed and split once, not once per 32 rows of H
    if (fused_done) {}
    else if (dd_bigk && panel_dw_supported(hb)) SMX_CHECK(launch_panel_dw(m->st, hb));
    else if (!(hb.skip_dw && hb.skip_dd)) SMX_CHECK(launch_out_head_bwd(m->st, hb));
  }
  {
    // weight gradient and input gradient of every head read the same dP and are independent:
    // one grouped launch (dW tiles + split-K dX slabs side by side)
    std::vector<GemmArgs> grp;
    std::vector<int> is_dx;
    for (int ch = 0; ch < n_heads && !hbwd; ++ch) {
      if (m->scvi && !m->out_has_W[ch]) {   // no Dense head: the per-gene vector's gradient is the column sum of the plane's d raw
        SMX_CHECK(launch_plane_colsum(m->st, dparams + (long)ch * m->Gp, ldp, G_(m, m->t_outb[ch]), ps.B, m->Gp, m->out_single[ch] ? m->G : 0));
        continue;
      }
      const TensorInfo& tw = m->tensors[m->t_outW[ch]];
      const float* dp = dparams + (m->scvi ? (long)ch * m->Gp : 0);
      const int ncols = m->scvi ? m->Gp : (int)ldp;
      GemmArgs g;  // dW = d^T dP, db = colsum(dP)
      g.A = dL.out_buf; g.lda = dL.out_p; g.a_kmajor = 1; g.B = dp; g.ldb = (int)ldp;
      g.C = G_(m, m->t_outW[ch]); g.ldc = tw.ld; g.M = dL.out_p; g.N = ncols; g.K = ps.B;
      g.colsum = G_(m, m->t_outb[ch]);
      want_sq(m, g, m->t_outW[ch]);
      g.tile = TILE_128x32;
      grp.push_back(g); is_dx.push_back(0);
      GemmArgs h;  // dd += dP W^T
      h.A = dp; h.lda = (int)ldp; h.B = P_(m, m->t_outW[ch]); h.ldb = tw.ld; h.b_nmajor = 1;
      h.C = nullptr; h.ldc = dL.out_p; h.slab_stride = dd_stride;
      h.M = ps.B; h.N = dL.out_p; h.K = ncols;
      h.split_k = suggest_split_k(ps.B, dL.out_p, ncols);
      h.tile = TILE_32x32_K4;
      grp.push_back(h); is_dx.push_back(1);
    }
    for (int j = 0; j < m->n_heads; ++j) {
      const TensorInfo& tw = m->tensors[m->t_labW[j]];
      GemmArgs g;
      g.A = dL.out_buf; g.lda = dL.out_p; g.a_kmajor = 1; g.B = m->laby_draw[j]; g.ldb = tw.ld;
      g.C = G_(m, m->t_labW[j]); g.ldc = tw.ld; g.M = dL.out_p; g.N = tw.ld; g.K = ps.B;
      g.colsum = G_(m, m->t_labb[j]);
      want_sq(m, g, m->t_labW[j]);
      if (m->lab_deferred) { lab_dw.push_back(g); continue; }   // (d d rode with the output head's backward launch)
      grp.push_back(g); is_dx.push_back(0);
      GemmArgs h;
      h.A = m->laby_draw[j]; h.lda = tw.ld; h.B = P_(m, m->t_labW[j]); h.ldb = tw.ld; h.b_nmajor = 1;
      h.C = nullptr; h.ldc = dL.out_p; h.slab_stride = dd_stride;
      h.M = ps.B; h.N = dL.out_p; h.K = tw.ld;
      grp.push_back(h); is_dx.push_back(1);
    }
    // slab slots: split factors are known up front (launch_gemm_group recomputes the same values)
    for (size_t i = 0; i < grp.size(); ++i) {
      if (!is_dx[i]) continue;
      GemmArgs& h = grp[i];
      const int BK = 128;  // K4 tile for split products; single-slab products may take either tile
      int eff = 1;
      if (h.split_k > 1) {
        const int chunk = round_up((h.K + h.split_k - 1) / h.split_k, BK);
        eff = (h.K + chunk - 1) / chunk;
      }
      h.C = m->slab + (long)n_slabs * dd_stride;
      n_slabs += eff;
      SMX_REQUIRE((size_t)n_slabs * (size_t)dd_stride <= m->slab_cap, "split-K slabs exceed the slab buffer");
    }
    if (!grp.empty()) {
      Timed t(m, hbwd ? "gemm_lab_bwd" : "gemm_out_bwd");
      for (size_t i = 0; i < grp.size(); i += SMX_GROUP_MAX) {
        const int n = (int)std::min<size_t>(SMX_GROUP_MAX, grp.size() - i);
        SMX_CHECK(launch_gemm_group(m->st, grp.data() + i, n));
      }
    }
    m->adam_early_pending = true;   // dW / db of every head are final now
    if (m->head_fused_bwd_done && m->ev_hf_fresh) SMX_CHECK(head_sweep_start(m));
    else if (!m->head_fused_bwd_done && !m->scvi && m->sq_count[(size_t)m->t_outW[0]] > 0 && !tuning_on("no_sweep_unfused") && head_sweep_ok(m)) {
      // (round 6) ... and behind the SEPARATE head products of a wide panel too (decoder layers other than 128 units: 128 x 20 000 with 256 units
      // 350 -> 330 us per step): the same sweep, started behind an event of this stream recorded here.  (scVI's three head tensors: 276.4 against
      // 277.4 us with it -- its riders stay.)  The tensor's norm is then summed from the products' slots directly instead of from the reduce
      // riders' partial sums: where the clip bites, the last bits of the update differ between the two forms.
      SMX_CHECK(head_sweep_prepare(m));
      SMX_HIP(hipEventRecord(m->ev_hf, m->st));
      SMX_CHECK(head_sweep_start(m));
    }
    m->ev_hf_fresh = false;
    m->head_fused_bwd_done = false;
    if (dp_chain_ok(m)) {   // head gradients are final (label heads whose weight gradient rides with the last launch of the pass: optimizer_pass)
      if (!m->lab_deferred) SMX_CHECK(dp_chain_start(m));
    } else if (dp_overlap(m)) {  // (the hand-written exchange) head gradients are final: reduce them while the rest of backward runs
      SMX_HIP(hipEventRecord(m->ev_c1, m->st));
      SMX_HIP(hipStreamWaitEvent(m->st_comm, m->ev_c1, 0));
      SMX_CHECK(dp_allreduce(m, m->bucket1_off, m->bucket1_count, m->st_comm));
      m->bucket1_in_flight = true;
    }
  }
  // ---- decoder MLP; the latent head's backward runs in the epilogue of the d z product ----
  const int lat_ld = m->lat_planes * m->Dp;
  EpiLatentBwd le;
  le.lat = m->latbuf; le.ld = lat_ld; le.sig = m->sig; le.eps = m->eps; le.kl_scale = c.beta * inv_gb;
  le.D = m->D; le.Dp = m->Dp; le.stochastic = m->stochastic; le.relu = (c.latent_activation == SMX_ACT_RELU);
  le.dlat = m->dlat;
  if (m->fvae) le.dz_add = m->dz_tc;
  if (m->scale) {
    le.dklz = m->dklz;
    ScalePriorArgs sp;
    sp.z = m->z; sp.B = ps.B; sp.D = m->D; sp.Dp = m->Dp; sp.C = c.n_components;
    sp.logits = P_(m, m->t_prLogits); sp.loc = P_(m, m->t_prLoc); sp.scale_raw = P_(m, m->t_prScale);
    sp.resp = m->resp; sp.kl_scale = c.beta * inv_gb;
    sp.g_logits = G_(m, m->t_prLogits); sp.g_loc = G_(m, m->t_prLoc); sp.g_scale = G_(m, m->t_prScale);
    sp.tie_mixtures = m->flags.tie_mixtures; sp.tie_loc = m->flags.tie_loc; sp.tie_scale = m->flags.tie_scale; sp.tril = m->scale_tril;
    sp.tril_part = m->tril_part; sp.tril_part_floats = m->tril_part_floats;
    SMX_CHECK(launch_scale_prior_bwd(m->st, sp));
  }
  // Products that only the optimiser reads (the weight gradients of the first decoder layer, of the latent head and of
  // the first encoder layers) run as ONE grouped launch at the end; the last encoder layer's BatchNorm-backward
  // launch computes d h = d lat W_lat^T itself.  SMX_NO_BWD_FRONT=1: the separate launches of before.
  const MlpLayer& eL = m->enc.back();
  const bool bfront = m->flags.bwd_front && !sync_bn_on(m, ps.training) && bn_bwd_front_supported(ps.B, lat_ld) && eL.out_p % 8 == 0;
  // fold_dz (round 6): the d z product and the latent head's backward inside the encoder's last BatchNorm-backward launch -- the plain
  // reparameterised latent of VAE / SISUA at D <= 32, a first decoder layer of 128 units, at most 128 cells, below the wide-panel width,
  // no second MLP sharing the launch (scvi)
  // (a property of the MODEL, not of the step: at a wide panel the optimiser's chunks may ride with the d z launch -- how many is scheduling
  // state --, so wide panels keep that launch whatever rides with it this step; the fold's rounding never depends on what else is going on)
  const bool wide_panel = m->Gp >= std::min(4096, head_fused_min_genes());
  m->fold_dz_now = bfront && !m->mixpost && !m->scvi && !m->scale && !m->fvae && m->stochastic && !wide_panel && !m->dec.empty() &&
                   m->dec[0].out_p == 128 && m->dec[0].in_p == m->Dp && (m->tensors[m->dec[0].tW].ld % 4) == 0 && m->tensors[m->dec[0].tW].ld >= 128 &&
                   bn_bwd_fold_supported(ps.B, lat_ld, m->Dp);
  std::vector<GemmArgs> tail;
  if (m->mixpost) {
    // mixture-density posterior: d z leaves the decoder as slabs, a launch of its own turns it into d lat (every component's
    // parameters through log q, the picked component's through z as well)
    int dz_slabs = 1;
    SMX_CHECK(mlp_backward(m, m->dec, ps, m->z, m->Dp, false, n_slabs, false, &dz_slabs, "", nullptr, nullptr, nullptr, bfront ? &tail : nullptr));
    MixLatArgs ma;
    ma.lat = m->latbuf; ma.ld = lat_ld; ma.B = ps.B; ma.D = m->D; ma.Dp = m->Dp; ma.C = c.n_components;
    ma.z = m->z; ma.eps = m->eps; ma.resp = m->resp; ma.pick = m->zpick;
    ma.dz = m->slab; ma.dz_slabs = dz_slabs; ma.dz_slab_stride = (long)ps.B * m->dec[0].in_p; ma.ldz = m->dec[0].in_p;
    ma.kl_scale = c.beta * inv_gb; ma.dlat = m->dlat;
    Timed t(m, "latent_bwd");
    SMX_CHECK(launch_mixlat_bwd(m->st, ma));
  } else {
    SMX_CHECK(mlp_backward(m, m->dec, ps, m->z, m->Dp, false, n_slabs, false, nullptr, "", &le, nullptr, nullptr, bfront ? &tail : nullptr));
  }
  for (const GemmArgs& g : lab_dw) tail.push_back(g);
  BnBwdArgs gf;
  int dh_slabs = 1;   // slabs of d h the encoder's backward sums
  {  // weight gradient of the latent head and d h = d lat * W_lat^T
    const TensorInfo& tw = m->tensors[m->t_latW];
    GemmArgs pair[2];
    GemmArgs& g = pair[0];
    g.A = eL.out_buf; g.lda = eL.out_p; g.a_kmajor = 1; g.B = m->dlat; g.ldb = lat_ld;
    g.C = G_(m, m->t_latW); g.ldc = tw.ld; g.M = eL.out_p; g.N = lat_ld; g.K = ps.B; g.colsum = G_(m, m->t_latb);
    want_sq(m, g, m->t_latW);
    if (bfront) {
      tail.push_back(g);
      gf.fD = m->dlat; gf.fld = lat_ld; gf.fW = P_(m, m->t_latW); gf.fldw = tw.ld; gf.fK = lat_ld;
      if (m->fold_dz_now) {
        const MlpLayer& d0 = m->dec[0];
        gf.fold_dz = 1; gf.zD = d0.dpre; gf.zld = d0.out_p; gf.zW = P_(m, d0.tW); gf.zldw = (int)m->tensors[d0.tW].ld; gf.zlb = le;
      }
    } else {   // independent: one grouped launch
      GemmArgs& h = pair[1];
      h.A = m->dlat; h.lda = lat_ld; h.B = P_(m, m->t_latW); h.ldb = tw.ld; h.b_nmajor = 1;
      h.C = m->slab; h.ldc = eL.out_p; h.slab_stride = (long)ps.B * eL.out_p;
      h.M = ps.B; h.N = eL.out_p; h.K = lat_ld;
      int effs[2] = {1, 1};
      if (lat_ld > 128) {   // (a wide latent head -- the mixture-density posterior's (1 + 2 C) planes: d h contracts over all of them)
        h.split_k = suggest_split_k(ps.B, eL.out_p, lat_ld);
        SMX_REQUIRE((size_t)std::max(h.split_k, 1) * (size_t)h.slab_stride <= m->slab_cap, "split-K slabs exceed the slab buffer");
      }
      Timed t(m, "gemm_lat_bwd");
      SMX_CHECK(launch_gemm_group(m->st, pair, 2, effs));
      dh_slabs = effs[1];
    }
  }
  GemmArgs dw0[2];
  int n_dw0 = 0;
  // scvi: the library encoder's last BatchNorm-backward takes its incoming gradient d h_l = d latl W_latl^T as a front
  // too (K = 32) -- and then runs beside the encoder's in ONE launch when d latl is there already (the row-local head
  // launch of the forward pass leaves it); the library head's weight gradient joins the grouped launch at the end
  BnBwdArgs gfl;
  bool lfront = false, twin_done = false;
  if (m->scvi) {
    const MlpLayer& lL = m->encl.back();
    const TensorInfo& tw = m->tensors[m->t_latlW];
    lfront = bfront && bn_bwd_front_supported(ps.B, 32) && lL.out_p % 8 == 0 && (tw.ld % 4) == 0;
    gfl.fD = m->dlatl; gfl.fld = 32; gfl.fW = P_(m, m->t_latlW); gfl.fldw = tw.ld; gfl.fK = 32;
  }
  const bool twin_bwd = m->scvi && lfront && m->scvi_fused;
  SMX_CHECK(mlp_backward(m, m->enc, ps, ps.Xsrc, m->Gp, true, dh_slabs, true, nullptr, "gemm_enc_dw", nullptr, &dw0[n_dw0], bfront ? &gf : nullptr,
                         bfront ? &tail : nullptr, twin_bwd ? &m->encl : nullptr, twin_bwd ? &gfl : nullptr, &twin_done));
  ++n_dw0;
  // ---- scvi library branch ----
  if (m->scvi) {
    if (!m->scvi_fused) {
      LibLatentArgs ll;
      ll.latl = m->latlbuf; ll.ld = 32; ll.B = ps.B; ll.library = ps.lib; ll.rows = ps.rows;
      ll.sig = m->lsig; ll.eps = m->leps; ll.dl = m->dl; ll.kl_scale = c.beta * inv_gb; ll.dlatl = m->dlatl;
      SMX_CHECK(launch_lib_latent_bwd(m->st, ll));
    }
    const MlpLayer& lL = m->encl.back();
    const TensorInfo& tw = m->tensors[m->t_latlW];
    GemmArgs g;
    g.A = lL.out_buf; g.lda = lL.out_p; g.a_kmajor = 1; g.B = m->dlatl; g.ldb = 32;
    g.C = G_(m, m->t_latlW); g.ldc = tw.ld; g.M = lL.out_p; g.N = 32; g.K = ps.B; g.colsum = G_(m, m->t_latlb);
    want_sq(m, g, m->t_latlW);
    if (lfront) {
      tail.push_back(g);
    } else {
      GemmArgs h;
      h.A = m->dlatl; h.lda = 32; h.B = P_(m, m->t_latlW); h.ldb = tw.ld; h.b_nmajor = 1;
      h.C = m->slab; h.ldc = lL.out_p; h.slab_stride = (long)ps.B * lL.out_p;
      h.M = ps.B; h.N = lL.out_p; h.K = 32;
      GemmArgs pair[2] = {g, h};   // weight and input gradient of the library head: independent, one grouped launch
      SMX_CHECK(launch_gemm_group(m->st, pair, 2));
    }
    SMX_CHECK(mlp_backward(m, m->encl, ps, ps.Xsrc, m->Gp, true, 1, true, nullptr, "gemm_encl_dw", nullptr, &dw0[n_dw0],
                           lfront ? &gfl : nullptr, bfront ? &tail : nullptr, nullptr, nullptr, nullptr, twin_done));
    ++n_dw0;
  }
  // the first-layer weight gradients (gather + log1p of the same resident rows) of the encoder and, for scvi,
  // the library encoder are independent: one grouped launch
  for (int q = 0; q < n_dw0; ++q) tail.push_back(dw0[q]);
  {
    Timed t(m, "gemm_enc_dw");
    // every product here contracts over the minibatch: the wide direct-operand kernel takes them all in one launch
    // (SMX_NO_WGRAD=1, input dropout or an unsupported shape: the LDS-tiled products)
    bool wg_ok = m->flags.wgrad && tail.size() <= SMX_GROUP_MAX;
    for (const GemmArgs& g : tail) wg_ok = wg_ok && wgrad_supported(g, ps.B);
    const int b3 = m->flags.bf16x3 < 0 ? (use_bf16x3((long)ps.B * m->Gp * m->k) ? 1 : 0) : m->flags.bf16x3;
    if (wg_ok) SMX_CHECK(launch_wgrad_group(m->st, tail.data(), (int)tail.size(), ps.B, b3));
    else if (tail.size() == 1) SMX_CHECK(launch_gemm(m->st, tail[0]));
    else
      for (size_t q = 0; q < tail.size(); q += SMX_GROUP_MAX)
        SMX_CHECK(launch_gemm_group(m->st, tail.data() + q, (int)std::min<size_t>(SMX_GROUP_MAX, tail.size() - q)));
  }
  return SMX_OK;
}

// everything of AdamArgs that does not depend on which launch carries the chunks
void fill_adam_args(smx_model* m, AdamArgs& a) {
  const smx_config& c = m->cfg;
  a.params = m->params; a.grads = m->grads; a.m = m->adam_m; a.v = m->adam_v;
  a.chunks = m->chunks; a.n_chunks = m->n_chunks; a.n_launch = m->n_chunks; a.gap_from = m->n_chunks; a.gap_len = 0;
  a.partial = m->partial; a.tensor_norm = m->tensor_norm;
  // norms from the products' partials when every large tensor has them (single GPU: under data parallelism the
  // norm is that of the all-reduced gradient, which only a pass after the collective can give)
  a.use_sq = (m->sq_slots != nullptr && !dp_active(m) && !tuning_on("no_sq_partials")) ? 1 : 0;
  for (size_t t = 0; t < m->tensors.size() && a.use_sq; ++t) {
    a.sq_first[t] = m->sq_first[t]; a.sq_count[t] = m->sq_count[t];
    if (m->sq_count[t] == 0 && m->tensors[t].count > SMX_SQ_SMALL_TENSOR) a.use_sq = 0;
    if (m->sq_reduced[t]) { a.sq_first[t] = m->sq_total_first + (int)t * SMX_SQR_PER_TENSOR; a.sq_count[t] = m->sq_reduced[t]; }   // riders have summed the slots
  }
  a.sq_slots = m->sq_slots;
  a.state = cur_state(m); a.b1 = c.adam_beta1; a.b2 = c.adam_beta2; a.eps = c.adam_eps; a.clipnorm = c.clipnorm;
  // the likelihood / KL / label kernels already scale by 1 / (batch * world), so the SUM all-reduce leaves the
  // global-mean gradient: nothing more to divide by (it used to be divided by world once more here)
  a.grad_scale = 1.f;
  if (m->scale && (m->flags.tie_loc || m->flags.tie_scale)) {
    a.tied_t0 = m->flags.tie_loc ? m->t_prLoc : -1; a.tied_t1 = m->flags.tie_scale ? m->t_prScale : -1;
    a.tied_inv = 1.f / (float)c.n_components;
  }
}

int optimizer_pass(smx_model* m) {
  const smx_config& c = m->cfg;
  if (dp_active(m) && m->have_pending_metrics) {   // no BatchNorm-backward launch took them along
    SMX_CHECK(launch_metrics(m->st, m->pending_metrics));
    m->have_pending_metrics = false;
  }
  const bool chain = dp_chain_ok(m);
  if (chain) {
    Timed t(m, "allreduce");
    SMX_CHECK(dp_chain_start(m));   // (a no-op when the forward or the backward pass started it)
    SMX_CHECK(dp_allreduce(m, 0, m->bucket1_off, m->st));   // front bucket [encoder / latent / decoder | BatchNorm statistics | ELBO scalars]
  } else if (dp_active(m)) {
    Timed t(m, "allreduce");
    if (m->bucket1_in_flight) {
      // front bucket [encoder/latent/decoder grads | BN stats | metrics] behind the head bucket on the
      // communication stream; the optimiser waits for both
      SMX_HIP(hipEventRecord(m->ev_c2, m->st));
      SMX_HIP(hipStreamWaitEvent(m->st_comm, m->ev_c2, 0));
      SMX_CHECK(dp_allreduce(m, 0, m->bucket1_off, m->st_comm));
      SMX_HIP(hipEventRecord(m->ev_c3, m->st_comm));
      SMX_HIP(hipStreamWaitEvent(m->st, m->ev_c3, 0));
      m->bucket1_in_flight = false;
    } else {
      SMX_CHECK(dp_allreduce(m, 0, m->grads_count, m->st));   // one all-reduce of the whole flat buffer
    }
  }
  AdamArgs a;
  fill_adam_args(m, a);
  if (chain) a.sq_chunks = m->chunk_first_head;   // (the heads' chunks have their norm pass on the communication stream)
  m->chain_started = false;
  if (dp_active(m) && m->bn_total && m->world > 1) {
    if (!a.use_sq) {   // (the usual case under data parallelism: the gradient-norm launch takes the update along)
      a.bn_moving = m->bn_moving; a.bn_batch = m->grads + m->tail_off_bn; a.bn_total = (int)m->bn_total;
      a.bn_inv_world = 1.f / (float)m->world; a.bn_momentum = c.bn_momentum;
    } else {
      hipLaunchKernelGGL(bn_moving_update_kernel, dim3((unsigned)((m->bn_total + 255) / 256)), dim3(256), 0, m->st,
                         m->bn_moving, m->grads + m->tail_off_bn, (int)m->bn_total, 1.f / (float)m->world,
                         c.bn_momentum);
    }
  }
  if (m->adam_early_from >= 0) {   // the head chunks have ridden along already
    a.gap_from = m->adam_early_from; a.gap_len = m->adam_early_to - m->adam_early_from;
    a.n_launch = m->n_chunks - a.gap_len;
  } else {
    a.gap_from = m->n_chunks; a.gap_len = 0; a.n_launch = m->n_chunks;
  }
  m->adam_early_from = -1;
  if (m->have_pending_metrics) { a.metrics = m->pending_metrics; a.with_metrics = 1; m->have_pending_metrics = false; }
  a.master = master_state(m); a.lr = c.lr; a.batch = m->seq_batch;
  if (dp_active(m)) { a.hist_dp = m->mhist; a.tail_metrics = m->grads + m->tail_off_metrics; }
  a.prepare_next = m->seq_prepare_next;
  if (a.prepare_next) { a.next_state = m->state3 + (m->par ^ 1); a.next_rows = m->rows2[m->par ^ 1]; a.order = m->order; }
  Timed t(m, "adam");
  SMX_CHECK(launch_adam(m->st, a));
  return SMX_OK;
}

// sparse store: expand the rows of this pass into the dense tile the readers of X take (they then index it with
// identity rows; everything else -- labels, library prior, label mask, lgx1, noise keys -- keeps the resident row ids)
int csr_stage(smx_model* m, Pass& ps) {
  if (!m->x_csr || ps.Xsrc != m->X) return SMX_OK;
  SMX_REQUIRE(ps.rows != nullptr && ps.B <= m->Bmax, "sparse store: resident rows only");
  SMX_REQUIRE(!(ps.training && m->cfg.input_dropout > 0.f), "sparse store: input dropout is keyed by the dense store's rows (use the float32 / uint16 store)");
  SMX_CHECK(launch_csr_expand(m->st, m->csr_indptr, m->csr_cols, m->csr_vals, ps.rows, 0, ps.B, m->Gp, m->xbatch));
  ps.Xsrc = m->xbatch; ps.xrows = nullptr; ps.x_u16 = 0;
  return SMX_OK;
}

// the whole training step as a launch sequence on m->st (capturable).
//   with_begin:   launch the state/row-id preparation kernel first (graph replay: every step, cursor kept in
//                 the master state; eager: only the first step of a train_steps call)
//   prepare_next: the optimiser kernel prepares the other parity's state + row ids for the step after
int train_sequence(smx_model* m, int B, bool with_begin, bool begin_from_master, uint32_t cursor, bool prepare_next) {
  Pass ps;
  ps.B = B; ps.rows = cur_rows(m); ps.xrows = ps.rows; ps.Xsrc = m->X; ps.x_u16 = m->x_u16; ps.lib = m->library; ps.lgx1 = m->lgx1;
  ps.cell_base = (uint32_t)m->cell_base; ps.training = 1; ps.sample = 0; ps.global_batch = B * m->world;
  m->seq_batch = B; m->seq_prepare_next = prepare_next ? 1 : 0;
  Timed t(m, "step");
  if (with_begin)
    SMX_CHECK(launch_step_begin(m->st, master_state(m), cur_state(m), m->order, cur_rows(m), B, begin_from_master ? 1 : 0,
                                cursor, m->cfg.lr, m->cfg.adam_beta1, m->cfg.adam_beta2));
  m->chain_started = false;
  { Timed null_pair(m, "null"); }  // an event pair around nothing: the timing method's own overhead
  SMX_CHECK(csr_stage(m, ps));     // sparse store: this minibatch's rows as a dense tile (no-op otherwise)
  SMX_CHECK(forward_pass(m, ps, true, true));
  SMX_CHECK(backward_pass(m, ps));
  SMX_CHECK(optimizer_pass(m));
  return SMX_OK;
}

int read_metrics(smx_model* m, smx_metrics* out) {
  if (!out) return SMX_OK;
  // into pinned memory (two real DMA copies; into pageable arrays the runtime stages each one synchronously)
  const size_t nt = m->tensors.size();
  if (!m->metrics_pin) SMX_HIP(hipHostMalloc((void**)&m->metrics_pin, (8 + nt + 1) * sizeof(float), hipHostMallocDefault));
  float* h = m->metrics_pin;
  float* norms_p = m->metrics_pin + 8;
  unsigned* p2p_err = reinterpret_cast<unsigned*>(m->metrics_pin + 8 + nt);
  *p2p_err = 0u;
  SMX_HIP(hipMemcpyAsync(h, m->grads + m->tail_off_metrics, 8 * sizeof(float), hipMemcpyDeviceToHost, m->st));
  SMX_HIP(hipMemcpyAsync(norms_p, m->tensor_norm, nt * sizeof(float), hipMemcpyDeviceToHost, m->st));
  if (m->p2p && m->p2p->error) SMX_HIP(hipMemcpyAsync(p2p_err, m->p2p->error, sizeof(unsigned), hipMemcpyDeviceToHost, m->st));
  SMX_HIP(hipStreamSynchronize(m->st));
  if (*p2p_err) {   // a bounded wait of the peer-to-peer exchange gave up on this rank or on a peer: every rank's step is garbage
    set_error(*p2p_err == 1u ? "peer-to-peer all-reduce: a wait on a peer's flag timed out (SMX_P2P_TIMEOUT_S)" :
                               "peer-to-peer all-reduce: a peer reported a timed-out wait");
    return SMX_ERR_COMM;
  }
  std::vector<float> norms(norms_p, norms_p + nt);
  out->loss = h[0]; out->nllk_x = h[1]; out->nllk_y = h[2]; out->kl = h[3]; out->kl_l = h[4];
  float mx = 0.f;
  for (float v : norms) mx = (v > mx || v != v) ? v : mx;
  out->grad_norm_max = mx;
  out->nan_flag = !(isfinite(h[0]) && isfinite(h[1]) && isfinite(h[3]) && isfinite(mx));
  out->step = (int32_t)m->h_next;
  out->tc = h[5]; out->dtc_loss = h[6]; out->nllk_o = h[7];
  if (m->n_observed && !isfinite(h[7])) out->nan_flag = 1;
  if (m->fvae && !(isfinite(h[5]) && isfinite(h[6]))) out->nan_flag = 1;
  return SMX_OK;
}

int upload_order(smx_model* m, const int32_t* order, size_t n, size_t n_steps) {
  if (n_steps > m->mhist_cap) {
    SMX_HIP(hipStreamSynchronize(m->st));
    drop_graphs(m);
    if (m->mhist) hipFree(m->mhist);
    m->mhist = nullptr; m->mhist_cap = 0;
    SMX_CHECK(dmalloc(&m->mhist, (n_steps * 2 + 64) * 8));
    m->mhist_cap = n_steps * 2 + 64;
  }
  m->mhist_steps = (int32_t)n_steps;
  if (n > m->order_cap) {
    SMX_HIP(hipStreamSynchronize(m->st));
    drop_graphs(m);
    if (m->order) hipFree(m->order);
    m->order = nullptr;
    m->order_cap = n * 2 + (size_t)m->Bmax;
    SMX_CHECK(dmalloc(&m->order, m->order_cap));
  }
  // through a pinned buffer of the model's own: the copy is then a real asynchronous DMA (from pageable memory the runtime
  // stages it synchronously: ~80 us for 10 KB, 4 us per step of a 20-step call).  The buffer is reused only after the
  // previous call's copy has run (an event; by then normally long done)
  if (n > m->order_pin_cap) {
    if (m->order_pin_busy) { SMX_HIP(hipEventSynchronize(m->ev_order)); m->order_pin_busy = false; }
    if (m->order_pin) hipHostFree(m->order_pin);
    m->order_pin = nullptr; m->order_pin_cap = 0;
    SMX_HIP(hipHostMalloc((void**)&m->order_pin, (n * 2 + (size_t)m->Bmax) * sizeof(int32_t), hipHostMallocDefault));
    m->order_pin_cap = n * 2 + (size_t)m->Bmax;
  }
  if (!m->ev_order) SMX_HIP(hipEventCreateWithFlags(&m->ev_order, hipEventDisableTiming));
  if (m->order_pin_busy) { SMX_HIP(hipEventSynchronize(m->ev_order)); m->order_pin_busy = false; }
  memcpy(m->order_pin, order, n * sizeof(int32_t));
  SMX_HIP(hipMemcpyAsync(m->order, m->order_pin, n * sizeof(int32_t), hipMemcpyHostToDevice, m->st));
  SMX_HIP(hipEventRecord(m->ev_order, m->st));
  m->order_pin_busy = true;
  SMX_HIP(hipMemsetAsync(&master_state(m)->cursor, 0, sizeof(uint32_t), m->st));
  return SMX_OK;
}

int check_rows(smx_model* m, const int32_t* ids, size_t n) {
  SMX_REQUIRE(m->X != nullptr, "no dataset uploaded (smx_dataset_upload)");
  for (size_t i = 0; i < n; ++i)
    if (ids[i] < 0 || (int64_t)ids[i] >= m->N) { set_error("row id out of range"); return SMX_ERR_INVALID; }
  return SMX_OK;
}

int launch_train(smx_model* m, int B, bool use_graph, int s_idx, int n_steps) {
  // With a communicator the RCCL all-reduce is captured too (RCCL supports stream capture);
  // SMX_NO_GRAPH_COMM=1 or a failed capture falls back to eager launches for good.
  static const bool no_graph_comm = tuning_on("no_graph_comm");
  // (the peer-to-peer exchange carries its epoch as a kernel argument: never captured)
  if (use_graph && !m->local && !m->p2p && !(m->comm && (no_graph_comm || m->graph_comm_failed)) && !m->use_injected && m->timing_label.empty()) {
    auto it = m->graphs.find(B);
    if (it == m->graphs.end()) {
      hipGraph_t graph = nullptr;
      SMX_HIP(hipStreamBeginCapture(m->st, hipStreamCaptureModeThreadLocal));
      m->capturing = true;
      m->par = 0;
      int rc = train_sequence(m, B, true, true, 0, false);
      m->capturing = false;
      hipError_t e = hipStreamEndCapture(m->st, &graph);
      hipGraphExec_t exec = nullptr;
      if (rc == SMX_OK && e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      if (graph) hipGraphDestroy(graph);
      if (rc != SMX_OK || e != hipSuccess) {
        (void)hipGetLastError();
        if (m->comm) {  // capture with the collective failed: run this and all later steps eagerly
          m->graph_comm_failed = true;
          SMX_CHECK(train_sequence(m, B, true, true, 0, false));
          m->h_next += 1;
          return SMX_OK;
        }
        if (rc != SMX_OK) return rc;
        set_error(std::string("graph capture failed: ") + hipGetErrorString(e));
        return SMX_ERR_HIP;
      }
      it = m->graphs.emplace(B, exec).first;
    }
    m->par = 0;
    SMX_HIP(hipGraphLaunch(it->second, m->st));
  } else {
    // eager: the preparation kernel runs once per call; afterwards each optimiser kernel prepares the
    // other parity's state + row ids, so a step is not fronted by a 1-workgroup latency kernel
    const bool first = (s_idx == 0), last = (s_idx == n_steps - 1);
    if (first) m->par = 0; else m->par ^= 1;
    SMX_CHECK(train_sequence(m, B, first, false, (uint32_t)s_idx, !last));
  }
  m->h_next += 1;
  return SMX_OK;
}

int setup_pass(smx_model* m, Pass& ps, const int32_t* row_ids, const float* host_x, const float* host_library,
                      int32_t batch, int training, int sample) {
  SMX_REQUIRE(batch > 0 && batch <= m->Bmax, "batch must be in 1..max_batch");
  ps.B = batch; ps.training = training; ps.sample = sample; ps.global_batch = batch;
  if (row_ids) {
    SMX_CHECK(check_rows(m, row_ids, (size_t)batch));
    // (through pinned staging of the library's own: measured, no gain -- 512 bytes from a pageable array take the runtime's fast path)
    SMX_HIP(hipMemcpyAsync(cur_rows(m), row_ids, (size_t)batch * sizeof(int32_t), hipMemcpyHostToDevice, m->st));
    ps.rows = cur_rows(m); ps.xrows = ps.rows; ps.Xsrc = m->X; ps.x_u16 = m->x_u16; ps.lib = m->library; ps.lgx1 = m->lgx1; ps.cell_base = (uint32_t)m->cell_base;
    SMX_CHECK(csr_stage(m, ps));
  } else {
    SMX_REQUIRE(host_x, "need row_ids or host_x");
    SMX_REQUIRE(!m->scvi || host_library, "scvi needs host_library with host_x");
    SMX_HIP(hipMemsetAsync(m->hostX, 0, (size_t)batch * m->Gp * sizeof(float), m->st));
    SMX_HIP(hipMemcpy2DAsync(m->hostX, (size_t)m->Gp * sizeof(float), host_x, (size_t)m->G * sizeof(float),
                             (size_t)m->G * sizeof(float), (size_t)batch, hipMemcpyHostToDevice, m->st));
    SMX_CHECK(launch_row_stats(m->st, m->hostX, 0, m->Gp, batch, m->G, m->hostLgx1, nullptr));
    if (host_library) SMX_HIP(hipMemcpy(m->hostLib, host_library, (size_t)batch * 2 * sizeof(float), hipMemcpyHostToDevice));
    ps.rows = nullptr; ps.Xsrc = m->hostX; ps.lib = m->hostLib; ps.lgx1 = m->hostLgx1; ps.cell_base = 0;
  }
  return SMX_OK;
}

}  // namespace smx

extern "C" {

int smx_train_step(smx_model* m, const int32_t* row_ids, int32_t batch, smx_metrics* out) {
  return smx_train_steps(m, row_ids, 1, batch, 0, out);
}
int smx_train_step_graph(smx_model* m, const int32_t* row_ids, int32_t batch, smx_metrics* out) {
  return smx_train_steps(m, row_ids, 1, batch, 1, out);
}

int smx_train_steps(smx_model* m, const int32_t* order, int32_t n_steps, int32_t batch, int use_graph, smx_metrics* out) {
  SMX_REQUIRE(m && n_steps > 0, "bad arguments");
  SMX_REQUIRE(batch > 0 && batch <= m->Bmax, "batch must be in 1..max_batch");
  if (order) {
    SMX_CHECK(check_rows(m, order, (size_t)n_steps * batch));
    SMX_CHECK(upload_order(m, order, (size_t)n_steps * batch, (size_t)n_steps));
  } else {
    SMX_REQUIRE(m->staged_steps == n_steps && m->staged_batch == batch, "order = NULL: no ids staged for this n_steps x batch (smx_train_stage)");
  }
  m->staged_steps = 0;   // (staged ids serve one call)
  // flag opt_shard: the heads' Adam moments outside this rank's slice are stale.  Steps that will NOT take the sharded chain (a captured graph,
  // the flag switched off since) update every element from its moments: gather them first (a collective -- every rank takes the same branch:
  // the flag, the communicator and use_graph are the job's, not a rank's)
  if (m->opt_stale && (use_graph || !(m->flags.opt_shard && smx::dp_chain_ok(m) && smx::dp_shard_available(m)))) SMX_CHECK(smx_opt_gather(m));
  int rc = SMX_OK;
  ++m->params_epoch;   // (the parameters are about to change)
  for (int s = 0; s < n_steps && rc == SMX_OK; ++s) rc = launch_train(m, batch, use_graph != 0, s, n_steps);
  { const int rj = smx::head_sweep_join(m); if (rc == SMX_OK) rc = rj; }   // every other entry point sees one stream
  if (rc != SMX_OK) return rc;
  if (m->use_injected) { m->use_injected = false; }
  // a non-finite loss / gradient norm is REPORTED (out->nan_flag), not an error of the call: terminate_on_nan
  // (configs/base.yaml:59) is the caller's decision
  SMX_CHECK(read_metrics(m, out));
  return SMX_OK;
}

int smx_train_stage(smx_model* m, const int32_t* order, int32_t n_steps, int32_t batch) {
  SMX_REQUIRE(m && order && n_steps > 0, "bad arguments");
  SMX_REQUIRE(batch > 0 && batch <= m->Bmax, "batch must be in 1..max_batch");
  SMX_CHECK(check_rows(m, order, (size_t)n_steps * batch));
  SMX_CHECK(upload_order(m, order, (size_t)n_steps * batch, (size_t)n_steps));
  m->staged_steps = n_steps; m->staged_batch = batch;
  return SMX_OK;
}

int smx_metrics_history(smx_model* m, int32_t n_steps, float* host) {
  SMX_REQUIRE(m && host && n_steps > 0 && n_steps <= m->mhist_steps, "no such history (steps of the last smx_train_steps call)");
  SMX_HIP(hipStreamSynchronize(m->st));
  SMX_HIP(hipMemcpy(host, m->mhist, (size_t)n_steps * 8 * sizeof(float), hipMemcpyDeviceToHost));
  return SMX_OK;
}

int smx_eval_step(smx_model* m, const int32_t* row_ids, int32_t batch, smx_metrics* out) {
  SMX_REQUIRE(m && row_ids, "bad arguments");
  Pass ps;
  SMX_CHECK(setup_pass(m, ps, row_ids, nullptr, nullptr, batch, 0, 0));
  SMX_CHECK(forward_pass(m, ps, true, false));
  SMX_CHECK(read_metrics(m, out));
  return SMX_OK;
}

}  // extern "C"
