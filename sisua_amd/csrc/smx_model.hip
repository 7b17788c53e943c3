#include <atomic>
// smx_model.hip -- model state: construction / destruction, tensors, BatchNorm statistics, noise injection, flags, timing.
#include "smx_model.h"

namespace smx {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* last_error_cstr() { return g_err.c_str(); }

int add_tensor(smx_model* m, const std::string& name, int rows, int cols, int chunks, bool vec) {
  TensorInfo t;
  t.name = name; t.rows = vec ? 1 : rows; t.cols = cols; t.chunks = chunks;
  t.chunk_w = cols / chunks; t.chunk_wp = round_up(t.chunk_w, 32);
  t.rows_p = vec ? 1 : round_up(rows, 32);
  t.ld = chunks * t.chunk_wp;
  t.offset = m->flat_count;
  t.count = (size_t)t.rows_p * t.ld;
  m->flat_count += (t.count + 63) / 64 * 64;
  m->tensors.push_back(t);
  return (int)m->tensors.size() - 1;
}

int build_mlp(smx_model* m, std::vector<MlpLayer>& mlp, const char* prefix, int n_in, int n, const int32_t* units,
              int stream0, float drop_p, bool batchnorm, float leak) {
  for (int i = 0; i < n; ++i) {
    MlpLayer L;
    L.in = n_in; L.in_p = round_up(n_in, 32); L.out = units[i]; L.out_p = round_up(units[i], 32);
    std::string p = std::string(prefix) + std::to_string(i);
    L.tW = add_tensor(m, p + "/W", n_in, units[i], 1, false);
    if (batchnorm) {
      L.tGamma = add_tensor(m, p + "/gamma", 1, units[i], 1, true);
      L.tBeta = add_tensor(m, p + "/beta", 1, units[i], 1, true);
      L.bn = (int)m->bn_w.size();
      m->bn_w.push_back(units[i]); m->bn_wp.push_back(L.out_p);
    } else {
      L.tBias = add_tensor(m, p + "/b", 1, units[i], 1, true);
    }
    L.stream = stream0 + i; L.drop_p = drop_p; L.leak = leak;
    mlp.push_back(L);
    n_in = units[i];
  }
  return n_in;
}

// the sparse store's arrays (m->X aliases the expansion tile while it is in use)
void release_csr(smx_model* m) {
  if (!m->x_csr) return;
  if (m->csr_indptr) hipFree(m->csr_indptr);
  if (m->csr_cols) hipFree(m->csr_cols);
  if (m->csr_vals) hipFree(m->csr_vals);
  if (m->xbatch) hipFree(m->xbatch);
  m->csr_indptr = nullptr; m->csr_cols = nullptr; m->csr_vals = nullptr; m->xbatch = nullptr;
  m->X = nullptr; m->x_csr = false;
}

NoiseKey make_key(smx_model* m, int stream, int sample, bool training) {
  NoiseKey nk;
  nk.k0 = (uint32_t)(m->cfg.seed & 0xFFFFFFFFu);
  nk.k1 = (uint32_t)(m->cfg.seed >> 32);
  nk.step = 0;
  nk.stream = (uint32_t)((stream & 0xFF) | ((sample & 0xFFFFFF) << 8));
  nk.step_ptr = training ? &cur_state(m)->step : nullptr;
  return nk;
}

const Injected* inj(smx_model* m, int stream) {
  if (!m->use_injected) return nullptr;
  auto it = m->injected.find(stream);
  return it == m->injected.end() ? nullptr : &it->second;
}

void drop_graphs(smx_model* m) {  // captured graphs bake device pointers in: drop them when a buffer moves
  for (auto& kv : m->graphs) hipGraphExecDestroy(kv.second);
  m->graphs.clear();
}

// pack logical host tensor <-> padded internal layout
void pack(const TensorInfo& t, const float* host, std::vector<float>& dev) {
  dev.assign(t.count, 0.f);
  for (int r = 0; r < t.rows; ++r)
    for (int ch = 0; ch < t.chunks; ++ch)
      memcpy(&dev[(size_t)r * t.ld + (size_t)ch * t.chunk_wp], host + (size_t)r * t.cols + (size_t)ch * t.chunk_w,
             sizeof(float) * t.chunk_w);
}
void unpack(const TensorInfo& t, const std::vector<float>& dev, float* host, float scale) {
  for (int r = 0; r < t.rows; ++r)
    for (int ch = 0; ch < t.chunks; ++ch)
      for (int j = 0; j < t.chunk_w; ++j)
        host[(size_t)r * t.cols + (size_t)ch * t.chunk_w + j] = dev[(size_t)r * t.ld + (size_t)ch * t.chunk_wp + j] * scale;
}

}  // namespace smx

// ---- developer knobs (smx_internal.h: tuning) -----------------------------------------------------------------------------
namespace smx {
static std::mutex g_tuning_mu;
static std::map<std::string, double>& tuning_map() {
  static std::map<std::string, double> mp;
  static bool parsed = false;
  if (!parsed) {   // SMX_TUNING="name=value,name=value" (a bare name means 1)
    parsed = true;
    if (const char* e = getenv("SMX_TUNING")) {
      std::string str(e);
      size_t i = 0;
      while (i < str.size()) {
        size_t j = str.find(',', i);
        if (j == std::string::npos) j = str.size();
        const std::string item = str.substr(i, j - i);
        const size_t eq = item.find('=');
        if (!item.empty()) mp[item.substr(0, eq)] = eq == std::string::npos ? 1.0 : atof(item.c_str() + eq + 1);
        i = j + 1;
      }
    }
  }
  return mp;
}
// moves on whenever a knob is set or cleared: anything cached under the knobs in force (the scoring head's W images: ADVICE r05) is keyed by it
std::atomic<unsigned long long> g_tuning_epoch{1};
unsigned long long tuning_epoch() { return g_tuning_epoch.load(); }
double tuning(const char* name, double dflt) {
  std::lock_guard<std::mutex> lk(g_tuning_mu);
  auto& mp = tuning_map();
  auto it = mp.find(name);
  return it == mp.end() ? dflt : it->second;
}
}  // namespace smx

extern "C" {

const char* smx_last_error(void) { return last_error_cstr(); }
int smx_abi_version(void) { return SMX_ABI_VERSION; }

// Host-side helper (no device work): the visit order of one epoch under a streaming shuffle buffer, the sequential part
// of sisua_amd/data.py::epoch_order (tf.data's .shuffle(buffer) semantics, _single_cell_base.py:597-600).  picks[t] are
// the caller's random integers (NumPy RandomState stream: the order is defined there); 2.3 ms per 3381-cell epoch in
// Python -- as long as the device needs for the epoch itself -- against ~10 us here.
int smx_shuffle_order(int32_t n_obs, int32_t buffer, const int64_t* picks, int32_t* out) {
  SMX_REQUIRE(n_obs >= 0 && buffer > 0 && (n_obs == 0 || (picks && out)), "bad arguments");
  std::vector<int32_t> buf((size_t)std::min(buffer, n_obs));
  for (size_t i = 0; i < buf.size(); ++i) buf[i] = (int32_t)i;
  int32_t nxt = (int32_t)buf.size();
  size_t len = buf.size();
  for (int32_t t = 0; t < n_obs; ++t) {
    SMX_REQUIRE(picks[t] >= 0 && len > 0, "negative pick");
    const size_t k = (size_t)(picks[t] % (int64_t)len);
    out[t] = buf[k];
    if (nxt < n_obs) buf[k] = nxt++;
    else { buf[k] = buf[len - 1]; --len; }
  }
  return SMX_OK;
}

int smx_set_tuning(const char* name, double value) {
  SMX_REQUIRE(name && *name, "null name");
  std::lock_guard<std::mutex> lk(smx::g_tuning_mu);
  smx::tuning_map()[name] = value;
  ++smx::g_tuning_epoch;
  return SMX_OK;
}
int smx_clear_tuning(const char* name) {
  std::lock_guard<std::mutex> lk(smx::g_tuning_mu);
  if (name && *name) smx::tuning_map().erase(name); else smx::tuning_map().clear();
  ++smx::g_tuning_epoch;
  return SMX_OK;
}

int smx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int smx_init(int device) {
  int n = smx_device_count();
  if (device < 0 || device >= n) {
    set_error("smx_init: no such HIP device (" + std::to_string(device) + " of " + std::to_string(n) + ")");
    return SMX_ERR_INVALID;
  }
  SMX_HIP(hipSetDevice(device));
  return SMX_OK;
}

int smx_synchronize(void) {
  SMX_HIP(hipDeviceSynchronize());
  return SMX_OK;
}

int smx_model_create(const smx_config* cfg, smx_model** out) {
  SMX_REQUIRE(cfg && out, "null argument");
  SMX_REQUIRE(cfg->abi_version == SMX_ABI_VERSION, "smx_config.abi_version mismatch");
  SMX_REQUIRE(cfg->n_genes > 0 && cfg->latent_dim > 0 && cfg->max_batch > 0, "n_genes, latent_dim, max_batch must be > 0");
  SMX_REQUIRE(cfg->n_enc >= 1 && cfg->n_enc <= SMX_MAX_LAYERS && cfg->n_dec >= 1 && cfg->n_dec <= SMX_MAX_LAYERS,
              "encoder/decoder need 1..8 layers");
  SMX_REQUIRE(cfg->model >= SMX_MODEL_VAE && cfg->model <= SMX_MODEL_SCALE_POST, "unknown model kind");
  if (cfg->model == SMX_MODEL_FVAE) {
    SMX_REQUIRE(cfg->disc_layers >= 1 && cfg->disc_layers <= SMX_MAX_LAYERS && cfg->disc_units >= 1, "fvae: discriminator needs 1..8 hidden layers");
    SMX_REQUIRE(cfg->disc_leak >= 0.f && cfg->disc_leak < 1.f, "fvae: leaky-ReLU slope in [0, 1)");
  }
  if (cfg->model == SMX_MODEL_SCALE || cfg->model == SMX_MODEL_SCALE_TRIL) SMX_REQUIRE(cfg->n_components >= 2 && cfg->n_components <= 32, "scale: 2..32 mixture components");
  if (cfg->model == SMX_MODEL_SCALE_TRIL) SMX_REQUIRE(cfg->latent_dim <= 32, "scale with full-covariance components: at most 32 latent dimensions");
  if (cfg->model == SMX_MODEL_SCALE_POST) SMX_REQUIRE(cfg->n_components >= 2 && cfg->n_components <= 8 && cfg->n_components <= cfg->latent_dim && cfg->latent_dim <= 64, "scale with a mixture-density posterior: 2 .. min(latent_dim, 8) components, at most 64 latent dimensions");
  SMX_REQUIRE(cfg->likelihood >= SMX_LLK_NB && cfg->likelihood <= SMX_LLK_MSE, "unknown likelihood");
  SMX_REQUIRE(cfg->n_labels >= 0 && cfg->n_labels <= SMX_MAX_LABELS, "too many label heads");
  // (SCALE with label heads = SCALAR, sisua/models/scale.py:52-59: the mixture prior of SCALE under SISUA's semi-supervised heads)
  int n_observed = 0;
  for (int j = 0; j < cfg->n_labels; ++j) {
    if (cfg->label_observed[j]) { SMX_REQUIRE(j == n_observed, "observed output heads come before the label heads"); ++n_observed; }
  }
  // (outputs[1:], tests/test_singlecell_models.py:129-141 / scvi.py:168-169: observed heads on any model's decoder output -- FactorVAE
  // (fvae.py:9-18 passes `outputs` through unchanged) and the mixture-density posterior included since round 5)
  if (cfg->model == SMX_MODEL_FVAE) {   // SemiFVAE's label variables (behind the observed outputs) are classified by the discriminator: no heads
    SMX_REQUIRE(cfg->n_labels - n_observed <= SMX_DISC_MAX_GROUPS, "fvae: at most 8 label variables");
    int classes = 0;
    for (int j = n_observed; j < cfg->n_labels; ++j) {
      SMX_REQUIRE(cfg->label_llk[j] == SMX_LABEL_ONEHOT && cfg->label_dim[j] >= 2, "fvae: label variables are one-hot with at least 2 classes");
      classes += cfg->label_dim[j];
    }
    SMX_REQUIRE(classes <= 32, "fvae: the label variables have at most 32 classes in all (the discriminator's logit layer)");
  }
  if (cfg->model == SMX_MODEL_SCALE_POST) SMX_REQUIRE(cfg->n_labels == n_observed, "scale with a mixture-density posterior: no label heads (observed outputs only)");
  SMX_REQUIRE(cfg->model == SMX_MODEL_SISUA || cfg->model == SMX_MODEL_FVAE || cfg->model == SMX_MODEL_SCALE || cfg->model == SMX_MODEL_SCALE_TRIL || cfg->n_labels == n_observed,
              "label heads need model = SISUA, SCALE (SCALAR) or FVAE (SemiFVAE)");
  SMX_REQUIRE(cfg->scvi_dispersion >= 0 && cfg->scvi_dispersion <= 2 && cfg->scvi_inflation >= 0 && cfg->scvi_inflation <= 2, "scvi_dispersion / scvi_inflation: 0 ('full'), 1 ('share') or 2 ('single')");
  SMX_REQUIRE(cfg->model == SMX_MODEL_SCVI || (cfg->scvi_dispersion == 0 && cfg->scvi_inflation == 0), "dispersion / inflation are options of scvi");
  if (cfg->model == SMX_MODEL_SCVI) {
    SMX_REQUIRE(cfg->likelihood == SMX_LLK_NBD || cfg->likelihood == SMX_LLK_ZINBD, "scvi supports nbd / zinbd only");
    SMX_REQUIRE(cfg->n_encl >= 1 && cfg->n_encl <= SMX_MAX_LAYERS, "scvi needs a library encoder");
  }
  SMX_REQUIRE(cfg->dropout_enc >= 0 && cfg->dropout_enc < 1 && cfg->dropout_dec >= 0 && cfg->dropout_dec < 1 &&
                  cfg->input_dropout >= 0 && cfg->input_dropout < 1, "dropout rates must be in [0,1)");
  int dev = 0;
  SMX_HIP(hipGetDevice(&dev));
  smx_model* m = new smx_model();
  m->cfg = *cfg; m->device = dev;
  m->G = cfg->n_genes; m->Gp = round_up(m->G, 32); m->D = cfg->latent_dim; m->Dp = round_up(m->D, 32);
  m->k = llk_planes(cfg->likelihood);
  m->stochastic = cfg->model != SMX_MODEL_DCA; m->scvi = cfg->model == SMX_MODEL_SCVI; m->scale = cfg->model == SMX_MODEL_SCALE || cfg->model == SMX_MODEL_SCALE_TRIL; m->scale_tril = cfg->model == SMX_MODEL_SCALE_TRIL;
  m->mixpost = cfg->model == SMX_MODEL_SCALE_POST;
  m->lat_planes = m->mixpost ? 1 + 2 * cfg->n_components : (m->stochastic ? 2 : 1);
  m->fvae = cfg->model == SMX_MODEL_FVAE; m->n_heads = m->fvae ? n_observed : cfg->n_labels; m->n_observed = n_observed;
  m->out_has_W[1] = cfg->scvi_dispersion == 0; m->out_has_W[2] = cfg->scvi_inflation == 0;
  m->out_single[1] = cfg->scvi_dispersion == 2; m->out_single[2] = cfg->scvi_inflation == 2;
  m->Bmax = cfg->max_batch;
  int rc = SMX_OK;
  auto fail = [&](int code) { smx_model_destroy(m); return code; };
  if (hipStreamCreate(&m->st) != hipSuccess) { set_error("hipStreamCreate failed"); return fail(SMX_ERR_HIP); }
  // ---- manifest (same order as oracle/sisua_oracle.py:manifest) ----
  const bool bnorm = cfg->batchnorm != 0;
  int h = build_mlp(m, m->enc, "enc", m->G, cfg->n_enc, cfg->enc_units, ST_ENC_DROPOUT, cfg->dropout_enc, bnorm);
  m->t_latW = add_tensor(m, "lat/W", h, m->lat_planes * m->D, m->lat_planes, false);
  m->t_latb = add_tensor(m, "lat/b", 1, m->lat_planes * m->D, m->lat_planes, true);
  if (m->scale) {   // trainable mixture prior: logits [C], means and raw scales [C][D]
    m->t_prLogits = add_tensor(m, "prior/logits", 1, cfg->n_components, 1, true);
    m->t_prLoc = add_tensor(m, "prior/loc", cfg->n_components, m->D, 1, false);
    m->t_prScale = add_tensor(m, "prior/scale", m->scale_tril ? cfg->n_components * m->D : cfg->n_components, m->D, 1, false);   // (tril: row c D + p = row p of L_c)
  }
  if (m->scvi) {
    int hl = build_mlp(m, m->encl, "encl", m->G, cfg->n_encl, cfg->encl_units, ST_ENCL_DROPOUT, cfg->dropout_enc, bnorm);
    m->t_latlW = add_tensor(m, "latl/W", hl, 2, 1, false);
    m->t_latlb = add_tensor(m, "latl/b", 1, 2, 1, true);
  }
  int hd = build_mlp(m, m->dec, "dec", m->D, cfg->n_dec, cfg->dec_units, ST_DEC_DROPOUT, cfg->dropout_dec, bnorm);
  if (m->fvae) {   // discriminator: Dense + bias + leaky ReLU, never BatchNorm / dropout; then the logit layer
    int32_t du[SMX_MAX_LAYERS];
    for (int i = 0; i < cfg->disc_layers; ++i) du[i] = cfg->disc_units;
    const int hu = build_mlp(m, m->disc, "disc", m->D, cfg->disc_layers, du, 0, 0.f, false, cfg->disc_leak);
    int n_out = cfg->n_labels > n_observed ? 0 : 1;   // (the label variables sit behind the observed outputs in the target order)
    for (int j = n_observed; j < cfg->n_labels; ++j) { n_out += cfg->label_dim[j]; m->lab_Pp[j] = round_up(cfg->label_dim[j], 32); }
    m->t_discoutW = add_tensor(m, "discout/W", hu, n_out, 1, false);
    m->t_discoutb = add_tensor(m, "discout/b", 1, n_out, 1, true);
  }
  if (m->scvi) {
    for (int ch = 0; ch < m->k; ++ch) {   // (scvi.py:66-86: no Dense head for a 'share'd plane -- its per-gene vector is out{ch}/b alone)
      if (m->out_has_W[ch]) m->t_outW[ch] = add_tensor(m, "out" + std::to_string(ch) + "/W", hd, m->G, 1, false);
      m->t_outb[ch] = add_tensor(m, "out" + std::to_string(ch) + "/b", 1, m->out_single[ch] ? 1 : m->G, 1, true);   // ('single': one scalar)
    }
  } else {
    m->t_outW[0] = add_tensor(m, "out/W", hd, m->k * m->G, m->k, false);
    m->t_outb[0] = add_tensor(m, "out/b", 1, m->k * m->G, m->k, true);
  }
  for (int j = 0; j < m->n_heads; ++j) {
    SMX_REQUIRE(cfg->label_dim[j] > 0, "label_dim must be > 0");
    SMX_REQUIRE(cfg->label_llk[j] >= SMX_LABEL_NB && cfg->label_llk[j] <= SMX_LABEL_ZINBD, "unknown label likelihood");
    if (cfg->label_llk[j] >= SMX_LABEL_MIXNB && cfg->label_llk[j] <= SMX_LABEL_MIXZINB) SMX_REQUIRE(cfg->label_components[j] >= 2 && cfg->label_components[j] <= 4, "mixture label heads have 2..4 components");
    if (cfg->label_llk[j] == SMX_LABEL_MIXTRIL) SMX_REQUIRE(cfg->label_dim[j] <= 64, "'mixtril' label heads take at most 64 label dimensions");
    // planes of the head: (log total_count, logits) | logits | C x (mixture logit, two component parameters) | 'mixtril': C mixture-logit
    // planes, C location planes, C x P planes = the columns of the components' scale factors (label_tril_kernel)
    m->lab_ky[j] = (cfg->label_llk[j] == SMX_LABEL_NB || cfg->label_llk[j] == SMX_LABEL_NBD) ? 2 : (cfg->label_llk[j] == SMX_LABEL_ZINB || cfg->label_llk[j] == SMX_LABEL_ZINBD) ? 3 :
                   cfg->label_llk[j] == SMX_LABEL_ONEHOT ? 1 :
                   cfg->label_llk[j] == SMX_LABEL_MIXTRIL ? cfg->label_components[j] * (2 + cfg->label_dim[j]) :
                   cfg->label_llk[j] == SMX_LABEL_MIXZINB ? 4 * cfg->label_components[j] : 3 * cfg->label_components[j];
    m->lab_Pp[j] = round_up(cfg->label_dim[j], 32);
    m->t_labW[j] = add_tensor(m, "lab" + std::to_string(j) + "/W", hd, m->lab_ky[j] * cfg->label_dim[j], m->lab_ky[j], false);
    m->t_labb[j] = add_tensor(m, "lab" + std::to_string(j) + "/b", 1, m->lab_ky[j] * cfg->label_dim[j], m->lab_ky[j], true);
  }
  // ---- flat buffers ----
  // layout: [encoder / latent / decoder tensors | tail: BN batch stats, metrics | output + label heads].
  // The head gradients (3/4 of the bytes) are produced by the FIRST backward launch, so under data
  // parallelism that contiguous bucket is all-reduced on a communication stream while the rest of
  // the backward pass runs; the front bucket (with the tail) follows when backward is done.
  size_t off = 0;
  for (size_t i = 0; i < m->bn_w.size(); ++i) { m->bn_off.push_back(off); off += 2 * (size_t)m->bn_wp[i]; }
  m->bn_total = off;
  {
    auto is_head = [](const std::string& n) { return n.compare(0, 3, "out") == 0 || n.compare(0, 3, "lab") == 0; };
    size_t cur = 0;
    for (auto& t : m->tensors) if (!is_head(t.name)) { t.offset = cur; cur += (t.count + 63) / 64 * 64; }
    m->tail_off_bn = cur;
    m->tail_off_metrics = m->tail_off_bn + (m->bn_total + 63) / 64 * 64;
    cur = m->tail_off_metrics + 64;
    m->bucket1_off = cur;
    for (auto& t : m->tensors) if (is_head(t.name)) { t.offset = cur; cur += (t.count + 63) / 64 * 64; }
    m->bucket1_count = cur - m->bucket1_off;
    m->flat_count = cur;
    m->grads_count = cur;
  }
  // (SMX_SHARD_SLACK floats behind the head bucket: flag opt_shard cuts the bucket into world slices of equal, 64-float-aligned length)
  if ((rc = dmalloc(&m->params, m->flat_count + SMX_SHARD_SLACK))) return fail(rc);
  if ((rc = dmalloc(&m->grads, m->grads_count + SMX_SHARD_SLACK))) return fail(rc);
  if ((rc = dmalloc(&m->adam_m, m->flat_count + SMX_SHARD_SLACK))) return fail(rc);
  if ((rc = dmalloc(&m->adam_v, m->flat_count + SMX_SHARD_SLACK))) return fail(rc);
  if ((rc = dmalloc(&m->bn_moving, m->bn_total))) return fail(rc);
  {  // moving variance starts at 1 (Keras)
    std::vector<float> init(m->bn_total, 0.f);
    for (size_t i = 0; i < m->bn_w.size(); ++i)
      for (int j = 0; j < m->bn_wp[i]; ++j) init[m->bn_off[i] + m->bn_wp[i] + j] = 1.f;
    if (m->bn_total && hipMemcpy(m->bn_moving, init.data(), init.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
      set_error("bn init copy failed"); return fail(SMX_ERR_HIP);
    }
  }
  // ---- activations ----
  const size_t B = m->Bmax;
  m->max_feat_p = m->Dp;
  auto alloc_mlp = [&](std::vector<MlpLayer>& mlp) {
    for (auto& L : mlp) {
      if (L.out_p > m->max_feat_p) m->max_feat_p = L.out_p;
      if ((rc = dmalloc(&L.xhat, B * L.out_p))) return rc;
      if ((rc = dmalloc(&L.out_buf, B * L.out_p))) return rc;
      if ((rc = dmalloc(&L.dpre, B * L.out_p))) return rc;
      if ((rc = dmalloc(&L.inv_std, (size_t)L.out_p))) return rc;
      if ((rc = dmalloc(&L.noise, B * L.out_p))) return rc;
    }
    return (int)SMX_OK;
  };
  if ((rc = alloc_mlp(m->enc)) || (rc = alloc_mlp(m->encl)) || (rc = alloc_mlp(m->dec))) return fail(rc);
  if (m->fvae) {   // the discriminator sees the stacked batch [z ; z_perm]: 2 B rows
    int up = 32;
    for (auto& L : m->disc) {
      up = std::max(up, L.out_p);
      if (L.out_p > m->max_feat_p) m->max_feat_p = L.out_p;
      if ((rc = dmalloc(&L.xhat, 2 * B * L.out_p)) || (rc = dmalloc(&L.out_buf, 2 * B * L.out_p)) || (rc = dmalloc(&L.dpre, 3 * B * L.out_p)))   // (d pre: both backward sweeps as rows of one buffer)
        return fail(rc);
    }
    // the two upstream gradients on the logits in ONE buffer: rows [0, 2B) the discriminator's objective, rows [2B, 3B) the TC term
    if ((rc = dmalloc(&m->zz, 2 * B * m->Dp)) || (rc = dmalloc(&m->u_d, 3 * B * 32)) ||
        (rc = dmalloc(&m->tc_cell, B)) || (rc = dmalloc(&m->dl_cell, 2 * B)) || (rc = dmalloc(&m->dz_tc, B * m->Dp)) ||
        (rc = dmalloc(&m->disc_dpre, B * up)) || (rc = dmalloc(&m->disc_db, (size_t)up)))
      return fail(rc);
  }
  m->slab_cap = (size_t)(64 * 3 + SMX_MAX_LABELS + 1) * B * m->max_feat_p;
  const size_t lat_ld = (size_t)m->lat_planes * (size_t)m->Dp;
  const size_t ldp = (size_t)m->k * m->Gp;
  // wide panels: scratch for the per-slice slabs of the products that contract over the gene axis (smx_bigk.hip)
  if (m->Gp >= std::min(4096, head_fused_min_genes())) {
    m->bigk_floats = (size_t)SMX_BIGK_MAX_SLICES * B * m->max_feat_p;
    if ((rc = dmalloc(&m->bigk_part, m->bigk_floats))) return fail(rc);
    float* tab = nullptr;
    if ((rc = dmalloc(&tab, (size_t)SMX_HEAD_FUSED_TAB_BYTES / 4))) return fail(rc);
    m->hf_tab = tab;
    if ((rc = head_fused_prepare())) return fail(rc);
  }
  if ((rc = dmalloc(&m->slab, m->slab_cap)) || (rc = dmalloc(&m->latbuf, B * lat_ld)) || (rc = dmalloc(&m->dlat, B * lat_ld)) ||
      (rc = dmalloc(&m->z, B * m->Dp)) || (rc = dmalloc(&m->noise_eps, B * m->Dp)) || (rc = dmalloc(&m->sig, B * m->Dp)) || (rc = dmalloc(&m->eps, B * m->Dp)) ||
      (rc = dmalloc(&m->kl, B)) || (rc = dmalloc(&m->P, B * ldp)) || (rc = dmalloc(&m->dP, B * ldp)) ||
      (rc = dmalloc(&m->llk_part, B * (size_t)std::max(std::max(loss_chunks_max(m->Gp), head_loss_chunks(m->Gp)), 256))) || (rc = dmalloc(&m->llk_y, B)) || (rc = dmalloc(&m->llk_o, B)) ||
      (rc = dmalloc(&m->rows2[0], B)) || (rc = dmalloc(&m->rows2[1], B)) || (rc = dmalloc(&m->state3, (size_t)3)) ||
      (rc = dmalloc(&m->hostX, B * m->Gp)) || (rc = dmalloc(&m->hostLib, B * 2)) || (rc = dmalloc(&m->hostLgx1, B)))
    return fail(rc);
  if (m->scale && ((rc = dmalloc(&m->resp, B * 32)) || (rc = dmalloc(&m->dklz, B * m->Dp)))) return fail(rc);
  if (m->scale_tril) {
    m->tril_part_floats = (size_t)cfg->n_components * ((B + 7) / 8) * m->D * (m->D + 2);
    if ((rc = dmalloc(&m->tril_part, m->tril_part_floats))) return fail(rc);
  }
  if (m->mixpost && ((rc = dmalloc(&m->resp, B * 32)) || (rc = dmalloc(&m->zmean, B * m->Dp)) || (rc = dmalloc(&m->zpick, B)))) return fail(rc);
  if (m->scvi) {
    if ((rc = dmalloc(&m->raw, B * ldp)) || (rc = dmalloc(&m->draw, B * ldp)) || (rc = dmalloc(&m->rho, B * m->Gp)) ||
        (rc = dmalloc(&m->latlbuf, B * 32)) || (rc = dmalloc(&m->dlatl, B * 32)) || (rc = dmalloc(&m->lsmp, B)) ||
        (rc = dmalloc(&m->lsig, B)) || (rc = dmalloc(&m->leps, B)) || (rc = dmalloc(&m->kl_l, B)) || (rc = dmalloc(&m->dl, B)))
      return fail(rc);
  }
  for (int j = 0; j < m->n_heads; ++j) {
    const size_t ld = m->tensors[m->t_labW[j]].ld;
    if ((rc = dmalloc(&m->laby_raw[j], B * ld)) || (rc = dmalloc(&m->laby_draw[j], B * ld))) return fail(rc);
  }
  // ---- optimiser chunk table ----
  std::vector<OptChunk> chunks;
  // floats per optimiser workgroup
  const int CH = 4096;
  for (size_t t = 0; t < m->tensors.size(); ++t) {
    const TensorInfo& ti = m->tensors[t];
    const int first = (int)chunks.size();
    const int n = (int)((ti.count + CH - 1) / CH);
    for (int i = 0; i < n; ++i) {
      OptChunk c;
      memset(&c, 0, sizeof(c));
      c.tensor = (int)t; c.offset = (int)(ti.offset + (size_t)i * CH);
      c.count = (int)((size_t)(i + 1) * CH <= ti.count ? CH : ti.count - (size_t)i * CH);
      c.first_chunk = first; c.n_chunks = n; c.tensor_count = (int32_t)ti.count;
      chunks.push_back(c);
    }
  }
  m->n_chunks = (int)chunks.size();
  m->chunks_floats = CH;
  m->chunk_first_head = m->n_chunks;
  for (size_t i = 0; i < chunks.size(); ++i)
    if (chunks[i].tensor == m->t_outW[0]) { m->chunk_first_head = (int)i; break; }
  m->chunk_first_label = m->n_chunks;
  if (m->n_heads > 0)
    for (size_t i = 0; i < chunks.size(); ++i)
      if (chunks[i].tensor == m->t_labW[0]) { m->chunk_first_label = (int)i; break; }
  if (m->tensors.size() <= SMX_MAX_TENSORS) {   // slots for the products' sum-of-squares partials (32 x 32 tiles at most)
    size_t total = 0;
    m->sq_first.assign(m->tensors.size(), 0);
    m->sq_count.assign(m->tensors.size(), 0);
    for (size_t t = 0; t < m->tensors.size(); ++t) {
      const TensorInfo& ti = m->tensors[t];
      m->sq_first[t] = (int)total;
      total += (size_t)((ti.rows_p + 31) / 32) * (size_t)((ti.ld + 31) / 32) * 8;   // (the wide head-backward kernel leaves 8 per tile when the planes are separate tensors)
    }
    m->sq_total_first = (int)total;   // SMX_SQR_PER_TENSOR more slots per tensor: the sums the reduce riders leave (attach_early_adam)
    m->sq_reduced.assign(m->tensors.size(), 0);
    if ((rc = dmalloc(&m->sq_slots, total + m->tensors.size() * SMX_SQR_PER_TENSOR))) return fail(rc);
  }
  if ((rc = dmalloc(&m->chunks, chunks.size())) || (rc = dmalloc(&m->partial, chunks.size())) || (rc = dmalloc(&m->shard_partial, chunks.size())) ||
      (rc = dmalloc(&m->tensor_norm, m->tensors.size())))
    return fail(rc);
  if (hipMemcpy(m->chunks, chunks.data(), chunks.size() * sizeof(OptChunk), hipMemcpyHostToDevice) != hipSuccess) {
    set_error("chunk table copy failed"); return fail(SMX_ERR_HIP);
  }
  *out = m;
  return SMX_OK;
}

int smx_model_destroy(smx_model* m) {
  if (!m) return SMX_OK;
  if (m->st) hipStreamSynchronize(m->st);
  for (auto& kv : m->graphs) hipGraphExecDestroy(kv.second);
  for (auto& ev : m->timing_events) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
  if (m->bigk_part) hipFree(m->bigk_part);
  if (m->hf_tab) hipFree(m->hf_tab);
  p2p_release(m);
  if (m->comm2 && g_rccl.CommDestroy) g_rccl.CommDestroy(m->comm2);   // (the heads' bucket's communicator: a split of `comm`, destroyed first)
  m->comm2 = nullptr;
  if (m->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(m->comm);
  m->comm = nullptr;
  m->local.reset();
  auto fr = [](void* p) { if (p) hipFree(p); };
  fr(m->local_scratch); fr(m->sync_buf);
  fr(m->params); fr(m->grads); fr(m->adam_m); fr(m->adam_v); fr(m->bn_moving);
  for (auto* mlp : {&m->enc, &m->encl, &m->dec})
    for (auto& L : *mlp) { fr(L.xhat); fr(L.out_buf); fr(L.dpre); fr(L.inv_std); fr(L.noise); }
  release_csr(m);   // (the sparse store: m->X aliased its expansion tile)
  fr(m->X); fr(m->library); fr(m->mask); fr(m->lgx1); fr(m->hostX); fr(m->hostLib); fr(m->hostLgx1);
  for (int j = 0; j < SMX_MAX_LABELS; ++j) { fr(m->Y[j]); fr(m->laby_raw[j]); fr(m->laby_draw[j]); }
  fr(m->rows2[0]); fr(m->rows2[1]); fr(m->order); fr(m->state3); fr(m->mhist);
  fr(m->resp); fr(m->dklz); fr(m->zmean); fr(m->zpick); fr(m->tril_part);
  for (auto& L : m->disc) { fr(L.xhat); fr(L.out_buf); fr(L.dpre); }
  fr(m->zz); fr(m->u_d); fr(m->tc_cell); fr(m->dl_cell); fr(m->dz_tc); fr(m->disc_dpre); fr(m->disc_db);
  fr(m->noise_eps); fr(m->latbuf); fr(m->dlat); fr(m->z); fr(m->sig); fr(m->eps); fr(m->kl);
  fr(m->latlbuf); fr(m->dlatl); fr(m->lsmp); fr(m->lsig); fr(m->leps); fr(m->kl_l); fr(m->dl);
  fr(m->P); fr(m->dP); fr(m->raw); fr(m->draw); fr(m->rho); fr(m->llk_part); fr(m->llk_y); fr(m->llk_o); fr(m->slab);
  fr(m->chunks); fr(m->partial); fr(m->shard_partial); fr(m->tensor_norm); fr(m->sq_slots);
  if (m->pinned) hipHostFree(m->pinned);
  if (m->order_pin) hipHostFree(m->order_pin);
  if (m->metrics_pin) hipHostFree(m->metrics_pin);
  if (m->score_pin) hipHostFree(m->score_pin);
  if (m->ev_order) hipEventDestroy(m->ev_order);
  if (m->pred_stage) hipFree(m->pred_stage);
  if (m->pred_target) hipFree(m->pred_target);
  if (m->pred_ids) hipFree(m->pred_ids);
  if (m->score_buf) hipFree(m->score_buf);
  if (m->score_wimg) hipFree(m->score_wimg);
  if (m->score_aux) hipFree(m->score_aux);
  for (auto& kv : m->injected) fr(kv.second.d);
  if (m->st_comm) { hipStreamSynchronize(m->st_comm); hipStreamDestroy(m->st_comm); }
  if (m->st_side) { hipStreamSynchronize(m->st_side); hipStreamDestroy(m->st_side); }
  if (m->ev_hf) hipEventDestroy(m->ev_hf);
  if (m->ev_sweep) hipEventDestroy(m->ev_sweep);
  if (m->ev_c1) hipEventDestroy(m->ev_c1);
  if (m->ev_c2) hipEventDestroy(m->ev_c2);
  if (m->ev_c3) hipEventDestroy(m->ev_c3);
  if (m->st) hipStreamDestroy(m->st);
  delete m;
  return SMX_OK;
}

int smx_num_tensors(const smx_model* m) { return m ? (int)m->tensors.size() : 0; }

int smx_tensor_info(const smx_model* m, int index, char* name, int name_cap, int32_t* rows, int32_t* cols) {
  SMX_REQUIRE(m && index >= 0 && index < (int)m->tensors.size(), "tensor index out of range");
  const TensorInfo& t = m->tensors[index];
  if (name && name_cap > 0) { strncpy(name, t.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (rows) *rows = t.rows;
  if (cols) *cols = t.cols;
  return SMX_OK;
}

static float* which_buf(smx_model* m, int which) {
  switch (which) { case 0: return m->params; case 1: return m->grads; case 2: return m->adam_m; case 3: return m->adam_v; }
  return nullptr;
}

int smx_get_tensor(smx_model* m, int which, int index, float* host) {
  SMX_REQUIRE(m && host && index >= 0 && index < (int)m->tensors.size(), "bad tensor index");
  float* base = which_buf(m, which);
  SMX_REQUIRE(base, "which must be 0..3");
  const TensorInfo& t = m->tensors[index];
  SMX_REQUIRE(!(m->opt_stale && (which == 2 || which == 3) && t.offset >= m->bucket1_off),
              "the heads' Adam moments are sharded over the ranks (flag opt_shard): call smx_opt_gather on every rank first");
  std::vector<float> dev(t.count);
  SMX_HIP(hipStreamSynchronize(m->st));
  SMX_HIP(hipMemcpy(dev.data(), base + t.offset, t.count * sizeof(float), hipMemcpyDeviceToHost));
  unpack(t, dev, host, 1.f);
  return SMX_OK;
}

int smx_set_tensor(smx_model* m, int which, int index, const float* host) {
  SMX_REQUIRE(m && host && index >= 0 && index < (int)m->tensors.size(), "bad tensor index");
  float* base = which_buf(m, which);
  SMX_REQUIRE(base, "which must be 0..3");
  const TensorInfo& t = m->tensors[index];
  std::vector<float> dev;
  pack(t, host, dev);
  SMX_HIP(hipStreamSynchronize(m->st));
  SMX_HIP(hipMemcpy(base + t.offset, dev.data(), t.count * sizeof(float), hipMemcpyHostToDevice));
  if (which == 0) ++m->params_epoch;
  return SMX_OK;
}

int smx_num_bn_layers(const smx_model* m) { return m ? (int)m->bn_w.size() : 0; }

int smx_get_bn(smx_model* m, int layer, int which, float* host, int32_t* width) {
  SMX_REQUIRE(m && layer >= 0 && layer < (int)m->bn_w.size() && (which == 0 || which == 1), "bad bn layer");
  if (width) *width = m->bn_w[layer];
  if (host) {
    SMX_HIP(hipStreamSynchronize(m->st));
    SMX_HIP(hipMemcpy(host, m->bn_moving + m->bn_off[layer] + (size_t)which * m->bn_wp[layer],
                      m->bn_w[layer] * sizeof(float), hipMemcpyDeviceToHost));
  }
  return SMX_OK;
}

int smx_set_bn(smx_model* m, int layer, int which, const float* host) {
  SMX_REQUIRE(m && host && layer >= 0 && layer < (int)m->bn_w.size() && (which == 0 || which == 1), "bad bn layer");
  SMX_HIP(hipStreamSynchronize(m->st));
  SMX_HIP(hipMemcpy(m->bn_moving + m->bn_off[layer] + (size_t)which * m->bn_wp[layer], host,
                    m->bn_w[layer] * sizeof(float), hipMemcpyHostToDevice));
  return SMX_OK;
}

int smx_get_step(const smx_model* m, int32_t* step) {
  SMX_REQUIRE(m && step, "null argument");
  *step = (int32_t)m->h_next;
  return SMX_OK;
}

int smx_set_step(smx_model* m, int32_t step) {
  SMX_REQUIRE(m && step >= 0, "bad step");
  StepState s[3];
  memset(s, 0, sizeof(s));
  s[2].next = (uint32_t)step;
  SMX_HIP(hipStreamSynchronize(m->st));
  SMX_HIP(hipMemcpy(m->state3, s, sizeof(s), hipMemcpyHostToDevice));
  m->h_next = (uint32_t)step;
  return SMX_OK;
}

int smx_set_noise(smx_model* m, int32_t stream, const float* data, int32_t batch, int32_t width) {
  SMX_REQUIRE(m && data && batch > 0 && batch <= m->Bmax && width > 0, "bad noise block");
  const int ld = round_up(width, 32);
  Injected& ij = m->injected[stream];
  if (ij.d && ij.ld != ld) { hipFree(ij.d); ij.d = nullptr; }
  if (!ij.d) { SMX_CHECK(dmalloc(&ij.d, (size_t)m->Bmax * ld)); ij.ld = ld; }
  SMX_HIP(hipStreamSynchronize(m->st));
  SMX_HIP(hipMemset(ij.d, 0, (size_t)m->Bmax * ld * sizeof(float)));
  SMX_HIP(hipMemcpy2D(ij.d, (size_t)ld * sizeof(float), data, (size_t)width * sizeof(float), (size_t)width * sizeof(float),
                      (size_t)batch, hipMemcpyHostToDevice));
  m->use_injected = true;
  return SMX_OK;
}

int smx_clear_noise(smx_model* m) {
  SMX_REQUIRE(m, "null model");
  SMX_HIP(hipStreamSynchronize(m->st));
  for (auto& kv : m->injected) if (kv.second.d) hipFree(kv.second.d);
  m->injected.clear();
  m->use_injected = false;
  return SMX_OK;
}

int smx_set_flag(smx_model* m, const char* name, int value) {
  SMX_REQUIRE(m && name, "null argument");
  SMX_HIP(hipStreamSynchronize(m->st));
  const std::string n(name);
  int* f = n == "head_loss" ? &m->flags.head_loss : n == "head_fused" ? &m->flags.head_fused : n == "head_sweep" ? &m->flags.head_sweep : n == "front" ? &m->flags.front : n == "bwd_front" ? &m->flags.bwd_front
         : n == "head_bwd" ? &m->flags.head_bwd : n == "wgrad" ? &m->flags.wgrad : n == "scvi_fused" ? &m->flags.scvi_fused
         : n == "twin" ? &m->flags.twin : n == "label_ride" ? &m->flags.label_ride : n == "act_epilogue" ? &m->flags.act_epilogue
         : n == "stacked_scoring" ? &m->flags.stacked_scoring : n == "bf16x3" ? &m->flags.bf16x3 : n == "opt_shard" ? &m->flags.opt_shard
         : n == "tie_mixtures" ? &m->flags.tie_mixtures : n == "tie_loc" ? &m->flags.tie_loc : n == "tie_scale" ? &m->flags.tie_scale : nullptr;
  SMX_REQUIRE(f, "unknown flag (head_loss, head_fused, head_sweep, front, bwd_front, head_bwd, wgrad, scvi_fused, twin, label_ride, act_epilogue, stacked_scoring, bf16x3, opt_shard; SCALE: tie_mixtures, tie_loc, tie_scale)");
  *f = (f == &m->flags.bf16x3 && value < 0) ? -1 : (value ? 1 : 0);   // bf16x3: -1 = by the width of the head (the default)
  drop_graphs(m);   // a captured step bakes the launch sequence in
  return SMX_OK;
}

int smx_timing_enable(smx_model* m, const char* kernel) {
  SMX_REQUIRE(m, "null model");
  SMX_HIP(hipStreamSynchronize(m->st));
  m->timing_label = kernel ? kernel : "";
  m->timing_reps = SMX_LOSS_TIMING_REPEAT;
  // "label@N": the idempotent likelihood kernels are launched N times per event pair instead of the default 8
  // (N = 1: one launch inside the pair, the figure rocprofv3's per-kernel duration is compared with)
  const size_t at = m->timing_label.find('@');
  if (at != std::string::npos) {
    m->timing_reps = std::max(1, atoi(m->timing_label.c_str() + at + 1));
    m->timing_label.resize(at);
  }
  m->timing_used = 0;
  return SMX_OK;
}

int smx_timing_read(smx_model* m, double* total_ms, int64_t* launches) {
  SMX_REQUIRE(m && total_ms && launches, "null argument");
  SMX_HIP(hipStreamSynchronize(m->st));
  double tot = 0.0;
  for (size_t i = 0; i < m->timing_used; ++i) {
    float ms = 0.f;
    SMX_HIP(hipEventElapsedTime(&ms, m->timing_events[i].first, m->timing_events[i].second));
    tot += ms;
  }
  *total_ms = tot; *launches = (int64_t)m->timing_used;
  m->timing_used = 0;
  return SMX_OK;
}

int64_t smx_loss_bytes_per_cell(const smx_model* m) {
  if (!m) return 0;
  return (int64_t)(4 + 8 * m->k) * m->G + 16 * (int64_t)m->D + 4;
}

// (forward_pass's own predicate, smx_step.hip: head_fused_ok)
int64_t smx_head_fused_bytes(const smx_model* m, int32_t batch) {
  if (!m || batch <= 0 || m->dec.empty()) return 0;
  return head_fused_ok(m, batch) ? (int64_t)head_fused_bytes(batch, m->G, m->Gp, m->k) : 0;
}

}  // extern "C"
