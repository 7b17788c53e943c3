// smx_model.h -- the model's state and the helpers shared by the host-side translation units (private to csrc/).
//
// HBM layout (all fp32, row-major, every feature axis padded to a multiple of 32
// so rows are 128-byte aligned and float4 accesses never straddle a row):
//   X        [n_cells][Gp]            resident counts (zero padded), gathered by row id
//   params   flat buffer, tensors in manifest order; W as [in_p][chunks*chunk_wp]
//            (output head: k planes of Gp; latent head: mu | s planes of Dp)
//   grads    same layout + tail [BN batch stats | 8 metric scalars]: ONE buffer,
//            ONE all-reduce per step under data parallelism
//   adam m/v same layout
//   P, dP    [B][k*Gp] distribution parameter planes and their gradients
// Padded rows/columns of every weight stay exactly zero (their gradients are zero
// by construction), so padded lanes never leak into logical results.
#pragma once
#include <dlfcn.h>
#include <limits.h>
#include <math.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <chrono>
#include <map>
#include <string>
#include <vector>

#include "../../include/sisua_hip.h"
#include "smx_internal.h"

using namespace smx;

#define SMX_CHECK(expr)            \
  do {                             \
    int rc_ = (expr);              \
    if (rc_ != SMX_OK) return rc_; \
  } while (0)
#define SMX_REQUIRE(cond, msg)                 \
  do {                                         \
    if (!(cond)) { set_error(msg); return SMX_ERR_INVALID; } \
  } while (0)

#define SMX_LOSS_TIMING_REPEAT 8
#define SMX_BIGK_MAX_SLICES 256   // about one K slice per CU (smx_bigk.hip)
enum { ST_INPUT_DROPOUT = 0, ST_ENC_DROPOUT = 16, ST_ENCL_DROPOUT = 32, ST_DEC_DROPOUT = 48, ST_EPS_Z = 64, ST_EPS_L = 65 };

namespace smx {

struct TensorInfo {
  std::string name;
  int rows = 1, cols = 0;               // logical
  int chunks = 1, chunk_w = 0, chunk_wp = 0;
  int rows_p = 1, ld = 0;
  size_t offset = 0, count = 0;
};

struct MlpLayer {
  int in = 0, in_p = 0, out = 0, out_p = 0;
  int tW = -1, tGamma = -1, tBeta = -1, tBias = -1;
  int bn = -1;
  int stream = 0;
  float drop_p = 0.f;
  float leak = 0.f;         // activation slope for y <= 0 (0: ReLU; the FactorVAE discriminator: 0.2)
  float *xhat = nullptr, *out_buf = nullptr, *inv_std = nullptr, *dpre = nullptr;
  float* noise = nullptr;   // [Bmax][out_p] dropout multipliers drawn ahead of the layer (first decoder layer: by the first encoder BatchNorm launch)
};

struct Injected { float* d = nullptr; int ld = 0; };

struct RcclApi {
  void* lib = nullptr;
  std::string path, hip_path;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;   // (optional: flag opt_shard)
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, void*) = nullptr;   // (optional: a second communicator for the heads' bucket)
  ncclResult_t (*GetVersion)(int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
extern RcclApi g_rccl;
int load_rccl();   // smx_comm.hip

// ---- loopback communicator (test hook, smx_comm_init_local) -----------------------------------------------
// N models of ONE process on ONE device, each driven by its own host thread, all-reduce their flat buffers through
// events and a summing kernel instead of RCCL: the whole world > 1 arithmetic of the step (loss scaling by the
// global batch, the norm of the reduced gradient, averaged moving statistics, SyncBatchNorm's mid-pass
// collectives) runs on the single GPU of a test box.  Summation order is rank 0..N-1 on every rank.
#define SMX_LOCAL_MAX 8
struct LocalGroup {
  std::mutex mu;
  std::condition_variable cv;
  int world = 0, arrived = 0;
  uint64_t gen = 0;
  bool broken = false;
  const float* src[SMX_LOCAL_MAX] = {};          // this collective's source pointer of every rank
  hipEvent_t ready[SMX_LOCAL_MAX] = {}, done[SMX_LOCAL_MAX] = {};
  ~LocalGroup() {
    for (int r = 0; r < SMX_LOCAL_MAX; ++r) { if (ready[r]) hipEventDestroy(ready[r]); if (done[r]) hipEventDestroy(done[r]); }
  }
  // host rendezvous of the member threads; false after a timeout (a member died) -- the group is then unusable
  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (broken) return false;
    const uint64_t g0 = gen;
    if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); return true; }
    if (!cv.wait_for(lk, std::chrono::seconds(20), [&] { return gen != g0 || broken; }) || broken) { broken = true; cv.notify_all(); return false; }
    return true;
  }
};

// ---- peer-to-peer all-reduce over IPC-mapped buffers (smx_p2p.hip) ----------------------------------------------------
#define SMX_P2P_MAX 8
struct P2PState {
  int rank = 0, world = 1, pending_world = 0; unsigned epoch = 0; bool owns_region = false;
  float* grads[SMX_P2P_MAX] = {};          // every rank's flat gradient buffer (own: m->grads; peers: IPC-mapped)
  void* region_base[SMX_P2P_MAX] = {};     // every rank's communication region
  unsigned* flags[SMX_P2P_MAX] = {};       // ... its flag block [2][SMX_P2P_MAX]
  float* staging[SMX_P2P_MAX] = {};        // ... its reduce-scatter output (1 / world of the buffer)
  float* scratch[SMX_P2P_MAX] = {};        // ... its small-buffer scratch
  size_t staging_floats = 0, scratch_floats = 0;
  unsigned* done = nullptr; unsigned* error = nullptr;   // local words behind the flags
  long long timeout_ticks = 3000000000LL;                // bound of a device-side wait, 100 MHz ticks (SMX_P2P_TIMEOUT_S)
};
int p2p_allreduce(smx_model* m, float* buf, size_t count, hipStream_t st);
bool p2p_selected(const smx_model* m);   // the hand-written exchange is attached AND the form in force takes it (smx_comm.hip)
void p2p_release(smx_model* m);
int ensure_sync_buf(smx_model* m);
int ensure_comm_stream(smx_model* m);

}  // namespace smx

struct smx_model {
  smx_config cfg;
  int device = 0;
  hipStream_t st = nullptr;
  bool head_loss = false;             // this step's output product ran fused with the likelihood (smx_headloss.hip)
  // code-path switches (smx_set_flag; defaults from the SMX_NO_* environment variables): 1 = the default wide / fused
  // kernels, 0 = the separate-launch forms they replaced (kept for eval, for shapes the fused kernels do not take, and
  // as A/B references)
  struct Flags {
    int head_loss = tuning_on("no_head_loss") ? 0 : 1;    // output product + likelihood in one kernel
    int head_fused = tuning_on("no_head_fused") ? 0 : 1;  // wide panels: ... and both of the head's backward products in the same launch (smx_headfused.hip)
    int head_sweep = tuning_on("no_head_sweep") ? 0 : 1;  // wide panels, one GPU, eager steps: the heads' optimiser update as a background sweep on a second stream (smx_step.hip)
    int front = tuning_on("no_front") ? 0 : 1;            // latent sample + first decoder product inside BatchNorm-forward
    int bwd_front = tuning_on("no_bwd_front") ? 0 : 1;    // d h inside BatchNorm-backward, weight gradients grouped at the end
    int head_bwd = tuning_on("no_head_bwd") ? 0 : 1;      // both backward products of the output head in one wide launch
    int wgrad = tuning_on("no_wgrad") ? 0 : 1;            // K = minibatch weight gradients as the wide direct-operand kernel
    int scvi_fused = tuning_on("no_scvi_fused") ? 0 : 1;  // scvi: library latent + softmax head + likelihood + their backward as one row-local launch
    int twin = tuning_on("no_twin") ? 0 : 1;              // scvi: first layers of both encoders (and pairs of heads) side by side in one launch
    int act_epilogue = tuning_on("no_act_epilogue") ? 0 : 1;  // layers without BatchNorm / dropout: bias + activation (and its derivative) in the products' store paths
    int label_ride = tuning_on("no_label_ride") ? 0 : 1;  // label heads' backward inside the output head's backward launch + the final grouped launch
    int stacked_scoring = tuning_on("no_stacked_scoring") ? 0 : 1;  // marginal_llk: all posterior draws as rows of ONE decoder pass
    // training products of the output head (fused head, its backward, the encoder's weight gradient) from bf16 MFMAs on
    // three-way split operands: 1 always, 0 never (the exact-f32 MFMA forms), -1 from the width (SMX_BF16X3_MIN_WORK)
    int bf16x3 = -1;
    // data parallel, chained two-bucket form: the heads' optimiser state SHARDED over the ranks (smx_step.hip: dp_chain_start) -- reduce-scatter of
    // the head bucket, clip + Adam on this rank's 1 / world slice, all-gather of the updated parameters.  Off by default (the north star: one all-reduce).
    int opt_shard = 0;
  int tie_mixtures = 0, tie_loc = 0, tie_scale = 0;   // SCALE (scale.py:29-33): the prior's mixture weights fixed / one location / one scale for every component
  } flags;
  int chunk_first_head = 0;           // first optimiser chunk of the output / label heads (they are last in the table)
  int chunk_first_label = 0;          // first optimiser chunk of the label heads (n_chunks without label heads)
  bool lab_deferred = false;          // this step's label-head weight gradients come with the grouped launch at the END of backward
  int adam_early_to = -1;             // chunks [adam_early_from, adam_early_to) of this step were applied early
  bool adam_early_pending = false;    // the heads' gradients are final: the next BatchNorm-backward launch may carry their update
  int adam_early_from = -1;           // >= 0: chunks [adam_early_from, n_chunks) of this step were applied early
  int adam_ride_b = 0;                // wide panels: this many of the waiting chunks go with the latent head's backward product
  int adam_rest_from = 0, adam_rest_to = 0;   // ... and [adam_rest_from, adam_rest_to) wait for the next BatchNorm-backward launch to carry them
  // wide panels, one GPU, eager steps: the heads' update as a background sweep on a second stream between this step's output head and
  // the next step's (smx_step.hip: head_sweep_start / head_sweep_join)
  hipStream_t st_side = nullptr;
  hipEvent_t ev_hf = nullptr, ev_sweep = nullptr; int ev_mode = -1;   // (ev_mode: created with 1 / without 0 the system-scope fence: head_sweep_prepare)
  bool sweep_pending = false;          // the main stream has not been ordered behind the last sweep yet
  bool ev_hf_fresh = false;            // ev_hf was recorded behind THIS step's output head
  bool head_fused_bwd_done = false;   // this backward pass found dW / db / d d of the output head done by the forward pass's launch
  bool x_u16 = false;   // the resident matrix is stored as uint16 counts (smx_dataset_upload_u16)
  // compact sparse store (smx_dataset_upload_csr): CSR arrays resident, the minibatch's rows expanded per pass into xbatch
  int64_t* csr_indptr = nullptr; int32_t* csr_cols = nullptr; float* csr_vals = nullptr; bool x_csr = false;
  float* xbatch = nullptr;   // [Bmax][Gp]
  float* pred_stage = nullptr; size_t pred_floats = 0;   // device staging of smx_predict (one chunk of cells, laid out like the caller's arrays)
  float* score_buf = nullptr; size_t score_floats = 0;   // smx_marginal_llk, stacked draws: z | lw | two activation buffers | last layer (k-major f32 or bf16 split) | likelihood partials
  float* score_wimg = nullptr; size_t score_wimg_floats = 0;   // the output head's W as bf16 slab images (smx_score.hip)
  float* score_aux = nullptr; size_t score_aux_floats = 0;     // scoring calls: running log-sum-exp state, staged target counts and their row constants
  float* pinned = nullptr; size_t pinned_floats = 0;   // host staging for the parameter planes handed back by smx_forward / smx_decode
  // sum-of-squares slots written by the weight-gradient products (per-tensor clipnorm without a separate pass)
  float* sq_slots = nullptr; std::vector<int> sq_first, sq_count; std::vector<char> sq_reduced; int sq_total_first = 0;
  int G = 0, Gp = 0, D = 0, Dp = 0, k = 0, Bmax = 0;
  bool stochastic = true, scvi = false, scale = false, fvae = false;
  bool mixpost = false;      // SMX_MODEL_SCALE_POST: q(z|x) a mixture of cfg.n_components diagonal Gaussians (lat head: 1 + 2 C planes)
  int lat_planes = 2;        // planes of width Dp of the latent head's output: 2 (mu, raw sigma), 1 (deterministic), 1 + 2 C (mixture posterior)
  float* zmean = nullptr; int32_t* zpick = nullptr;   // mixture posterior: the mixture's mean [B][Dp] (what predict / encode report), the picked component [B]
  float* tril_part = nullptr; size_t tril_part_floats = 0;   // scale_tril: scratch of the prior's backward (partial sums per component and cell group)
  bool scale_tril = false;   // SCALE with full-covariance components (SMX_MODEL_SCALE_TRIL): prior/scale holds C lower-triangular D x D factors
  int n_heads = 0;                    // label heads on the decoder (0 for fvae: SemiFVAE's labels go to the discriminator)
  // fvae: discriminator on z (smx_factor.hip)
  std::vector<MlpLayer> disc; int t_discoutW = -1, t_discoutb = -1;
  float *zz = nullptr, *u_d = nullptr, *tc_cell = nullptr, *dl_cell = nullptr, *dz_tc = nullptr;
  float *disc_dpre = nullptr, *disc_db = nullptr;
  int t_prLogits = -1, t_prLoc = -1, t_prScale = -1;    // scale: Gaussian-mixture prior
  float *resp = nullptr, *dklz = nullptr;
  std::vector<TensorInfo> tensors;
  size_t flat_count = 0, tail_off_bn = 0, tail_off_metrics = 0, grads_count = 0;
  float *params = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr;
  std::vector<MlpLayer> enc, encl, dec;
  int t_latW = -1, t_latb = -1, t_latlW = -1, t_latlb = -1;
  int t_outW[3] = {-1, -1, -1}, t_outb[3] = {-1, -1, -1};
  int t_labW[SMX_MAX_LABELS], t_labb[SMX_MAX_LABELS];
  int lab_ky[SMX_MAX_LABELS], lab_Pp[SMX_MAX_LABELS];
  // batch-norm moving stats: layer i at bn_moving + bn_off[i]: mean[w_p] then var[w_p]
  std::vector<int> bn_w, bn_wp;
  std::vector<size_t> bn_off;
  float* bn_moving = nullptr;
  size_t bn_total = 0;
  // dataset
  float* X = nullptr; int64_t N = 0; int64_t cell_base = 0;
  float* Y[SMX_MAX_LABELS] = {nullptr, nullptr, nullptr, nullptr};
  float* library = nullptr; uint8_t* mask = nullptr; float* lgx1 = nullptr;
  // host-batch staging for smx_forward(host_x)
  float* hostX = nullptr; float* hostLib = nullptr; float* hostLgx1 = nullptr;
  // step state
  int32_t* rows2[2] = {nullptr, nullptr}; int32_t* order = nullptr; size_t order_cap = 0;
  // pinned staging for the row ids of a train_steps call: hipMemcpyAsync from the caller's pageable array cost ~80 us per call
  int32_t* order_pin = nullptr; size_t order_pin_cap = 0; hipEvent_t ev_order = nullptr; bool order_pin_busy = false;
  int32_t* pred_ids = nullptr; int pred_ids_batch = 0;   // smx_predict super-batches: noise ids (row % batch)
  float* pred_target = nullptr; size_t pred_target_floats = 0;   // smx_predict_stat(log_prob): a batch of target rows [Bmax][Gp]
  float* metrics_pin = nullptr;   // pinned landing area of read_metrics: 8 ELBO scalars + one gradient norm per tensor
  float* score_pin = nullptr; size_t score_pin_floats = 0;   // pinned landing area of the scoring entry points' results (smx_scoring.hip: score_landing)
  float* mhist = nullptr; size_t mhist_cap = 0; int32_t mhist_steps = 0;   // ELBO scalars of every step of the last train_steps call
  int32_t staged_steps = 0, staged_batch = 0;   // row ids made resident by smx_train_stage for the next smx_train_steps(order = NULL)
  StepState* state3 = nullptr;  // [0],[1]: per-step state by parity, [2]: master counter
  int par = 0; uint32_t h_next = 0;
  MetricsArgs pending_metrics; bool have_pending_metrics = false, metrics_before_allreduce = false;
  int seq_batch = 0, seq_prepare_next = 0;
  // this pass's first encoder BatchNorm launch has drawn, on otherwise idle CUs, what the decoder's front launch would
  // draw redundantly in each of its workgroups: eps of the latent sample (-> noise_eps) / the dropout multipliers of
  // the first decoder layer (-> dec[0].noise)
  bool ahead_front_eps = false, ahead_front_drop = false;
  bool scvi_fused = false;     // this training pass ran the scvi head as ONE row-local launch (smx_scvi.hip)
  bool encl_twinned = false;   // ... and the library encoder's first layer beside the encoder's (one product + one BatchNorm launch)
  float* noise_eps = nullptr;  // [Bmax][Dp] eps drawn ahead of the latent head
  float *latbuf = nullptr, *dlat = nullptr, *z = nullptr, *sig = nullptr, *eps = nullptr, *kl = nullptr;
  float *latlbuf = nullptr, *dlatl = nullptr, *lsmp = nullptr, *lsig = nullptr, *leps = nullptr, *kl_l = nullptr, *dl = nullptr;
  float *P = nullptr, *dP = nullptr, *raw = nullptr, *draw = nullptr, *rho = nullptr, *llk_part = nullptr;
  float* laby_raw[SMX_MAX_LABELS] = {nullptr, nullptr, nullptr, nullptr};
  float* laby_draw[SMX_MAX_LABELS] = {nullptr, nullptr, nullptr, nullptr};
  float* llk_y = nullptr;
  float* llk_o = nullptr;   // [B] sum of the observed extra outputs' log-likelihoods (cfg.label_observed)
  int n_observed = 0;       // heads [0, n_observed) are observed output variables, the rest label variables
  bool out_single[3] = {false, false, false};   // scvi: plane c is ONE trainable scalar ('single')
  bool out_has_W[3] = {true, true, true};   // scvi: plane c of the gene output is a Dense head (false: a shared per-gene vector, cfg.scvi_dispersion / scvi_inflation)
  float* slab = nullptr; size_t slab_cap = 0; int max_feat_p = 0;
  void* hf_tab = nullptr;             // scratch of the fused output head (smx_headfused.hip: the split views of the decoder output)
  float* shard_partial = nullptr;   // [n_chunks] per-chunk sums of squares of this rank's slice (flag opt_shard), summed over the ranks
  bool opt_stale = false;            // the heads' Adam moments outside this rank's slice are stale (flag opt_shard): smx_opt_gather brings them in
  // parameters written (an optimiser step, smx_set_tensor(which = 0)) -> params_epoch moves on; the scoring head's bf16 images of W_out are kept while it stands
  unsigned long long params_epoch = 1, wimg_epoch = 0, wimg_tuning = 0; int wimg_key = 0;
  long wide_dd_stride = 0;
  const float* wide_dd_src = nullptr;   // where those slabs are (bigk_part or the slab buffer)
  int wide_dd_slabs = 0;   // > 0: this step's d d waits as that many column-major slabs in bigk_part for the decoder's BatchNorm-backward launch (bn_wide_bwd_kernel)
  bool head_fused = false; int head_fused_sq = 0;   // this step's output head ran as ONE launch (loss + dW + db + d d): backward_pass skips its products
  float* bigk_part = nullptr; size_t bigk_floats = 0;   // [SMX_BIGK_MAX_SLICES][Bmax][max_feat_p]: per-slice slabs of smx_bigk.hip (wide panels only)
  // optimiser
  OptChunk* chunks = nullptr; int n_chunks = 0; int chunks_floats = 4096; float* partial = nullptr; float* tensor_norm = nullptr;
  // noise injection
  std::map<int, Injected> injected; bool use_injected = false;
  // comm
  ncclComm_t comm = nullptr; int rank = 0, world = 1;
  ncclComm_t comm2 = nullptr;   // the heads' bucket's own communicator (ncclCommSplit of `comm`): its exchange runs BESIDE the collectives of the main stream
  std::shared_ptr<LocalGroup> local; float* local_scratch = nullptr; size_t local_scratch_cap = 0;   // loopback communicator (tests)
  std::shared_ptr<P2PState> p2p;   // hand-written two-shot all-reduce over IPC-mapped peer buffers (smx_p2p.hip); takes precedence over RCCL
  // SyncBatchNorm (opt-in, smx_comm_set_sync_bn): per BN launch one small all-reduce of per-rank column statistics
  bool sync_bn = false; float* sync_buf = nullptr; size_t sync_cap = 0;
  bool dp_force = false, dp_two_buckets = false;   // SMX_FORCE_ALLREDUCE / SMX_DP_BUCKETS=2, read when the communicator is attached
  // the exchange form asked for (smx_comm_set_form): 0 = the library's rule (bytes of head gradients, the hand-written exchange whenever it is
  // attached), 1 = ONE all-reduce of the flat buffer on the model's stream, 2 = the two-bucket chain, 3 = the hand-written exchange (one bucket)
  int dp_form = 0;
  hipStream_t st_comm = nullptr; hipEvent_t ev_c1 = nullptr, ev_c2 = nullptr, ev_c3 = nullptr;
  size_t bucket1_off = 0, bucket1_count = 0;   // gradients of the output / label heads: ready first, reduced early
  bool bucket1_in_flight = false;
  bool fold_dz_now = false;      // this backward pass: the d z product + latent backward run inside the encoder's BatchNorm-backward launch (smx_step.hip)
  bool chain_started = false;    // this step's head bucket went: all-reduce -> norms -> clip + Adam sweep on the communication stream (smx_step.hip: dp_chain_start)
  // graphs
  std::map<int, hipGraphExec_t> graphs;
  bool capturing = false;
  bool graph_comm_failed = false;
  // timing
  std::string timing_label; int timing_reps = SMX_LOSS_TIMING_REPEAT; std::vector<std::pair<hipEvent_t, hipEvent_t>> timing_events; size_t timing_used = 0;
};

namespace smx {

struct Timed {
  smx_model* m; hipEvent_t stop = nullptr;
  Timed(smx_model* m_, const char* label) : m(m_) {
    if (m->capturing || m->timing_label.empty() || m->timing_label != label) return;
    if (m->timing_used == m->timing_events.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      m->timing_events.push_back({a, b});
    }
    auto& ev = m->timing_events[m->timing_used++];
    hipEventRecord(ev.first, m->st);
    stop = ev.second;
  }
  ~Timed() { if (stop) hipEventRecord(stop, m->st); }
};

template <typename T>
int dmalloc(T** p, size_t n) {
  if (n == 0) n = 1;
  hipError_t e = hipMalloc((void**)p, n * sizeof(T));
  if (e != hipSuccess) { set_error(std::string("hipMalloc failed: ") + hipGetErrorString(e)); return SMX_ERR_NOMEM; }
  e = hipMemset(*p, 0, n * sizeof(T));
  if (e != hipSuccess) { set_error(std::string("hipMemset failed: ") + hipGetErrorString(e)); return SMX_ERR_HIP; }
  return SMX_OK;
}

inline int32_t* cur_rows(smx_model* m) { return m->rows2[m->par]; }
inline StepState* cur_state(smx_model* m) { return m->state3 + m->par; }
inline StepState* master_state(smx_model* m) { return m->state3 + 2; }
inline float* P_(smx_model* m, int t) { return m->params + m->tensors[t].offset; }
inline float* G_(smx_model* m, int t) { return m->grads + m->tensors[t].offset; }

// ---- one pass description ----------------------------------------------------
struct Pass {
  int B = 0;
  const int32_t* rows = nullptr;   // device row ids into X (nullptr: identity on Xsrc)
  const int32_t* xrows = nullptr;  // ... as the readers of X see them: == rows, or nullptr when Xsrc already holds the minibatch's rows (sparse store)
  const float* Xsrc = nullptr;     // m->X, m->hostX, or the expanded minibatch of the sparse store
  int x_u16 = 0;                   // Xsrc is the compact uint16 store (resident rows only)
  const float* lib = nullptr;      // library [..][2] matching Xsrc indexing
  const float* lgx1 = nullptr;
  uint32_t cell_base = 0;
  int training = 1;
  int sample = 0;
  int global_batch = 0;
};

// smx_model.hip
int add_tensor(smx_model* m, const std::string& name, int rows, int cols, int chunks, bool vec);
int build_mlp(smx_model* m, std::vector<MlpLayer>& mlp, const char* prefix, int n_in, int n, const int32_t* units,
              int stream0, float drop_p, bool batchnorm, float leak = 0.f);
void release_csr(smx_model* m);
NoiseKey make_key(smx_model* m, int stream, int sample, bool training);
const Injected* inj(smx_model* m, int stream);
void pack(const TensorInfo& t, const float* host, std::vector<float>& dev);
void unpack(const TensorInfo& t, const std::vector<float>& dev, float* host, float scale);
void drop_graphs(smx_model* m);
// smx_comm.hip
bool dp_active(const smx_model* m);
bool dp_shard_available(const smx_model* m);   // flag opt_shard can be honoured: the loopback communicator, or RCCL with ncclReduceScatter / ncclAllGather
int dp_reduce_scatter(smx_model* m, float* buf, size_t slice, hipStream_t st, bool second = false);   // in place: rank r's sum lands in buf + r * slice
int dp_all_gather(smx_model* m, float* buf, size_t slice, hipStream_t st, bool second = false);       // in place: rank r contributes buf + r * slice
bool dp_overlap(const smx_model* m);
int dp_allreduce_buf(smx_model* m, float* buf, size_t count, hipStream_t st, bool second = false);   // second: the heads' bucket (its own communicator / scratch)
int dp_allreduce(smx_model* m, size_t off, size_t count, hipStream_t st, bool second = false);
bool dp_chain_ok(const smx_model* m);
// smx_step.hip
// mode: 0 full forward; 1 decoder only (z given in m->z); 2 resample (encoder outputs m->latbuf / m->latlbuf kept,
// only the latent draw and everything after it run again); 3 encoders + latent moments only
int forward_pass(smx_model* m, const Pass& ps, bool with_loss, bool backward, int mode = 0);
int backward_pass(smx_model* m, const Pass& ps);
int optimizer_pass(smx_model* m);
int csr_stage(smx_model* m, Pass& ps);
int check_rows(smx_model* m, const int32_t* ids, size_t n);
int read_metrics(smx_model* m, smx_metrics* out);
int setup_pass(smx_model* m, Pass& ps, const int32_t* row_ids, const float* host_x, const float* host_library,
               int32_t batch, int training, int sample);
// smx_predict.hip
bool stacked_scoring_ok(const smx_model* m);
bool head_fused_ok(const smx_model* m, int B);   // a training step of B cells takes the one-launch output head (smx_step.hip)
int stacked_decoder(smx_model* m, const float* z, long rows, float* const* hb, int last_form, float* ht, const float** out, int* out_ld);

}  // namespace smx
