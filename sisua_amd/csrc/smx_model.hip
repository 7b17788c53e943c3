// smx_model.hip -- model state, step orchestration and the C-ABI of include/sisua_hip.h.
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

namespace smx {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace smx
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
enum { ST_INPUT_DROPOUT = 0, ST_ENC_DROPOUT = 16, ST_ENCL_DROPOUT = 32, ST_DEC_DROPOUT = 48, ST_EPS_Z = 64, ST_EPS_L = 65 };

namespace {

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
  float* noise = nullptr;   // [Bmax][out_p] dropout multipliers produced ahead of the layer (fused small-layer path)
};

struct Injected { float* d = nullptr; int ld = 0; };

struct RcclApi {
  void* lib = nullptr;
  std::string path, hip_path;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

std::string path_of_symbol(const void* sym) {
  Dl_info info;
  if (sym && dladdr(sym, &info) && info.dli_fname) {
    char real[PATH_MAX];
    return realpath(info.dli_fname, real) ? std::string(real) : std::string(info.dli_fname);
  }
  return "";
}

// RCCL is resolved DETERMINISTICALLY as the sibling of the HIP runtime this process actually runs on: a process
// holds exactly one libamdhip64.so.7 (ROCm's, or the copy bundled with torch when torch was imported first -- same
// soname), and the communication library must have been built against that one.  A bare dlopen("librccl.so.1")
// would return whichever copy happens to be mapped already.  SMX_RCCL_PATH overrides; smx_comm_library() reports.
int load_rccl() {
  if (g_rccl.lib) return SMX_OK;
  g_rccl.hip_path = path_of_symbol((const void*)&hipGetDeviceCount);
  std::vector<std::string> cands;
  if (const char* e = getenv("SMX_RCCL_PATH")) cands.push_back(e);
  const size_t slash = g_rccl.hip_path.rfind('/');
  if (slash != std::string::npos) {
    const std::string dir = g_rccl.hip_path.substr(0, slash + 1);
    cands.push_back(dir + "librccl.so.1");
    cands.push_back(dir + "librccl.so");
  }
  cands.push_back("librccl.so.1");
  cands.push_back("librccl.so");
  void* h = nullptr;
  std::string why;
  for (const std::string& c : cands) {
    if (c.find('/') != std::string::npos && access(c.c_str(), R_OK) != 0) continue;
    h = dlopen(c.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
    why = dlerror();
  }
  if (!h) { set_error("cannot load librccl (looked beside " + g_rccl.hip_path + "): " + why); return SMX_ERR_COMM; }
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_rccl.GetVersion = (decltype(g_rccl.GetVersion))dlsym(h, "ncclGetVersion");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
    set_error("librccl lacks a required symbol");
    dlclose(h);
    return SMX_ERR_COMM;
  }
  g_rccl.path = path_of_symbol((const void*)g_rccl.AllReduce);
  // the bound RCCL must sit on the SAME HIP runtime as this library: two runtimes in one process do not share
  // streams.  RCCL's own libamdhip64 dependency resolves by soname to the mapped copy, so it suffices that
  // only one copy is mapped -- checked by asking the dynamic loader where RCCL's hipMalloc would come from.
  if (void* sym = dlsym(h, "hipGetDeviceCount")) {   // found through RCCL's dependency chain
    const std::string theirs = path_of_symbol(sym);
    if (!theirs.empty() && !g_rccl.hip_path.empty() && theirs != g_rccl.hip_path) {
      set_error("librccl (" + g_rccl.path + ") runs on " + theirs + " but libsisua_hip on " + g_rccl.hip_path);
      dlclose(h);
      return SMX_ERR_COMM;
    }
  }
  g_rccl.lib = h;
  return SMX_OK;
}

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

__global__ void bn_moving_update_kernel(float* moving, const float* batch_sum, int n, float inv_world, float momentum) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) moving[i] = moving[i] * momentum + batch_sum[i] * inv_world * (1.f - momentum);
}

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
struct LocalSrc { const float* p[SMX_LOCAL_MAX]; int n; };
__global__ void local_sum_kernel(LocalSrc s, float* dst, size_t count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    float acc = s.p[0][i];
    for (int r = 1; r < s.n; ++r) acc += s.p[r][i];
    dst[i] = acc;
  }
}

}  // namespace

struct smx_model {
  smx_config cfg;
  int device = 0;
  hipStream_t st = nullptr, st2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_fork2 = nullptr, ev_join = nullptr;
  bool forked = false;
  bool head_fused = false;
  bool head_loss = false;             // this step's output product ran fused with the likelihood (smx_headloss.hip)
  // code-path switches (smx_set_flag; defaults from the SMX_NO_* environment variables): 1 = the default wide / fused
  // kernels, 0 = the separate-launch forms they replaced (kept for eval, for shapes the fused kernels do not take, and
  // as A/B references)
  struct Flags {
    int head_loss = getenv("SMX_NO_HEAD_LOSS") ? 0 : 1;    // output product + likelihood in one kernel
    int front = getenv("SMX_NO_FRONT") ? 0 : 1;            // latent sample + first decoder product inside BatchNorm-forward
    int bwd_front = getenv("SMX_NO_BWD_FRONT") ? 0 : 1;    // d h inside BatchNorm-backward, weight gradients grouped at the end
    int head_bwd = getenv("SMX_NO_HEAD_BWD") ? 0 : 1;      // both backward products of the output head in one wide launch
    int wgrad = getenv("SMX_NO_WGRAD") ? 0 : 1;            // K = minibatch weight gradients as the wide direct-operand kernel
    int scvi_fused = getenv("SMX_NO_SCVI_FUSED") ? 0 : 1;  // scvi: library latent + softmax head + likelihood + their backward as one row-local launch
    int twin = getenv("SMX_NO_TWIN") ? 0 : 1;              // scvi: first layers of both encoders (and pairs of heads) side by side in one launch
    int act_epilogue = getenv("SMX_NO_ACT_EPILOGUE") ? 0 : 1;  // layers without BatchNorm / dropout: bias + activation (and its derivative) in the products' store paths
    int label_ride = getenv("SMX_NO_LABEL_RIDE") ? 0 : 1;  // label heads' backward inside the output head's backward launch + the final grouped launch
    int stacked_scoring = getenv("SMX_NO_STACKED_SCORING") ? 0 : 1;  // marginal_llk: all posterior draws as rows of ONE decoder pass
  } flags;
  int chunk_first_head = 0;           // first optimiser chunk of the output / label heads (they are last in the table)
  int chunk_first_label = 0;          // first optimiser chunk of the label heads (n_chunks without label heads)
  bool lab_deferred = false;          // this step's label-head weight gradients come with the grouped launch at the END of backward
  int adam_early_to = -1;             // chunks [adam_early_from, adam_early_to) of this step were applied early
  bool adam_early_pending = false;    // the heads' gradients are final: the next BatchNorm-backward launch may carry their update
  int adam_early_from = -1;           // >= 0: chunks [adam_early_from, n_chunks) of this step were applied early
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
  float* sq_slots = nullptr; std::vector<int> sq_first, sq_count; std::vector<char> sq_reduced; int sq_total_first = 0;   // this step's output head ran as the fused kernel (smx_head.hip)
  int G = 0, Gp = 0, D = 0, Dp = 0, k = 0, Bmax = 0;
  bool stochastic = true, scvi = false, scale = false, fvae = false;
  int n_heads = 0;                    // label heads on the decoder (0 for fvae: SemiFVAE's labels go to the discriminator)
  // fvae: discriminator on z (smx_factor.hip)
  std::vector<MlpLayer> disc; int t_discoutW = -1, t_discoutb = -1;
  float *zz = nullptr, *u_tc = nullptr, *u_d = nullptr, *tc_cell = nullptr, *dl_cell = nullptr, *dz_tc = nullptr;
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
  float* mhist = nullptr; size_t mhist_cap = 0; int32_t mhist_steps = 0;   // ELBO scalars of every step of the last train_steps call
  StepState* state3 = nullptr;  // [0],[1]: per-step state by parity, [2]: master counter
  int par = 0; uint32_t h_next = 0;
  MetricsArgs pending_metrics; bool have_pending_metrics = false, metrics_before_allreduce = false;
  int seq_batch = 0, seq_prepare_next = 0;
  bool eps_ahead_ok = false;   // latent head fusable: eps may be drawn ahead by the first BN launch
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
  float* slab = nullptr; size_t slab_cap = 0; int max_feat_p = 0;
  // optimiser
  OptChunk* chunks = nullptr; int n_chunks = 0; int chunks_floats = 4096; float* partial = nullptr; float* tensor_norm = nullptr;
  // noise injection
  std::map<int, Injected> injected; bool use_injected = false;
  // comm
  ncclComm_t comm = nullptr; int rank = 0, world = 1;
  std::shared_ptr<LocalGroup> local; float* local_scratch = nullptr; size_t local_scratch_cap = 0;   // loopback communicator (tests)
  // SyncBatchNorm (opt-in, smx_comm_set_sync_bn): per BN launch one small all-reduce of per-rank column statistics
  bool sync_bn = false; float* sync_buf = nullptr; size_t sync_cap = 0;
  bool dp_force = false, dp_two_buckets = false;   // SMX_FORCE_ALLREDUCE / SMX_DP_BUCKETS=2, read when the communicator is attached
  hipStream_t st_comm = nullptr; hipEvent_t ev_c1 = nullptr, ev_c2 = nullptr, ev_c3 = nullptr;
  size_t bucket1_off = 0, bucket1_count = 0;   // gradients of the output / label heads: ready first, reduced early
  bool bucket1_in_flight = false;
  // graphs
  std::map<int, hipGraphExec_t> graphs;
  bool capturing = false;
  bool graph_comm_failed = false;
  // timing
  std::string timing_label; std::vector<std::pair<hipEvent_t, hipEvent_t>> timing_events; size_t timing_used = 0;
};

namespace {

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
              int stream0, float drop_p, bool batchnorm, float leak = 0.f) {
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

template <typename T>
int dmalloc(T** p, size_t n) {
  if (n == 0) n = 1;
  hipError_t e = hipMalloc((void**)p, n * sizeof(T));
  if (e != hipSuccess) { set_error(std::string("hipMalloc failed: ") + hipGetErrorString(e)); return SMX_ERR_NOMEM; }
  e = hipMemset(*p, 0, n * sizeof(T));
  if (e != hipSuccess) { set_error(std::string("hipMemset failed: ") + hipGetErrorString(e)); return SMX_ERR_HIP; }
  return SMX_OK;
}

int32_t* cur_rows(smx_model* m) { return m->rows2[m->par]; }
StepState* cur_state(smx_model* m) { return m->state3 + m->par; }
StepState* master_state(smx_model* m) { return m->state3 + 2; }
float* P_(smx_model* m, int t) { return m->params + m->tensors[t].offset; }
float* G_(smx_model* m, int t) { return m->grads + m->tensors[t].offset; }

// the sparse store's arrays (m->X aliases the expansion tile while it is in use)
static void release_csr(smx_model* m) {
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

void fill_adam_args(smx_model* m, AdamArgs& a);
bool dp_active(const smx_model* m);
int dp_allreduce_buf(smx_model* m, float* buf, size_t count, hipStream_t st);
// SyncBatchNorm applies to training passes of a data-parallel job only (eval mode uses the moving statistics)
bool sync_bn_on(const smx_model* m, int training) { return m->sync_bn && training && m->cfg.batchnorm && dp_active(m); }
BnSyncArgs sync_args(smx_model* m) { BnSyncArgs y; y.gather = m->sync_buf; y.rank = m->rank; y.world = m->world; return y; }

// shapes / modes under which the decoder's first BatchNorm launch takes the latent sample and its product along
// (forward_pass adds what depends on injected noise)
bool use_mid(const smx_model* m, int B);
static bool front_shapes_ok(smx_model* m, const Pass& ps) {
  if (use_mid(m, ps.B)) return false;
  const int lat_ld = m->stochastic ? 2 * m->Dp : m->Dp;
  static const bool no_fz = getenv("SMX_SMALL_FUSION") == nullptr;
  const bool fuse_lat = !no_fz && !m->scale && latent_head_fusable(m->enc.back().out_p, lat_ld, m->Dp);
  return m->flags.front && !m->scale && !fuse_lat && !sync_bn_on(m, ps.training) && bn_front_supported(ps.B, m->Dp) &&
         (m->Dp == 32 || m->Dp == 64) && m->dec[0].in_p == m->Dp && m->dec[0].out_p % 8 == 0 && (lat_ld % 4) == 0;
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
    // measured: the 4-workgroup fused small-layer kernels are 3 us/step SLOWER than two wider launches;
    // opt-in (SMX_SMALL_FUSION=1) and covered by tests/test_gpu_variants.py
    static const bool no_fz = getenv("SMX_SMALL_FUSION") == nullptr;
    static const bool no_ahead = getenv("SMX_NO_NOISE_AHEAD") != nullptr;
    const bool no_twin = !m->flags.twin;
    const bool ahead = !no_fz && !no_ahead && !m->scvi;
    const bool sync = sync_bn_on(m, ps.training) && L.bn >= 0;
    // hidden -> hidden layers 32 / 64 / 128 wide: the BatchNorm launch stages the layer's INPUT tile [B][K] in LDS and forms its
    // own columns as dot products -- the same front the first decoder layer uses for the latent sample, here as a plain
    // copy (no product launch; the reference's default networks are [64, 64], configs/base.yaml:10-17)
    LatentArgs dense_la;
    const bool dense_front = m->flags.front && no_fz && !sync && !(front != nullptr && i == 0) && !(i == 0 && in_is_x) && L.leak == 0.f &&
                             (L.in_p == 32 || L.in_p == 64 || (L.in_p == 128 && ps.B <= 128)) && bn_front_supported(ps.B, L.in_p) && L.out_p % 8 == 0 &&
                             (ld % 4) == 0 && !use_mid(m, ps.B);
    if (dense_front) {
      dense_la.stochastic = 0; dense_la.relu = 0; dense_la.training = ps.training;
      dense_la.lat = in; dense_la.ld = ld; dense_la.B = ps.B; dense_la.D = L.in; dense_la.Dp = L.in_p;
    }
    const LatentArgs* front_i = (front != nullptr && i == 0) ? front : (dense_front ? &dense_la : nullptr);
    const bool with_front = front_i != nullptr;   // the BatchNorm launch produces its own input (latent sample / input tile + product)
    const bool fuse = !no_fz && !sync && !with_front && !(i == 0 && in_is_x) && L.leak == 0.f && dense_bn_fusable(ps.B, L.in_p);
    int eff = 1;
    SMX_REQUIRE((size_t)std::max(g.split_k, 1) * (size_t)g.slab_stride <= m->slab_cap, "split-K slabs exceed the slab buffer");
    // ---- the twin's first layer beside this one: one product launch, one BatchNorm launch ----
    bool dual = false;
    GemmArgs g2;
    if (twin && i == 0 && !fuse && !with_front && !sync && !no_twin && no_fz && in_is_x && !twin->empty() && bn_dual_supported(ps.B) &&
        !(ps.training && m->cfg.input_dropout > 0.f) && (*twin)[0].in_p == L.in_p && L.leak == 0.f && (*twin)[0].leak == 0.f) {
      MlpLayer& T = (*twin)[0];
      float* slab2 = m->slab + (size_t)std::max(g.split_k, 1) * (size_t)g.slab_stride;
      g2 = make_gemm(T, in, ld, true, slab2);
      dual = ((size_t)std::max(g.split_k, 1) * ((size_t)g.slab_stride + (size_t)g2.slab_stride) <= m->slab_cap);
    }
    // layers without BatchNorm and without dropout (the FactorVAE discriminator; plain autoencoders at evaluation): bias
    // and activation in the product's own store path -- no bias / activation launch
    const bool epi_act = m->flags.act_epilogue && !dual && !fuse && !with_front && !sync && L.bn < 0 && !(ps.training && L.drop_p > 0.f) &&
                         g.split_k <= 1 && !m->use_injected;
    if (epi_act) {
      g.bias = P_(m, L.tBias); g.act = 1; g.leak = L.leak; g.C = L.out_buf; g.ldc = L.out_p; g.split_k = 1;
      Timed t(m, (i == 0 && in_is_x) ? label0 : "gemm_mlp_fwd");
      SMX_CHECK(launch_gemm(m->st, g));
      in = L.out_buf; ld = L.out_p;
      continue;
    }
    if (dual) {
      Timed t(m, label0);
      SMX_CHECK(launch_gemm_dual(m->st, g, g2, &eff));
    } else if (!fuse && !with_front) {
      Timed t(m, (i == 0 && in_is_x) ? label0 : "gemm_mlp_fwd");
      SMX_CHECK(launch_gemm(m->st, g, &eff));
    }
    BnFwdArgs b = make_bn(L, m->slab, eff, g.slab_stride);
    if (fuse && ahead && !b.inj_mask && b.drop_p > 0.f) { b.inj_mask = L.noise; b.inj_ld = L.out_p; }
    if (!fuse && ahead && i == 0 && in_is_x && &mlp == &m->enc) {
      // this launch precedes every fused small layer of the step: draw their noise on otherwise idle CUs
      auto add = [&](float* dst, int ld, int width, int normal, float p, int stream) {
        if (b.n_jobs < SMX_NOISE_JOBS) {
          NoiseJob& j = b.jobs[b.n_jobs++];
          j.dst = dst; j.ld = ld; j.width = width; j.normal = normal; j.p = p;
          j.stream = (uint32_t)((stream & 0xFF) | ((ps.sample & 0xFFFFFF) << 8));
        }
      };
      if (ps.training) {
        for (size_t q = 1; q < m->enc.size(); ++q)
          if (m->enc[q].drop_p > 0.f && dense_bn_fusable(ps.B, m->enc[q].in_p) && !inj(m, m->enc[q].stream))
            add(m->enc[q].noise, m->enc[q].out_p, m->enc[q].out, 0, m->enc[q].drop_p, m->enc[q].stream);
        for (size_t q = 0; q < m->dec.size(); ++q)
          if (m->dec[q].drop_p > 0.f && dense_bn_fusable(ps.B, m->dec[q].in_p) && !inj(m, m->dec[q].stream))
            add(m->dec[q].noise, m->dec[q].out_p, m->dec[q].out, 0, m->dec[q].drop_p, m->dec[q].stream);
      }
      if (m->stochastic && !inj(m, ST_EPS_Z) && m->eps_ahead_ok) add(m->noise_eps, m->Dp, m->D, 1, 0.f, ST_EPS_Z);
      if (b.n_jobs) { b.nk.step_ptr = ps.training ? &cur_state(m)->step : nullptr; }
    }
    if (!fuse && !ahead && !no_ahead && !sync && i == 0 && in_is_x && &mlp == &m->enc && ps.training && front_shapes_ok(m, ps) &&
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
    } else if (fuse) {
      Timed t(m, "dense_bn_fwd");
      SMX_CHECK(launch_dense_bn_act_fwd(m->st, in, ld, L.in_p, P_(m, L.tW), tw.ld, b));
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
void attach_early_adam(smx_model* m, BnBwdArgs& b) {
  if (!m->adam_early_pending) return;
  m->adam_early_pending = false;
  static const bool off = getenv("SMX_NO_ADAM_EARLY") != nullptr;
  if (off || dp_active(m) || !m->sq_slots || m->chunk_first_head >= m->n_chunks || getenv("SMX_NO_SQ_PARTIALS") != nullptr) return;
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
    return;
  }
  fill_adam_args(m, b.adam);
  b.adam.use_sq = 1;
  for (size_t t = 0; t < m->tensors.size(); ++t) { b.adam.sq_first[t] = m->sq_first[t]; b.adam.sq_count[t] = m->sq_count[t]; }
  b.adam.master = nullptr; b.adam.with_metrics = 0;
  // (label heads whose weight gradients come with the grouped launch at the END of the backward pass stay with the
  // optimiser launch)
  const int early_to = m->lab_deferred ? m->chunk_first_label : m->n_chunks;
  b.adam_first = m->chunk_first_head;
  b.adam_count = early_to - m->chunk_first_head;
  m->adam_early_from = m->chunk_first_head; m->adam_early_to = early_to;
}

// ask the product that writes the gradient of tensor t for sum-of-squares partials
void want_sq(smx_model* m, GemmArgs& g, int t) {
  if (!m->sq_slots || getenv("SMX_NO_SQ_PARTIALS") != nullptr) return;   // read per call: tests toggle it
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
    if (front) { b.front = 1; b.fD = front->fD; b.fld = front->fld; b.fW = front->fW; b.fldw = front->fldw; b.fK = front->fK; }
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
    if (defer && i == 0 && lat_epi) {   // d z (+ latent-head backward) alone; d W joins the final grouped launch
      defer->push_back(g);
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

// ---- side stream: work that only the optimiser waits for (metrics, weight gradients of the
// output head and of the middle layers) leaves the critical path ------------------------------
bool side_ok(smx_model* m) { return m->st2 != nullptr && m->timing_label.empty(); }
hipStream_t side_stream(smx_model* m) { return (m->forked && side_ok(m)) ? m->st2 : m->st; }
int side_fork(smx_model* m, hipEvent_t ev) {
  if (!side_ok(m)) return SMX_OK;
  SMX_HIP(hipEventRecord(ev, m->st));
  SMX_HIP(hipStreamWaitEvent(m->st2, ev, 0));
  m->forked = true;
  return SMX_OK;
}
int side_join(smx_model* m) {
  if (!m->forked) return SMX_OK;
  SMX_HIP(hipEventRecord(m->ev_join, m->st2));
  SMX_HIP(hipStreamWaitEvent(m->st, m->ev_join, 0));
  m->forked = false;
  return SMX_OK;
}

// data-parallel overlap: two buckets on a communication stream (eager launches only)
bool dp_active(const smx_model* m) {
  // dp_force: exercise RCCL on a 1-rank communicator (tests); local: the loopback communicator of the tests
  return (m->comm && (m->world > 1 || m->dp_force)) || (m->local && m->world > 1);
}
// Measured on a 1-rank communicator: the cross-stream events of the two-bucket form cost +42 us per step,
// one all-reduce on the model's own stream +2.6 us.  The overlap only pays when the collective itself is
// much longer than that, so the default is the single all-reduce; SMX_DP_BUCKETS=2 selects the overlap.
bool dp_overlap(const smx_model* m) {
  return dp_active(m) && m->dp_two_buckets && !m->capturing && m->st_comm != nullptr && !m->local;
}
int local_allreduce(smx_model* m, float* buf, size_t count, hipStream_t st) {
  LocalGroup& g = *m->local;
  const int me = m->rank;
  SMX_REQUIRE(count <= m->local_scratch_cap, "loopback all-reduce: scratch too small");
  { std::lock_guard<std::mutex> lk(g.mu); g.src[me] = buf; }
  SMX_HIP(hipEventRecord(g.ready[me], st));
  if (!g.barrier()) { set_error("loopback communicator: a member did not arrive (timeout)"); return SMX_ERR_COMM; }
  LocalSrc src;
  src.n = g.world;
  for (int r = 0; r < g.world; ++r) {
    src.p[r] = g.src[r];
    if (r != me) SMX_HIP(hipStreamWaitEvent(st, g.ready[r], 0));
  }
  const unsigned blocks = (unsigned)std::min<size_t>((count + 255) / 256, 2048);
  hipLaunchKernelGGL(local_sum_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, m->local_scratch, count);
  SMX_HIP(hipEventRecord(g.done[me], st));
  if (!g.barrier()) { set_error("loopback communicator: a member did not arrive (timeout)"); return SMX_ERR_COMM; }
  for (int r = 0; r < g.world; ++r)
    if (r != me) SMX_HIP(hipStreamWaitEvent(st, g.done[r], 0));   // nobody still reads this rank's buffer
  SMX_HIP(hipMemcpyAsync(buf, m->local_scratch, count * sizeof(float), hipMemcpyDeviceToDevice, st));
  return SMX_OK;
}
int dp_allreduce_buf(smx_model* m, float* buf, size_t count, hipStream_t st) {
  if (m->local) return local_allreduce(m, buf, count, st);
  ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclFloat32, ncclSum, m->comm, st);
  if (r != ncclSuccess) {
    set_error(std::string("ncclAllReduce failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
    return SMX_ERR_COMM;
  }
  return SMX_OK;
}
int dp_allreduce(smx_model* m, size_t off, size_t count, hipStream_t st) { return dp_allreduce_buf(m, m->grads + off, count, st); }

// fused output head (product + likelihood + dW/db in one kernel): count heads with raw parameter planes.
// Opt-in (SMX_OUT_FUSED=1): parity-green but not faster -- 16 genes x whole batch per workgroup gives only
// Gp/16 = 126 workgroups at 8kly width, so the two MFMA phases (3.5 + 3.2 us) and the likelihood (6.1 us, four
// elements per lane at two waves per SIMD) run on half the chip: 20.2 us against 8.0 + 5.5 + ~8 us for the
// separate product / loss / weight-gradient launches (step 125 vs 123 us).
bool use_fused_head(const smx_model* m, int B) {
  static const bool on = getenv("SMX_OUT_FUSED") != nullptr && atoi(getenv("SMX_OUT_FUSED")) != 0;
  if (!on || m->scvi || m->dec.empty()) return false;
  return out_head_supported(B, m->dec.back().out_p, m->Gp);
}

// Training steps of the count heads with raw parameter planes (VAE / DCA / SISUA): output product + likelihood +
// dP in ONE wide kernel, P never materialised (smx_headloss.hip).  SMX_NO_HEAD_LOSS=1 keeps the product / loss
// kernel pair (what eval, predict and the scoring paths always use).
bool use_head_loss(const smx_model* m, int B) {
  if (!m->flags.head_loss || m->scvi || m->dec.empty()) return false;
  return head_loss_supported(B, m->dec.back().out_p, m->Gp);
}

bool use_mid(const smx_model* m, int B) {
  if (m->x_csr) return false;   // (the single-workgroup middle reads X through the resident row ids)
  // single-workgroup fusion of the middle is opt-in until it beats the per-operator path
  static const bool off = getenv("SMX_FUSED") == nullptr;
  if (off || m->scvi || m->scale || m->fvae || B > 128 || m->sync_bn) return false;
  for (auto* mlp : {&m->enc, &m->dec})
    for (auto& L : *mlp) if (L.out_p > 128) return false;
  if ((m->stochastic ? 2 : 1) * m->Dp > 128) return false;
  return true;
}

void fill_mid_layer(smx_model* m, const MlpLayer& L, MidLayer& o, bool training) {
  memset(&o, 0, sizeof(o));
  const TensorInfo& tw = m->tensors[L.tW];
  o.W = P_(m, L.tW); o.ldw = tw.ld;
  if (m->cfg.batchnorm) {
    o.gamma = P_(m, L.tGamma); o.beta = P_(m, L.tBeta);
    o.moving_mean = m->bn_moving + m->bn_off[L.bn]; o.moving_var = o.moving_mean + L.out_p;
    o.batch_mean = m->grads + m->tail_off_bn + m->bn_off[L.bn]; o.batch_var = o.batch_mean + L.out_p;
    o.dgamma = G_(m, L.tGamma); o.dbeta = G_(m, L.tBeta);
  } else {
    o.bias = P_(m, L.tBias); o.dbias = G_(m, L.tBias);
  }
  o.xhat = L.xhat; o.outb = L.out_buf; o.inv_std = L.inv_std; o.dpre = L.dpre;
  o.in_p = L.in_p; o.out = L.out; o.out_p = L.out_p;
  o.drop_p = training ? L.drop_p : 0.f; o.stream = (uint32_t)L.stream;
  if (const Injected* ij = inj(m, L.stream)) { o.inj_mask = ij->d; o.inj_ld = ij->ld; }
}

void fill_mid_args(smx_model* m, const Pass& ps, MidArgs& a) {
  memset(&a, 0, sizeof(a));
  const smx_config& c = m->cfg;
  a.B = ps.B; a.batchnorm = c.batchnorm; a.training = ps.training; a.update_moving = (m->world == 1);
  a.momentum = c.bn_momentum; a.eps = c.bn_eps;
  const MlpLayer& e0 = m->enc[0];
  a.h0 = e0.out_buf; a.h0_w = e0.out_p;
  a.n_enc = (int)m->enc.size() - 1;
  for (int i = 0; i < a.n_enc; ++i) fill_mid_layer(m, m->enc[i + 1], a.enc[i], ps.training != 0);
  const TensorInfo& tl = m->tensors[m->t_latW];
  a.Wlat = P_(m, m->t_latW); a.ld_wlat = tl.ld; a.blat = P_(m, m->t_latb);
  a.lat_in_p = m->enc.back().out_p; a.lat_ld = (m->stochastic ? 2 : 1) * m->Dp; a.D = m->D; a.Dp = m->Dp;
  a.stochastic = m->stochastic; a.relu = (c.latent_activation == SMX_ACT_RELU);
  a.latbuf = m->latbuf; a.z = m->z; a.sig = m->sig; a.eps_out = m->eps; a.kl = m->kl;
  a.n_dec = (int)m->dec.size();
  for (int i = 0; i < a.n_dec; ++i) fill_mid_layer(m, m->dec[i], a.dec[i], ps.training != 0);
  a.k0 = (uint32_t)(c.seed & 0xFFFFFFFFu); a.k1 = (uint32_t)(c.seed >> 32);
  a.step_ptr = ps.training ? &cur_state(m)->step : nullptr; a.step = 0; a.sample = (uint32_t)ps.sample;
  a.rows = ps.rows; a.cell_base = ps.cell_base;
  if (const Injected* ij = inj(m, ST_EPS_Z)) { a.inj_eps = ij->d; a.inj_eps_ld = ij->ld; }
  a.kl_scale = c.beta / (float)ps.global_batch; a.dlat = m->dlat;
  a.dpre_enc0 = e0.dpre; a.enc0_out = e0.out_buf; a.enc0_xhat = e0.xhat; a.enc0_inv_std = e0.inv_std;
  a.enc0_out_w = e0.out; a.enc0_out_p = e0.out_p; a.enc0_drop_p = ps.training ? e0.drop_p : 0.f;
  if (c.batchnorm) { a.enc0_gamma = P_(m, e0.tGamma); a.enc0_dgamma = G_(m, e0.tGamma); a.enc0_dbeta = G_(m, e0.tBeta); }
  else a.enc0_dbias = G_(m, e0.tBias);
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
int forward_pass(smx_model* m, const Pass& ps, bool with_loss, bool backward, int mode = 0) {
  const bool decode_only = (mode == 1), resample = (mode == 2);
  const bool encode_only = (mode == 3);   // encoders + latent heads + latent moments / draw 0, no decoder (the stacked-draw paths)
  const smx_config& c = m->cfg;
  const float inv_gb = 1.f / (float)ps.global_batch;
  const bool mid = (mode == 0) && use_mid(m, ps.B);
  m->head_loss = false;
  m->ahead_front_eps = m->ahead_front_drop = false;
  m->scvi_fused = false; m->encl_twinned = false;
  bool front_ok = false; LatentArgs front_la;
  if (mid) {
    SMX_CHECK(mlp_forward(m, m->enc, ps, ps.Xsrc, m->Gp, true, "gemm_enc_fwd", 1));
    MidArgs ma;
    fill_mid_args(m, ps, ma);
    ma.dbg = nullptr;
    Timed t(m, "mid_fwd");
    SMX_CHECK(launch_mid_fwd(m->st, ma));
  }
  if (!decode_only && !mid) {
  // ---- encoder ----
  bool twin_done = false;
  if (!resample) SMX_CHECK(mlp_forward(m, m->enc, ps, ps.Xsrc, m->Gp, true, "gemm_enc_fwd", -1, nullptr, 0, m->scvi ? &m->encl : nullptr, &twin_done));
  m->encl_twinned = twin_done;
  const MlpLayer& eL = m->enc.back();
  const int lat_ld = m->stochastic ? 2 * m->Dp : m->Dp;
  static const bool no_fz = getenv("SMX_SMALL_FUSION") == nullptr;
  const bool fuse_lat = !no_fz && !m->scale && !resample && latent_head_fusable(eL.out_p, lat_ld, m->Dp);
  if (!fuse_lat && !resample) {
    const TensorInfo& tw = m->tensors[m->t_latW];
    GemmArgs g;
    g.A = eL.out_buf; g.lda = eL.out_p; g.B = P_(m, m->t_latW); g.ldb = tw.ld;
    g.C = m->latbuf; g.ldc = lat_ld; g.M = ps.B; g.N = lat_ld; g.K = eL.out_p; g.bias = P_(m, m->t_latb);
    Timed t(m, "gemm_lat_fwd");
    SMX_CHECK(launch_gemm(m->st, g));
  }
  LatentArgs la;
  la.stochastic = m->stochastic; la.relu = (c.latent_activation == SMX_ACT_RELU); la.training = ps.training;
  la.lat = m->latbuf; la.ld = lat_ld; la.B = ps.B; la.D = m->D; la.Dp = m->Dp;
  la.nk = make_key(m, ST_EPS_Z, ps.sample, ps.training != 0);
  la.rows = ps.rows; la.cell_base = ps.cell_base;
  if (const Injected* ij = inj(m, ST_EPS_Z)) { la.inj_eps = ij->d; la.inj_ld = ij->ld; }
  static const bool no_ahead = getenv("SMX_NO_NOISE_AHEAD") != nullptr;
  if (fuse_lat && m->stochastic && !la.inj_eps && !no_ahead && !m->scvi && m->eps_ahead_ok) { la.inj_eps = m->noise_eps; la.inj_ld = m->Dp; }
  if (m->ahead_front_eps && !la.inj_eps) { la.inj_eps = m->noise_eps; la.inj_ld = m->Dp; }
  la.z = m->z; la.sig = m->sig; la.eps = m->eps; la.kl = m->kl;
  // The latent sample + KL and the first decoder product run INSIDE the decoder's first BatchNorm launch (two
  // launches fewer) when the shapes allow; SMX_NO_FRONT=1 keeps the three-launch form.
  front_ok = !encode_only && front_shapes_ok(m, ps) && (!la.inj_eps || (la.inj_ld % 4) == 0);
  front_la = la;
  if (front_ok) {
    // (launched below with the decoder)
  } else if (fuse_lat) {
    Timed t(m, "latent_head_fwd");
    SMX_CHECK(launch_latent_head_fwd(m->st, la, eL.out_buf, eL.out_p, eL.out_p, P_(m, m->t_latW), m->tensors[m->t_latW].ld,
                                     P_(m, m->t_latb), m->latbuf));
  } else {
    Timed t(m, "latent_fwd");
    SMX_CHECK(launch_latent_fwd(m->st, la));
  }
  if (m->scale) {   // Monte-Carlo KL against the mixture prior at the z just drawn (overwrites the analytic KL)
    ScalePriorArgs sp;
    sp.z = m->z; sp.sig = m->sig; sp.eps = m->eps; sp.B = ps.B; sp.D = m->D; sp.Dp = m->Dp; sp.C = c.n_components;
    sp.logits = P_(m, m->t_prLogits); sp.loc = P_(m, m->t_prLoc); sp.scale_raw = P_(m, m->t_prScale);
    sp.kl = m->kl; sp.resp = m->resp; sp.dklz = m->dklz;
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
  if (!mid) SMX_CHECK(mlp_forward(m, m->dec, ps, m->z, m->Dp, false, "", -1, front_ok ? &front_la : nullptr));
  const MlpLayer& dL = m->dec.back();
  const long ldp = (long)m->k * m->Gp;
  if (m->scvi) {
    GemmArgs hg[3];
    for (int ch = 0; ch < m->k; ++ch) {
      const TensorInfo& tw = m->tensors[m->t_outW[ch]];
      GemmArgs& g = hg[ch];
      g.A = dL.out_buf; g.lda = dL.out_p; g.B = P_(m, m->t_outW[ch]); g.ldb = tw.ld;
      g.C = m->raw + (long)ch * m->Gp; g.ldc = (int)ldp; g.M = ps.B; g.N = m->Gp; g.K = dL.out_p;
      g.bias = P_(m, m->t_outb[ch]);
    }
    {
      // the heads read the same decoder output: pairs of them side by side in one launch
      const bool no_twin = !m->flags.twin;
      Timed t(m, "gemm_out_fwd");
      int ch = 0;
      for (; !no_twin && ch + 1 < m->k; ch += 2) SMX_CHECK(launch_gemm_dual(m->st, hg[ch], hg[ch + 1]));
      for (; ch < m->k; ++ch) SMX_CHECK(launch_gemm(m->st, hg[ch]));
    }
  }
  if (m->scvi && m->scvi_fused) {
    ScviTrainArgs st;
    scvi_train_args(m, ps, &st);
    // (timing mode: the idempotent launch repeated inside one event pair, as for the other likelihood kernels)
    const int reps = (!m->capturing && m->timing_label == "loss") ? SMX_LOSS_TIMING_REPEAT : 1;
    Timed t(m, "loss");
    for (int r = 0; r < reps; ++r) SMX_CHECK(launch_scvi_head_train(m->st, st));
  } else if (m->scvi) {
    ScviHeadArgs sh;
    sh.raw = m->raw; sh.planes = m->P; sh.ld = ldp; sh.plane_stride = m->Gp; sh.B = ps.B; sh.G = m->G; sh.Gp = m->Gp;
    sh.k = m->k; sh.l = m->lsmp; sh.clip_library = c.clip_library; sh.rho_raw = m->rho;
    SMX_CHECK(launch_scvi_head_fwd(m->st, sh));
  } else if ((m->head_loss = (with_loss && backward && !use_fused_head(m, ps.B) && use_head_loss(m, ps.B)))) {
    m->head_fused = false;   // the product runs below, fused with the likelihood
  } else if (!(m->head_fused = (with_loss && backward && use_fused_head(m, ps.B)))) {
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
    n_llk_chunks = head_loss_chunks(m->Gp);
    if (!m->capturing && m->timing_label == "out_head_product") {
      // timing mode: the product alone (P stored, no counts, no likelihood) -- what the fused kernel's time is
      // compared with to attribute the rest to the likelihood (bench.py, roofline)
      HeadLossArgs po = hl;
      po.product_only = 1; po.dP = m->P;
      Timed t(m, "out_head_product");
      for (int r = 0; r < SMX_LOSS_TIMING_REPEAT; ++r) SMX_CHECK(launch_out_head_loss(m->st, po));
    }
    const int reps = (!m->capturing && m->timing_label == "out_head") ? SMX_LOSS_TIMING_REPEAT : 1;   // idempotent
    Timed t(m, "out_head");
    for (int r = 0; r < reps; ++r) SMX_CHECK(launch_out_head_loss(m->st, hl));
  } else if (m->head_fused) {
    // training step of a count head: output product, likelihood, dP, dW_out and db_out in one kernel
    const TensorInfo& tw = m->tensors[m->t_outW[0]];
    OutHeadArgs oh;
    oh.H = dL.out_buf; oh.ldh = dL.out_p; oh.W = P_(m, m->t_outW[0]); oh.ldw = tw.ld; oh.bias = P_(m, m->t_outb[0]);
    oh.X = ps.Xsrc; oh.x_u16 = ps.x_u16; oh.ldx = m->Gp; oh.rows = ps.xrows;
    oh.dP = m->dP; oh.ldp = ldp; oh.plane_stride = m->Gp;
    oh.dW = G_(m, m->t_outW[0]); oh.db = G_(m, m->t_outb[0]);
    oh.llk_part = m->llk_part; oh.n_chunks = n_llk_chunks = out_head_chunks(m->Gp);
    oh.B = ps.B; oh.G = m->G; oh.Gp = m->Gp; oh.Hp = dL.out_p; oh.likelihood = c.likelihood; oh.grad_scale = -inv_gb;
    static const int head_diag = getenv("SMX_HEAD_DIAG") ? atoi(getenv("SMX_HEAD_DIAG")) : 0;
    oh.diag = head_diag;
    const int reps = (!m->capturing && m->timing_label == "out_head") ? SMX_LOSS_TIMING_REPEAT : 1;
    Timed t(m, "out_head");
    for (int r = 0; r < reps; ++r) SMX_CHECK(launch_out_head_train(m->st, oh));
  } else {
    // timing mode: the (idempotent) kernel is launched SMX_LOSS_TIMING_REPEAT times inside one event pair so
    // the pair's own ~5 us overhead can be separated from the per-launch time (bench.py)
    const int reps = (!m->capturing && m->timing_label == "loss") ? SMX_LOSS_TIMING_REPEAT : 1;
    Timed t(m, "loss");
    for (int r = 0; r < reps; ++r) SMX_CHECK(launch_count_loss(m->st, lo));
  }
  for (int j = 0; j < m->n_heads; ++j) {
    const TensorInfo& tw = m->tensors[m->t_labW[j]];
    LabelArgs lb;
    lb.kind = c.label_llk[j]; lb.C = c.label_components[j]; lb.raw = m->laby_raw[j]; lb.ld = tw.ld; lb.Y = m->Y[j]; lb.ldy = m->lab_Pp[j];
    lb.rows = ps.rows; lb.mask = m->mask; lb.B = ps.B; lb.P = c.label_dim[j]; lb.Pp = m->lab_Pp[j];
    lb.grad_scale = -c.alpha * inv_gb; lb.draw = m->laby_draw[j]; lb.llk = m->llk_y; lb.add = (j > 0);
    lb.backward = backward;
    SMX_CHECK(launch_label_loss(m->st, lb));
  }
  if (m->fvae) SMX_CHECK(factor_forward(m, ps, backward));
  MetricsArgs me;
  me.llk_part = m->llk_part; me.n_chunks = n_llk_chunks; me.lgx1 = ps.lgx1; me.rows = ps.rows;
  me.llk_y = c.n_labels ? m->llk_y : nullptr;
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
  if (c.n_labels && m->Y[0] && ps.Xsrc == m->X) { h.Y = m->Y[0]; h.ldy = m->lab_Pp[0]; h.rows = ps.rows; h.mask = m->mask; }
  h.u_tc = m->u_tc; h.u_d = m->u_d; h.tc_cell = m->tc_cell; h.dl_cell = m->dl_cell; h.llk_y = c.n_labels ? m->llk_y : nullptr;
  Timed t(m, "disc_head");
  SMX_CHECK(launch_disc_head(m->st, h));
  return SMX_OK;
}

// One backward sweep of the discriminator over the first `rows` rows of the stacked batch with upstream `up`
// [rows][32] on the logits.  with_grads: the discriminator's own gradients (its objective; nothing flows into z);
// otherwise only d objective / d z, left in m->dz_tc (the VAE objective's TC term; the weights are constants of it).
int factor_sweep(smx_model* m, int rows, const float* up, bool with_grads) {
  const MlpLayer& last = m->disc.back();
  const TensorInfo& two = m->tensors[m->t_discoutW];
  if (with_grads) {
    GemmArgs gw;
    gw.A = last.out_buf; gw.lda = last.out_p; gw.a_kmajor = 1; gw.B = up; gw.ldb = 32;
    gw.C = G_(m, m->t_discoutW); gw.ldc = two.ld; gw.M = last.out_p; gw.N = two.ld; gw.K = rows;
    gw.colsum = G_(m, m->t_discoutb);
    want_sq(m, gw, m->t_discoutW);
    Timed t(m, "disc_bwd");
    SMX_CHECK(launch_gemm(m->st, gw));
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
      SMX_CHECK(launch_gemm(m->st, g));
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
      SMX_CHECK(launch_gemm(m->st, h, &n_slabs));
    }
    if (ready) dpre_i = h.C;
  }
  return SMX_OK;
}

int factor_backward(smx_model* m, const Pass& ps) {
  SMX_CHECK(factor_sweep(m, 2 * ps.B, m->u_d, true));    // discriminator objective -> the discriminator's tensors
  SMX_CHECK(factor_sweep(m, ps.B, m->u_tc, false));      // gamma TC (+ alpha CE) -> d z
  return SMX_OK;
}

// fused-middle backward: slab-consuming BN backward of the last decoder layer, the single-workgroup
// chain down to d pre-activation of encoder layer 0, then the wide encoder weight gradient on the main
// stream while the small weight gradients run on the side stream.
int backward_mid(smx_model* m, const Pass& ps, int n_slabs) {
  const smx_config& c = m->cfg;
  MlpLayer& dL = m->dec.back();
  {
    BnBwdArgs b;
    b.dout = m->slab; b.n_slabs = n_slabs; b.slab_stride = (long)ps.B * dL.out_p; b.ld = dL.out_p;
    b.out = dL.out_buf; b.xhat = dL.xhat; b.inv_std = dL.inv_std;
    b.B = ps.B; b.H = dL.out; b.Hp = dL.out_p; b.batchnorm = c.batchnorm; b.training = ps.training;
    b.drop_scale = (ps.training && dL.drop_p > 0.f) ? 1.f / (1.f - dL.drop_p) : 1.f;
    b.dpre = dL.dpre;
    if (c.batchnorm) { b.gamma = P_(m, dL.tGamma); b.dgamma = G_(m, dL.tGamma); b.dbeta = G_(m, dL.tBeta); }
    else b.dbias = G_(m, dL.tBias);
    if (m->metrics_before_allreduce && m->have_pending_metrics) {
      b.metrics = m->pending_metrics; b.with_metrics = 1; m->have_pending_metrics = false;
    }
    attach_early_adam(m, b);
    Timed t(m, "bn_bwd");
    SMX_CHECK(launch_bn_act_bwd(m->st, b));
  }
  MidArgs ma;
  fill_mid_args(m, ps, ma);
  {
    Timed t(m, "mid_bwd");
    SMX_CHECK(launch_mid_bwd(m->st, ma));
  }
  auto dwa = [&](const float* A, int lda, int M, const float* Bm, int ldb, int N, int tW, float* colsum, bool xform) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.a_kmajor = 1; g.B = Bm; g.ldb = ldb;
    g.C = G_(m, tW); g.ldc = m->tensors[tW].ld; g.M = M; g.N = N; g.K = ps.B; g.colsum = colsum;
    want_sq(m, g, tW);
    if (xform) {
      g.use_xform = 1;
      g.xf.rows = ps.xrows; g.xf.u16 = ps.x_u16; g.xf.log1p = c.log_norm; g.xf.cell_base = ps.cell_base;
      if (ps.training && c.input_dropout > 0.f) {
        g.xf.drop_p = c.input_dropout; g.xf.drop_scale = 1.f / (1.f - c.input_dropout);
        g.xf.nk = make_key(m, ST_INPUT_DROPOUT, ps.sample, true);
        if (const Injected* ij = inj(m, ST_INPUT_DROPOUT)) { g.xf.inj_mask = ij->d; g.xf.inj_ld = ij->ld; }
      }
    }
    return g;
  };
  // all weight gradients below the output head are independent now: ONE grouped launch
  std::vector<GemmArgs> grp;
  grp.push_back(dwa(ps.Xsrc, m->Gp, m->enc[0].in_p, m->enc[0].dpre, m->enc[0].out_p, m->enc[0].out_p, m->enc[0].tW, nullptr, true));
  for (size_t i = 0; i < m->dec.size(); ++i) {
    MlpLayer& L = m->dec[i];
    const float* in = (i == 0) ? m->z : m->dec[i - 1].out_buf;
    grp.push_back(dwa(in, L.in_p, L.in_p, L.dpre, L.out_p, L.out_p, L.tW, nullptr, false));
  }
  const int lat_ld = (m->stochastic ? 2 : 1) * m->Dp;
  grp.push_back(dwa(m->enc.back().out_buf, m->enc.back().out_p, m->enc.back().out_p, m->dlat, lat_ld, lat_ld, m->t_latW,
                    G_(m, m->t_latb), false));
  for (size_t i = 1; i < m->enc.size(); ++i) {
    MlpLayer& L = m->enc[i];
    grp.push_back(dwa(m->enc[i - 1].out_buf, L.in_p, L.in_p, L.dpre, L.out_p, L.out_p, L.tW, nullptr, false));
  }
  Timed t(m, "gemm_enc_dw");
  for (size_t i = 0; i < grp.size(); i += SMX_GROUP_MAX) {
    const int n = (int)std::min<size_t>(SMX_GROUP_MAX, grp.size() - i);
    SMX_CHECK(launch_gemm_group(m->st, grp.data() + i, n));
  }
  return SMX_OK;
}

int backward_pass(smx_model* m, const Pass& ps) {
  const smx_config& c = m->cfg;
  std::fill(m->sq_count.begin(), m->sq_count.end(), 0);   // the products of this step report what they wrote
  std::fill(m->sq_reduced.begin(), m->sq_reduced.end(), 0);
  m->adam_early_pending = false; m->adam_early_from = -1;
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
  const bool hbwd = m->flags.head_bwd && !m->head_fused && head_bwd_supported(ps.B, dL.out_p, m->Gp) && (!m->scvi || (m->k >= 2 && m->k <= 3));
  if (hbwd) {
    const TensorInfo& tw = m->tensors[m->t_outW[0]];
    HeadBwdArgs hb;
    hb.D = dL.out_buf; hb.ldd = dL.out_p; hb.dP = dparams; hb.ldp = ldp; hb.W = P_(m, m->t_outW[0]); hb.ldw = tw.ld;
    hb.dW = G_(m, m->t_outW[0]); hb.db = G_(m, m->t_outb[0]);
    if (m->scvi) {
      hb.sep = 1;
      for (int ch = 0; ch < m->k; ++ch) {
        hb.Wp[ch] = P_(m, m->t_outW[ch]); hb.dWp[ch] = G_(m, m->t_outW[ch]); hb.dbp[ch] = G_(m, m->t_outb[ch]);
        if (m->sq_slots && getenv("SMX_NO_SQ_PARTIALS") == nullptr) {
          hb.sqp[ch] = m->sq_slots + m->sq_first[(size_t)m->t_outW[ch]]; hb.sq_countp[ch] = &m->sq_count[(size_t)m->t_outW[ch]];
        }
      }
    }
    hb.B = ps.B; hb.Hp = dL.out_p; hb.Gp = m->Gp; hb.n_planes = m->k;
    hb.n_slices = head_bwd_slices(ldp, ldp <= 8192 ? 16 : 32, &hb.k_chunk);
    hb.slab = m->slab; hb.slab_stride = dd_stride;
    SMX_REQUIRE((size_t)hb.n_slices * (size_t)dd_stride <= m->slab_cap, "split-K slabs exceed the slab buffer");
    if (!m->scvi && m->sq_slots && getenv("SMX_NO_SQ_PARTIALS") == nullptr) {
      hb.sq_part = m->sq_slots + m->sq_first[(size_t)m->t_outW[0]]; hb.sq_count = &m->sq_count[(size_t)m->t_outW[0]];
    }
    n_slabs = hb.n_slices;
    // label heads (SISUA / MISA): d d += d Y W_lab^T as extra slabs of this launch, the head's weight gradient with the
    // grouped launch at the end of the backward pass -- instead of a grouped launch of their own here (8.6 us at C4)
    if (m->n_heads > 0 && m->flags.label_ride && !m->fvae && !use_mid(m, ps.B)) {
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
    Timed t(m, "gemm_out_bwd");
    SMX_CHECK(launch_out_head_bwd(m->st, hb));
  }
  {
    // weight gradient and input gradient of every head read the same dP and are independent:
    // one grouped launch (dW tiles + split-K dX slabs side by side)
    std::vector<GemmArgs> grp;
    std::vector<int> is_dx;
    for (int ch = 0; ch < n_heads && !hbwd; ++ch) {
      const TensorInfo& tw = m->tensors[m->t_outW[ch]];
      const float* dp = dparams + (m->scvi ? (long)ch * m->Gp : 0);
      const int ncols = m->scvi ? m->Gp : (int)ldp;
      GemmArgs g;  // dW = d^T dP, db = colsum(dP)
      g.A = dL.out_buf; g.lda = dL.out_p; g.a_kmajor = 1; g.B = dp; g.ldb = (int)ldp;
      g.C = G_(m, m->t_outW[ch]); g.ldc = tw.ld; g.M = dL.out_p; g.N = ncols; g.K = ps.B;
      g.colsum = G_(m, m->t_outb[ch]);
      want_sq(m, g, m->t_outW[ch]);
      g.tile = TILE_128x32;
      if (!m->head_fused) { grp.push_back(g); is_dx.push_back(0); }   // the fused head already wrote dW / db
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
    if (dp_overlap(m)) {  // head gradients are final: reduce them while the rest of backward runs
      SMX_HIP(hipEventRecord(m->ev_c1, m->st));
      SMX_HIP(hipStreamWaitEvent(m->st_comm, m->ev_c1, 0));
      SMX_CHECK(dp_allreduce(m, m->bucket1_off, m->bucket1_count, m->st_comm));
      m->bucket1_in_flight = true;
    }
  }
  if (use_mid(m, ps.B)) return backward_mid(m, ps, n_slabs);
  // ---- decoder MLP; the latent head's backward runs in the epilogue of the d z product ----
  const int lat_ld = m->stochastic ? 2 * m->Dp : m->Dp;
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
    SMX_CHECK(launch_scale_prior_bwd(m->st, sp));
  }
  // Products that only the optimiser reads (the weight gradients of the first decoder layer, of the latent head and of
  // the first encoder layers) run as ONE grouped launch at the end; the last encoder layer's BatchNorm-backward
  // launch computes d h = d lat W_lat^T itself.  SMX_NO_BWD_FRONT=1: the separate launches of before.
  const MlpLayer& eL = m->enc.back();
  const bool bfront = m->flags.bwd_front && !sync_bn_on(m, ps.training) && bn_bwd_front_supported(ps.B, lat_ld) && eL.out_p % 8 == 0;
  std::vector<GemmArgs> tail;
  SMX_CHECK(mlp_backward(m, m->dec, ps, m->z, m->Dp, false, n_slabs, false, nullptr, "", &le, nullptr, nullptr, bfront ? &tail : nullptr));
  for (const GemmArgs& g : lab_dw) tail.push_back(g);
  BnBwdArgs gf;
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
    } else {   // independent: one grouped launch
      GemmArgs& h = pair[1];
      h.A = m->dlat; h.lda = lat_ld; h.B = P_(m, m->t_latW); h.ldb = tw.ld; h.b_nmajor = 1;
      h.C = m->slab; h.ldc = eL.out_p; h.slab_stride = (long)ps.B * eL.out_p;
      h.M = ps.B; h.N = eL.out_p; h.K = lat_ld;
      Timed t(m, "gemm_lat_bwd");
      SMX_CHECK(launch_gemm_group(m->st, pair, 2));
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
  SMX_CHECK(mlp_backward(m, m->enc, ps, ps.Xsrc, m->Gp, true, 1, true, nullptr, "gemm_enc_dw", nullptr, &dw0[n_dw0], bfront ? &gf : nullptr,
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
    if (wg_ok) SMX_CHECK(launch_wgrad_group(m->st, tail.data(), (int)tail.size(), ps.B));
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
  a.use_sq = (m->sq_slots != nullptr && !dp_active(m) && getenv("SMX_NO_SQ_PARTIALS") == nullptr) ? 1 : 0;
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
}

int optimizer_pass(smx_model* m) {
  const smx_config& c = m->cfg;
  SMX_CHECK(side_join(m));
  if (dp_active(m) && m->have_pending_metrics) {   // no BatchNorm-backward launch took them along
    SMX_CHECK(launch_metrics(m->st, m->pending_metrics));
    m->have_pending_metrics = false;
  }
  if (dp_active(m)) {
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
  { Timed null_pair(m, "null"); }  // an event pair around nothing: the timing method's own overhead
  SMX_CHECK(csr_stage(m, ps));     // sparse store: this minibatch's rows as a dense tile (no-op otherwise)
  SMX_CHECK(forward_pass(m, ps, true, true));
  SMX_CHECK(backward_pass(m, ps));
  SMX_CHECK(optimizer_pass(m));
  return SMX_OK;
}

int read_metrics(smx_model* m, smx_metrics* out) {
  if (!out) return SMX_OK;
  float h[8];
  std::vector<float> norms(m->tensors.size());
  SMX_HIP(hipMemcpyAsync(h, m->grads + m->tail_off_metrics, sizeof(h), hipMemcpyDeviceToHost, m->st));
  SMX_HIP(hipMemcpyAsync(norms.data(), m->tensor_norm, norms.size() * sizeof(float), hipMemcpyDeviceToHost, m->st));
  SMX_HIP(hipStreamSynchronize(m->st));
  out->loss = h[0]; out->nllk_x = h[1]; out->nllk_y = h[2]; out->kl = h[3]; out->kl_l = h[4];
  float mx = 0.f;
  for (float v : norms) mx = (v > mx || v != v) ? v : mx;
  out->grad_norm_max = mx;
  out->nan_flag = !(isfinite(h[0]) && isfinite(h[1]) && isfinite(h[3]) && isfinite(mx));
  out->step = (int32_t)m->h_next;
  out->tc = h[5]; out->dtc_loss = h[6];
  if (m->fvae && !(isfinite(h[5]) && isfinite(h[6]))) out->nan_flag = 1;
  return SMX_OK;
}

void drop_graphs(smx_model* m) {  // captured graphs bake device pointers in: drop them when a buffer moves
  for (auto& kv : m->graphs) hipGraphExecDestroy(kv.second);
  m->graphs.clear();
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
  SMX_HIP(hipMemcpyAsync(m->order, order, n * sizeof(int32_t), hipMemcpyHostToDevice, m->st));
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
  static const bool no_graph_comm = getenv("SMX_NO_GRAPH_COMM") != nullptr;
  if (use_graph && !m->local && !(m->comm && (no_graph_comm || m->graph_comm_failed)) && !m->use_injected && m->timing_label.empty()) {
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

}  // namespace

// ===========================================================================
// C-ABI
// ===========================================================================
extern "C" {

const char* smx_last_error(void) { return g_err.c_str(); }
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
  SMX_REQUIRE(cfg->model >= SMX_MODEL_VAE && cfg->model <= SMX_MODEL_FVAE, "unknown model kind");
  if (cfg->model == SMX_MODEL_FVAE) {
    SMX_REQUIRE(cfg->disc_layers >= 1 && cfg->disc_layers <= SMX_MAX_LAYERS && cfg->disc_units >= 1, "fvae: discriminator needs 1..8 hidden layers");
    SMX_REQUIRE(cfg->disc_leak >= 0.f && cfg->disc_leak < 1.f, "fvae: leaky-ReLU slope in [0, 1)");
    SMX_REQUIRE(cfg->n_labels <= 1, "fvae: at most one (one-hot) label variable");
    if (cfg->n_labels == 1)
      SMX_REQUIRE(cfg->label_llk[0] == SMX_LABEL_ONEHOT && cfg->label_dim[0] >= 2 && cfg->label_dim[0] <= 32, "fvae: the label variable is one-hot with 2..32 classes");
  }
  if (cfg->model == SMX_MODEL_SCALE) SMX_REQUIRE(cfg->n_components >= 2 && cfg->n_components <= 32, "scale: 2..32 mixture components");
  SMX_REQUIRE(cfg->likelihood >= SMX_LLK_NB && cfg->likelihood <= SMX_LLK_ZINBD, "unknown likelihood");
  SMX_REQUIRE(cfg->n_labels >= 0 && cfg->n_labels <= SMX_MAX_LABELS, "too many label heads");
  SMX_REQUIRE(cfg->model == SMX_MODEL_SISUA || cfg->model == SMX_MODEL_FVAE || cfg->n_labels == 0, "label heads need model = SISUA");
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
  m->k = (cfg->likelihood == SMX_LLK_ZINB || cfg->likelihood == SMX_LLK_ZINBD) ? 3 : 2;
  m->stochastic = cfg->model != SMX_MODEL_DCA; m->scvi = cfg->model == SMX_MODEL_SCVI; m->scale = cfg->model == SMX_MODEL_SCALE;
  m->fvae = cfg->model == SMX_MODEL_FVAE; m->n_heads = m->fvae ? 0 : cfg->n_labels;
  m->Bmax = cfg->max_batch;
  int rc = SMX_OK;
  auto fail = [&](int code) { smx_model_destroy(m); return code; };
  if (hipStreamCreate(&m->st) != hipSuccess) { set_error("hipStreamCreate failed"); return fail(SMX_ERR_HIP); }
  // Forked streams inside the captured graph measured SLOWER on ROCm 7.2 (+38 us per step: the
  // cross-stream dependencies cost more than the overlap buys); opt-in only.
  if (getenv("SMX_SIDE_STREAM")) {
    if (hipStreamCreate(&m->st2) != hipSuccess || hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_fork2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming) != hipSuccess) {
      set_error("side stream creation failed"); return fail(SMX_ERR_HIP);
    }
  }
  // ---- manifest (same order as oracle/sisua_oracle.py:manifest) ----
  const bool bnorm = cfg->batchnorm != 0;
  int h = build_mlp(m, m->enc, "enc", m->G, cfg->n_enc, cfg->enc_units, ST_ENC_DROPOUT, cfg->dropout_enc, bnorm);
  m->t_latW = add_tensor(m, "lat/W", h, (m->stochastic ? 2 : 1) * m->D, m->stochastic ? 2 : 1, false);
  m->t_latb = add_tensor(m, "lat/b", 1, (m->stochastic ? 2 : 1) * m->D, m->stochastic ? 2 : 1, true);
  if (m->scale) {   // trainable mixture prior: logits [C], means and raw scales [C][D]
    m->t_prLogits = add_tensor(m, "prior/logits", 1, cfg->n_components, 1, true);
    m->t_prLoc = add_tensor(m, "prior/loc", cfg->n_components, m->D, 1, false);
    m->t_prScale = add_tensor(m, "prior/scale", cfg->n_components, m->D, 1, false);
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
    const int n_out = cfg->n_labels ? cfg->label_dim[0] : 1;
    m->t_discoutW = add_tensor(m, "discout/W", hu, n_out, 1, false);
    m->t_discoutb = add_tensor(m, "discout/b", 1, n_out, 1, true);
    if (cfg->n_labels) m->lab_Pp[0] = round_up(cfg->label_dim[0], 32);
  }
  if (m->scvi) {
    for (int ch = 0; ch < m->k; ++ch) {
      m->t_outW[ch] = add_tensor(m, "out" + std::to_string(ch) + "/W", hd, m->G, 1, false);
      m->t_outb[ch] = add_tensor(m, "out" + std::to_string(ch) + "/b", 1, m->G, 1, true);
    }
  } else {
    m->t_outW[0] = add_tensor(m, "out/W", hd, m->k * m->G, m->k, false);
    m->t_outb[0] = add_tensor(m, "out/b", 1, m->k * m->G, m->k, true);
  }
  for (int j = 0; j < m->n_heads; ++j) {
    SMX_REQUIRE(cfg->label_dim[j] > 0, "label_dim must be > 0");
    SMX_REQUIRE(cfg->label_llk[j] >= SMX_LABEL_NB && cfg->label_llk[j] <= SMX_LABEL_MIXNB, "unknown label likelihood");
    if (cfg->label_llk[j] == SMX_LABEL_MIXNB) SMX_REQUIRE(cfg->label_components[j] >= 2 && cfg->label_components[j] <= 4, "mixture label heads have 2..4 components");
    m->lab_ky[j] = cfg->label_llk[j] == SMX_LABEL_NB ? 2 : cfg->label_llk[j] == SMX_LABEL_ONEHOT ? 1 : 3 * cfg->label_components[j];
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
  if ((rc = dmalloc(&m->params, m->flat_count))) return fail(rc);
  if ((rc = dmalloc(&m->grads, m->grads_count))) return fail(rc);
  if ((rc = dmalloc(&m->adam_m, m->flat_count))) return fail(rc);
  if ((rc = dmalloc(&m->adam_v, m->flat_count))) return fail(rc);
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
      if ((rc = dmalloc(&L.xhat, 2 * B * L.out_p)) || (rc = dmalloc(&L.out_buf, 2 * B * L.out_p)) || (rc = dmalloc(&L.dpre, 2 * B * L.out_p)))
        return fail(rc);
    }
    if ((rc = dmalloc(&m->zz, 2 * B * m->Dp)) || (rc = dmalloc(&m->u_tc, B * 32)) || (rc = dmalloc(&m->u_d, 2 * B * 32)) ||
        (rc = dmalloc(&m->tc_cell, B)) || (rc = dmalloc(&m->dl_cell, 2 * B)) || (rc = dmalloc(&m->dz_tc, B * m->Dp)) ||
        (rc = dmalloc(&m->disc_dpre, B * up)) || (rc = dmalloc(&m->disc_db, (size_t)up)))
      return fail(rc);
  }
  m->slab_cap = (size_t)(64 * 3 + SMX_MAX_LABELS + 1) * B * m->max_feat_p;
  const size_t lat_ld = (m->stochastic ? 2 : 1) * (size_t)m->Dp;
  const size_t ldp = (size_t)m->k * m->Gp;
  if ((rc = dmalloc(&m->slab, m->slab_cap)) || (rc = dmalloc(&m->latbuf, B * lat_ld)) || (rc = dmalloc(&m->dlat, B * lat_ld)) ||
      (rc = dmalloc(&m->z, B * m->Dp)) || (rc = dmalloc(&m->noise_eps, B * m->Dp)) || (rc = dmalloc(&m->sig, B * m->Dp)) || (rc = dmalloc(&m->eps, B * m->Dp)) ||
      (rc = dmalloc(&m->kl, B)) || (rc = dmalloc(&m->P, B * ldp)) || (rc = dmalloc(&m->dP, B * ldp)) ||
      (rc = dmalloc(&m->llk_part, B * (size_t)std::max(std::max(loss_chunks_max(m->Gp), out_head_chunks(m->Gp)), head_loss_chunks(m->Gp)))) || (rc = dmalloc(&m->llk_y, B)) ||
      (rc = dmalloc(&m->rows2[0], B)) || (rc = dmalloc(&m->rows2[1], B)) || (rc = dmalloc(&m->state3, (size_t)3)) ||
      (rc = dmalloc(&m->hostX, B * m->Gp)) || (rc = dmalloc(&m->hostLib, B * 2)) || (rc = dmalloc(&m->hostLgx1, B)))
    return fail(rc);
  if (m->scale && ((rc = dmalloc(&m->resp, B * 32)) || (rc = dmalloc(&m->dklz, B * m->Dp)))) return fail(rc);
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
  m->eps_ahead_ok = latent_head_fusable(m->enc.back().out_p, (m->stochastic ? 2 : 1) * m->Dp, m->Dp);
  // ---- optimiser chunk table ----
  std::vector<OptChunk> chunks;
  // floats per optimiser workgroup (SMX_OPT_CHUNK = 1024 | 2048 | 4096 | 8192)
  static const int ch_env = getenv("SMX_OPT_CHUNK") ? atoi(getenv("SMX_OPT_CHUNK")) : 0;
  const int CH = (ch_env == 1024 || ch_env == 2048 || ch_env == 4096 || ch_env == 8192) ? ch_env : 4096;
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
  if ((rc = dmalloc(&m->chunks, chunks.size())) || (rc = dmalloc(&m->partial, chunks.size())) ||
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
  if (m->st2) hipStreamSynchronize(m->st2);
  for (auto& kv : m->graphs) hipGraphExecDestroy(kv.second);
  for (auto& ev : m->timing_events) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
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
  fr(m->resp); fr(m->dklz);
  for (auto& L : m->disc) { fr(L.xhat); fr(L.out_buf); fr(L.dpre); }
  fr(m->zz); fr(m->u_tc); fr(m->u_d); fr(m->tc_cell); fr(m->dl_cell); fr(m->dz_tc); fr(m->disc_dpre); fr(m->disc_db);
  fr(m->noise_eps); fr(m->latbuf); fr(m->dlat); fr(m->z); fr(m->sig); fr(m->eps); fr(m->kl);
  fr(m->latlbuf); fr(m->dlatl); fr(m->lsmp); fr(m->lsig); fr(m->leps); fr(m->kl_l); fr(m->dl);
  fr(m->P); fr(m->dP); fr(m->raw); fr(m->draw); fr(m->rho); fr(m->llk_part); fr(m->llk_y); fr(m->slab);
  fr(m->chunks); fr(m->partial); fr(m->tensor_norm); fr(m->sq_slots);
  if (m->pinned) hipHostFree(m->pinned);
  if (m->pred_stage) hipFree(m->pred_stage);
  if (m->score_buf) hipFree(m->score_buf);
  if (m->score_wimg) hipFree(m->score_wimg);
  if (m->score_aux) hipFree(m->score_aux);
  for (auto& kv : m->injected) fr(kv.second.d);
  if (m->st_comm) { hipStreamSynchronize(m->st_comm); hipStreamDestroy(m->st_comm); }
  if (m->ev_c1) hipEventDestroy(m->ev_c1);
  if (m->ev_c2) hipEventDestroy(m->ev_c2);
  if (m->ev_c3) hipEventDestroy(m->ev_c3);
  if (m->ev_fork) hipEventDestroy(m->ev_fork);
  if (m->ev_fork2) hipEventDestroy(m->ev_fork2);
  if (m->ev_join) hipEventDestroy(m->ev_join);
  if (m->st2) hipStreamDestroy(m->st2);
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

static int dataset_upload_impl(smx_model* m, const void* X, bool u16, int64_t n_cells, const float* const* labels,
                               const float* library, const uint8_t* label_mask, int64_t cell_id_base);

int smx_dataset_upload(smx_model* m, const float* X, int64_t n_cells, const float* const* labels, const float* library,
                       const uint8_t* label_mask, int64_t cell_id_base) {
  return dataset_upload_impl(m, X, false, n_cells, labels, library, label_mask, cell_id_base);
}

int smx_dataset_upload_u16(smx_model* m, const uint16_t* X, int64_t n_cells, const float* const* labels, const float* library,
                           const uint8_t* label_mask, int64_t cell_id_base) {
  return dataset_upload_impl(m, X, true, n_cells, labels, library, label_mask, cell_id_base);
}

static int upload_side_arrays(smx_model* m, int64_t n_cells, const float* const* labels, const float* library, const uint8_t* label_mask) {
  int rc;
  for (int j = 0; j < m->cfg.n_labels; ++j) {
    const int P = m->cfg.label_dim[j], Pp = m->lab_Pp[j];
    if ((rc = dmalloc(&m->Y[j], (size_t)n_cells * Pp))) return rc;
    SMX_HIP(hipMemcpy2D(m->Y[j], (size_t)Pp * sizeof(float), labels[j], (size_t)P * sizeof(float), (size_t)P * sizeof(float),
                        (size_t)n_cells, hipMemcpyHostToDevice));
  }
  if (library) {
    if ((rc = dmalloc(&m->library, (size_t)n_cells * 2))) return rc;
    SMX_HIP(hipMemcpy(m->library, library, (size_t)n_cells * 2 * sizeof(float), hipMemcpyHostToDevice));
  }
  if (label_mask) {
    if ((rc = dmalloc(&m->mask, (size_t)n_cells))) return rc;
    SMX_HIP(hipMemcpy(m->mask, label_mask, (size_t)n_cells, hipMemcpyHostToDevice));
  }
  return SMX_OK;
}

// Compact sparse store: the counts as CSR (indptr [n_cells + 1], column indices and values of the non-zeros, rows in
// order, columns < n_genes) -- 8 bytes per non-zero instead of 4 per entry (7-12 % non-zeros in the named datasets).
// Every pass expands its minibatch's rows into a dense float32 tile first (csr_stage), so results are bit-identical to
// the float32 store; the resident-matrix kernels (library statistics, corruption) stay with the dense stores.
int smx_dataset_upload_csr(smx_model* m, const int64_t* indptr, const int32_t* cols, const float* vals, int64_t n_cells,
                           const float* const* labels, const float* library, const uint8_t* label_mask, int64_t cell_id_base) {
  SMX_REQUIRE(m && indptr && n_cells > 0, "bad dataset");
  SMX_REQUIRE(n_cells < (int64_t)1 << 31, "row ids are int32");
  SMX_REQUIRE(!m->scvi || library, "scvi needs the library prior (scvi.py:100-105)");
  for (int j = 0; j < m->cfg.n_labels; ++j) SMX_REQUIRE(labels && labels[j], "missing label matrix");
  const int64_t nnz = indptr[n_cells];
  SMX_REQUIRE(indptr[0] == 0 && nnz >= 0 && (nnz == 0 || (cols && vals)), "bad CSR arrays");
  for (int64_t r = 0; r < n_cells; ++r) SMX_REQUIRE(indptr[r + 1] >= indptr[r], "CSR indptr must not decrease");
  for (int64_t i = 0; i < nnz; ++i) SMX_REQUIRE(cols[i] >= 0 && cols[i] < m->G, "CSR column index out of range");
  SMX_HIP(hipStreamSynchronize(m->st));
  drop_graphs(m);
  auto fr = [](void* p) { if (p) hipFree(p); };
  release_csr(m);
  fr(m->X); fr(m->library); fr(m->mask); fr(m->lgx1);
  m->X = nullptr; m->library = nullptr; m->mask = nullptr; m->lgx1 = nullptr;
  for (int j = 0; j < SMX_MAX_LABELS; ++j) { fr(m->Y[j]); m->Y[j] = nullptr; }
  m->N = n_cells; m->cell_base = cell_id_base; m->x_u16 = false;
  int rc;
  m->x_csr = true;
  if ((rc = dmalloc(&m->csr_indptr, (size_t)n_cells + 1)) || (rc = dmalloc(&m->csr_cols, (size_t)std::max<int64_t>(nnz, 1))) ||
      (rc = dmalloc(&m->csr_vals, (size_t)std::max<int64_t>(nnz, 1))) || (rc = dmalloc(&m->xbatch, (size_t)m->Bmax * m->Gp)) ||
      (rc = dmalloc(&m->lgx1, (size_t)n_cells)))
    return rc;
  m->X = m->xbatch;   // (non-null: "a dataset is resident"; csr_stage fills it per pass)
  SMX_HIP(hipMemcpy(m->csr_indptr, indptr, ((size_t)n_cells + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  if (nnz) {
    SMX_HIP(hipMemcpy(m->csr_cols, cols, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
    SMX_HIP(hipMemcpy(m->csr_vals, vals, (size_t)nnz * sizeof(float), hipMemcpyHostToDevice));
  }
  SMX_CHECK(launch_csr_row_stats(m->st, m->csr_indptr, m->csr_vals, m->N, m->lgx1));
  return upload_side_arrays(m, n_cells, labels, library, label_mask);
}

static int dataset_upload_impl(smx_model* m, const void* X, bool u16, int64_t n_cells, const float* const* labels,
                               const float* library, const uint8_t* label_mask, int64_t cell_id_base) {
  SMX_REQUIRE(m && X && n_cells > 0, "bad dataset");
  SMX_REQUIRE(n_cells < (int64_t)1 << 31, "row ids are int32");
  SMX_REQUIRE(!m->scvi || library, "scvi needs the library prior (scvi.py:100-105)");
  for (int j = 0; j < m->cfg.n_labels; ++j) SMX_REQUIRE(labels && labels[j], "missing label matrix");
  SMX_HIP(hipStreamSynchronize(m->st));
  drop_graphs(m);
  auto fr = [](void* p) { if (p) hipFree(p); };
  release_csr(m);
  fr(m->X); fr(m->library); fr(m->mask); fr(m->lgx1);
  m->X = nullptr; m->library = nullptr; m->mask = nullptr; m->lgx1 = nullptr;
  for (int j = 0; j < SMX_MAX_LABELS; ++j) { fr(m->Y[j]); m->Y[j] = nullptr; }
  m->N = n_cells; m->cell_base = cell_id_base;
  int rc;
  m->x_u16 = u16;
  if (u16) {   // compact store: uint16 counts, same row pitch in ELEMENTS (Gp), half the bytes
    uint16_t* xh = nullptr;
    if ((rc = dmalloc(&xh, (size_t)n_cells * m->Gp)) || (rc = dmalloc(&m->lgx1, (size_t)n_cells))) return rc;
    m->X = reinterpret_cast<float*>(xh);
    SMX_HIP(hipMemcpy2D(xh, (size_t)m->Gp * sizeof(uint16_t), X, (size_t)m->G * sizeof(uint16_t), (size_t)m->G * sizeof(uint16_t),
                        (size_t)n_cells, hipMemcpyHostToDevice));
  } else {
    if ((rc = dmalloc(&m->X, (size_t)n_cells * m->Gp)) || (rc = dmalloc(&m->lgx1, (size_t)n_cells))) return rc;
    SMX_HIP(hipMemcpy2D(m->X, (size_t)m->Gp * sizeof(float), X, (size_t)m->G * sizeof(float), (size_t)m->G * sizeof(float),
                        (size_t)n_cells, hipMemcpyHostToDevice));
  }
  // per-row constant sum_g lgamma(x+1) of the likelihood, on the device (one wave per row)
  SMX_CHECK(launch_row_stats(m->st, m->X, m->x_u16 ? 1 : 0, m->Gp, m->N, m->G, m->lgx1, nullptr));
  return upload_side_arrays(m, n_cells, labels, library, label_mask);
}

int64_t smx_dataset_size(const smx_model* m) { return m ? m->N : 0; }

int smx_dataset_library(smx_model* m, float stats[2]) {
  SMX_REQUIRE(m && m->X && m->N > 0, "no resident dataset");
  SMX_REQUIRE(!m->x_csr, "the resident-matrix kernels take a dense store (float32 / uint16), not the sparse one");
  SMX_HIP(hipStreamSynchronize(m->st));
  double* work = nullptr;   // [N] log counts + [2] moments
  int rc;
  if ((rc = dmalloc(&work, (size_t)m->N + 2))) return rc;
  if (!m->library && (rc = dmalloc(&m->library, (size_t)m->N * 2))) { hipFree(work); return rc; }
  drop_graphs(m);   // a captured step may hold the old (null) library pointer
  rc = launch_row_stats(m->st, m->X, m->x_u16 ? 1 : 0, m->Gp, m->N, m->G, m->lgx1, work);
  if (rc == SMX_OK) rc = launch_library_stats(m->st, work, m->N, work + m->N, m->library);
  double h[2] = {0.0, 0.0};
  if (rc == SMX_OK) {
    hipError_t e = hipMemcpyAsync(h, work + m->N, sizeof(h), hipMemcpyDeviceToHost, m->st);
    if (e == hipSuccess) e = hipStreamSynchronize(m->st);
    if (e != hipSuccess) { set_error(std::string("dataset_library failed: ") + hipGetErrorString(e)); rc = SMX_ERR_HIP; }
  }
  hipFree(work);
  if (rc == SMX_OK && stats) { stats[0] = (float)h[0]; stats[1] = (float)h[1]; }
  return rc;
}

int smx_dataset_corrupt(smx_model* m, double dropout, double retain_rate, uint64_t seed, int64_t* n_corrupted) {
  SMX_REQUIRE(m && m->X && m->N > 0, "no resident dataset");
  SMX_REQUIRE(!m->x_csr, "the resident-matrix kernels take a dense store (float32 / uint16), not the sparse one");
  SMX_REQUIRE(dropout >= 0.0 && dropout < 1.0, "dropout value must be >= 0 and < 1");   // utils.py:184-185
  SMX_REQUIRE(retain_rate >= 0.0 && retain_rate <= 1.0, "retain_rate must be in [0, 1]");
  if (n_corrupted) *n_corrupted = 0;
  if (!((dropout > 0.0 && dropout < 1.0) || (retain_rate > 0.0 && retain_rate < 1.0))) return SMX_OK;   // utils.py:188-189
  SMX_HIP(hipStreamSynchronize(m->st));
  unsigned long long* hist = nullptr;
  int rc;
  if ((rc = dmalloc(&hist, 256))) return rc;
  CorruptArgs a;
  a.X = m->X; a.ld = m->Gp; a.N = m->N; a.G = m->G; a.u16 = m->x_u16 ? 1 : 0;
  a.k0 = (uint32_t)(seed & 0xFFFFFFFFu); a.k1 = (uint32_t)(seed >> 32); a.cell_base = (uint32_t)m->cell_base;
  a.hist = hist;
  a.thr_binom = (uint64_t)floor(retain_rate * 4294967296.0);
  unsigned long long h[256];
  unsigned long long rank = 0;   // 1-based rank of the threshold key among the keys that share the prefix
  bool nothing = false;
  for (int pass = 0; pass < 8 && rc == SMX_OK && !nothing; ++pass) {
    hipError_t e = hipMemsetAsync(hist, 0, sizeof(h), m->st);
    if (e == hipSuccess) rc = launch_corrupt_hist(m->st, a, pass);
    if (rc == SMX_OK && e == hipSuccess) e = hipMemcpyAsync(h, hist, sizeof(h), hipMemcpyDeviceToHost, m->st);
    if (rc == SMX_OK && e == hipSuccess) e = hipStreamSynchronize(m->st);
    if (e != hipSuccess) { set_error(std::string("dataset_corrupt failed: ") + hipGetErrorString(e)); rc = SMX_ERR_HIP; }
    if (rc != SMX_OK) break;
    if (pass == 0) {
      unsigned long long nnz = 0;
      for (int d = 0; d < 256; ++d) nnz += h[d];
      rank = (unsigned long long)floor(dropout * (double)nnz);   // int(np.floor(dropout * len(i))), utils.py:213-215
      if (rank == 0) { nothing = true; break; }
    }
    unsigned long long cum = 0;
    int digit = 255;
    for (int d = 0; d < 256; ++d) {
      if (cum + h[d] >= rank) { digit = d; break; }
      cum += h[d];
    }
    rank -= cum;
    a.prefix |= (uint64_t)digit << (56 - 8 * pass);
  }
  if (rc == SMX_OK && !nothing) {
    hipError_t e = hipMemsetAsync(hist, 0, sizeof(unsigned long long), m->st);
    if (e == hipSuccess) rc = launch_corrupt_apply(m->st, a);
    // the per-row constant sum lgamma(x+1) follows the matrix
    if (rc == SMX_OK) rc = launch_row_stats(m->st, m->X, m->x_u16 ? 1 : 0, m->Gp, m->N, m->G, m->lgx1, nullptr);
    if (rc == SMX_OK && e == hipSuccess) e = hipMemcpyAsync(h, hist, sizeof(unsigned long long), hipMemcpyDeviceToHost, m->st);
    if (rc == SMX_OK && e == hipSuccess) e = hipStreamSynchronize(m->st);
    if (e != hipSuccess) { set_error(std::string("dataset_corrupt failed: ") + hipGetErrorString(e)); rc = SMX_ERR_HIP; }
    if (rc == SMX_OK && n_corrupted) *n_corrupted = (int64_t)h[0];
  }
  hipFree(hist);
  return rc;
}

int smx_dataset_read(smx_model* m, int64_t row0, int64_t n_rows, float* X, float* row_const, float* library) {
  SMX_REQUIRE(m && m->X, "no resident dataset");
  SMX_REQUIRE(row0 >= 0 && n_rows > 0 && row0 + n_rows <= m->N, "rows out of range");
  SMX_HIP(hipStreamSynchronize(m->st));
  if (X && m->x_csr) {   // the sparse store's rows, expanded a tile at a time
    for (int64_t r = 0; r < n_rows; r += m->Bmax) {
      const int B = (int)std::min<int64_t>(m->Bmax, n_rows - r);
      SMX_CHECK(launch_csr_expand(m->st, m->csr_indptr, m->csr_cols, m->csr_vals, nullptr, (long)(row0 + r), B, m->Gp, m->xbatch));
      SMX_HIP(hipMemcpy2DAsync(X + (size_t)r * m->G, (size_t)m->G * sizeof(float), m->xbatch, (size_t)m->Gp * sizeof(float),
                               (size_t)m->G * sizeof(float), (size_t)B, hipMemcpyDeviceToHost, m->st));
      SMX_HIP(hipStreamSynchronize(m->st));
    }
  } else if (X && m->x_u16) {
    std::vector<uint16_t> tmp((size_t)n_rows * m->G);
    SMX_HIP(hipMemcpy2D(tmp.data(), (size_t)m->G * sizeof(uint16_t), reinterpret_cast<const uint16_t*>(m->X) + (size_t)row0 * m->Gp,
                        (size_t)m->Gp * sizeof(uint16_t), (size_t)m->G * sizeof(uint16_t), (size_t)n_rows, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tmp.size(); ++i) X[i] = (float)tmp[i];
  } else if (X)
    SMX_HIP(hipMemcpy2D(X, (size_t)m->G * sizeof(float), m->X + (size_t)row0 * m->Gp, (size_t)m->Gp * sizeof(float),
                        (size_t)m->G * sizeof(float), (size_t)n_rows, hipMemcpyDeviceToHost));
  if (row_const) SMX_HIP(hipMemcpy(row_const, m->lgx1 + row0, (size_t)n_rows * sizeof(float), hipMemcpyDeviceToHost));
  if (library) {
    SMX_REQUIRE(m->library, "no library prior resident");
    SMX_HIP(hipMemcpy(library, m->library + 2 * row0, (size_t)n_rows * 2 * sizeof(float), hipMemcpyDeviceToHost));
  }
  return SMX_OK;
}

int smx_train_step(smx_model* m, const int32_t* row_ids, int32_t batch, smx_metrics* out) {
  return smx_train_steps(m, row_ids, 1, batch, 0, out);
}
int smx_train_step_graph(smx_model* m, const int32_t* row_ids, int32_t batch, smx_metrics* out) {
  return smx_train_steps(m, row_ids, 1, batch, 1, out);
}

int smx_train_steps(smx_model* m, const int32_t* order, int32_t n_steps, int32_t batch, int use_graph, smx_metrics* out) {
  SMX_REQUIRE(m && order && n_steps > 0, "bad arguments");
  SMX_REQUIRE(batch > 0 && batch <= m->Bmax, "batch must be in 1..max_batch");
  SMX_CHECK(check_rows(m, order, (size_t)n_steps * batch));
  SMX_CHECK(upload_order(m, order, (size_t)n_steps * batch, (size_t)n_steps));
  for (int s = 0; s < n_steps; ++s) SMX_CHECK(launch_train(m, batch, use_graph != 0, s, n_steps));
  if (m->use_injected) { m->use_injected = false; }
  // a non-finite loss / gradient norm is REPORTED (out->nan_flag), not an error of the call: terminate_on_nan
  // (configs/base.yaml:59) is the caller's decision
  SMX_CHECK(read_metrics(m, out));
  return SMX_OK;
}

int smx_metrics_history(smx_model* m, int32_t n_steps, float* host) {
  SMX_REQUIRE(m && host && n_steps > 0 && n_steps <= m->mhist_steps, "no such history (steps of the last smx_train_steps call)");
  SMX_HIP(hipStreamSynchronize(m->st));
  SMX_HIP(hipMemcpy(host, m->mhist, (size_t)n_steps * 8 * sizeof(float), hipMemcpyDeviceToHost));
  return SMX_OK;
}

static int setup_pass(smx_model* m, Pass& ps, const int32_t* row_ids, const float* host_x, const float* host_library,
                      int32_t batch, int training, int sample) {
  SMX_REQUIRE(batch > 0 && batch <= m->Bmax, "batch must be in 1..max_batch");
  ps.B = batch; ps.training = training; ps.sample = sample; ps.global_batch = batch;
  if (row_ids) {
    SMX_CHECK(check_rows(m, row_ids, (size_t)batch));
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

int smx_eval_step(smx_model* m, const int32_t* row_ids, int32_t batch, smx_metrics* out) {
  SMX_REQUIRE(m && row_ids, "bad arguments");
  Pass ps;
  SMX_CHECK(setup_pass(m, ps, row_ids, nullptr, nullptr, batch, 0, 0));
  SMX_CHECK(forward_pass(m, ps, true, false));
  SMX_CHECK(read_metrics(m, out));
  return SMX_OK;
}

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
  const int lat_ld = m->stochastic ? 2 * Dp : Dp;
  std::vector<float> tmp;
  auto fetch2d = [&](float* dst, const float* src, int ld, int w) -> int {
    if (!dst) return SMX_OK;
    SMX_HIP(hipMemcpy2D(dst, (size_t)w * sizeof(float), src, (size_t)ld * sizeof(float), (size_t)w * sizeof(float), (size_t)B,
                        hipMemcpyDeviceToHost));
    return SMX_OK;
  };
  SMX_CHECK(fetch2d(z_mean, m->latbuf, lat_ld, D));
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

static bool stacked_scoring_ok(const smx_model* m);
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

// Decoder layers over `rows` stacked rows (evaluation mode: moving statistics, no dropout; smx_score.hip).  The last
// layer's output: last_form 0 row-major f32 in place, 1 k-major f32 in ht [Hp][rows], 2 its three-way bf16 split in ht.
static bool stacked_scoring_ok(const smx_model* m);
static int stacked_decoder(smx_model* m, const float* z, long rows, float* const* hb, int last_form, float* ht, const float** out, int* out_ld) {
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

int smx_predict(smx_model* m, const float* host_x, const float* host_library, int64_t n_cells, int32_t batch, int32_t n_samples,
                float* z_mean, float* z_scale, float* z_samples, float* l_mean, float* l_scale, float* l_samples,
                float* x_params, float* const* y_params) {
  SMX_REQUIRE(m && host_x && n_cells > 0 && n_samples > 0, "bad arguments");
  SMX_REQUIRE(batch > 0 && batch <= m->Bmax, "batch must be in 1..max_batch");
  const size_t N = (size_t)n_cells, G = (size_t)m->G, D = (size_t)m->D, k = (size_t)m->k, S = (size_t)n_samples;
  const int Dp = m->Dp, lat_ld = m->stochastic ? 2 * Dp : Dp;
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
  for (int j = 0; j < m->n_heads; ++j)
    if (y_params && y_params[j]) { wy[j] = (size_t)m->lab_ky[j] * (size_t)m->cfg.label_dim[j]; per_cell += S * wy[j]; }
  SMX_REQUIRE(per_cell > 0, "no output requested");
  // 128 MB of staging (SMX_PREDICT_STAGE_FLOATS: tests force several chunks on small problems)
  const size_t cap_floats = getenv("SMX_PREDICT_STAGE_FLOATS") ? (size_t)std::max(1L, atol(getenv("SMX_PREDICT_STAGE_FLOATS"))) : (size_t)32 << 20;
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
  for (int j = 0; j < m->n_heads; ++j)
    if (wy[j]) { s_y[j] = st; st += S * C * wy[j]; }
  auto out = [&](float* dst, const float* src, size_t count) -> int {   // one contiguous device -> host copy
    SMX_HIP(hipMemcpyAsync(dst, src, count * sizeof(float), hipMemcpyDeviceToHost, m->st));
    return SMX_OK;
  };
  const bool stack = S > 1 && stacked_scoring_ok(m) && !m->scvi;
  for (size_t c0 = 0; c0 < N; c0 += C) {
    const size_t Cn = std::min(C, N - c0);   // cells of this chunk
    for (size_t b0 = 0; b0 < Cn; b0 += (size_t)batch) {
      const int B = (int)std::min<size_t>((size_t)batch, Cn - b0);
      const size_t g0 = c0 + b0;
      Pass ps;
      SMX_CHECK(setup_pass(m, ps, nullptr, host_x + g0 * G, host_library ? host_library + g0 * 2 : nullptr, B, 0, 0));
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
          add1(s_zm ? s_zm + b0 * D : nullptr, D, m->latbuf, (size_t)lat_ld, D);
          add1(s_zs ? s_zs + b0 * D : nullptr, D, m->sig, (size_t)Dp, D);
          if (J.n) { hipLaunchKernelGGL(pack_kernel, dim3(8, (unsigned)J.n, 1), dim3(256), 0, m->st, J); SMX_HIP(hipGetLastError()); }
        }
        for (size_t s0 = 0; s0 < S; s0 += (size_t)Sc) {
          const int Sn = (int)std::min<size_t>((size_t)Sc, S - s0);
          const long rows = (long)Sn * B;
          ScoreDrawArgs d;
          d.lat = m->latbuf; d.ld = 2 * Dp; d.B = B; d.D = m->D; d.Dp = Dp; d.S = Sn; d.s0 = (int)s0;
          d.nk = make_key(m, ST_EPS_Z, 0, false); d.rows = ps.rows; d.cell_base = ps.cell_base; d.z = zst; d.lw = lwst;
          SMX_CHECK(launch_score_draws(m->st, d));
          const float* hl = nullptr; int hld = 0;
          SMX_CHECK(stacked_decoder(m, zst, rows, hb, 0, nullptr, &hl, &hld));
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
          if (s_xp) {
            GemmArgs g;
            g.A = hl; g.lda = hld; g.B = P_(m, m->t_outW[0]); g.ldb = m->tensors[m->t_outW[0]].ld;
            g.C = Pst; g.ldc = (int)ldp; g.M = (int)rows; g.N = (int)ldp; g.K = hld; g.bias = P_(m, m->t_outb[0]); g.split_k = 1;
            SMX_CHECK(launch_gemm(m->st, g));
            for (size_t c = 0; c < k; ++c)
              addr(s_xp + ((s0 * k + c) * Cn + b0) * G, G, k * Cn * G, Pst + c * (size_t)m->Gp, ldp, (size_t)B * ldp, G);
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
        SMX_CHECK(forward_pass(m, ps, false, false, s == 0 ? 0 : 2));
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
          add(s_zm ? s_zm + b0 * D : nullptr, D, m->latbuf, (size_t)lat_ld, D);
          add(s_zs ? s_zs + b0 * D : nullptr, D, m->sig, (size_t)Dp, D);
          add(s_lm ? s_lm + b0 : nullptr, 1, m->latlbuf, 32, 1);
          add(s_ls ? s_ls + b0 : nullptr, 1, m->lsig, 1, 1);
        }
        add(s_zd ? s_zd + (s * Cn + b0) * D : nullptr, D, m->z, (size_t)Dp, D);
        add(s_ld ? s_ld + s * Cn + b0 : nullptr, 1, m->lsmp, 1, 1);
        if (s_xp)
          for (size_t c = 0; c < k; ++c) add(s_xp + ((s * k + c) * Cn + b0) * G, G, m->P + c * (size_t)m->Gp, k * (size_t)m->Gp, G);
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

// Stacked form (smx_score.hip): the encoder runs once, then the S draws of the B cells go through the decoder and the
// output head as S * B rows at a time (scvi: its library latent drawn per row as well, the raw planes materialised and a
// row-local softmax + likelihood launch, since its rate is normalised over all genes of a row; SCALE: its mixture prior
// in the latent part of log w).
static bool stacked_scoring_ok(const smx_model* m) {
  if (!m->flags.stacked_scoring || !m->stochastic || m->use_injected || m->dec.empty()) return false;
  if (m->scale && (m->Dp > 64 || m->cfg.n_components > 32)) return false;
  if (m->scvi && !scvi_score_supported(m->Gp)) return false;
  if (!head_loss_supported(1, m->dec.back().out_p, m->Gp) || (m->dec.back().out_p % 4)) return false;
  for (const MlpLayer& L : m->dec)
    if ((L.in_p % 4) || (L.out_p % 32)) return false;
  return m->dec[0].in_p == m->Dp;
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
  const long cap_rows = getenv("SMX_SCORE_ROWS") ? std::max(1L, atol(getenv("SMX_SCORE_ROWS"))) : (m->scvi ? 4096L : 16384L);
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
  const bool wide_head = m->scvi || getenv("SMX_SCORE_HEAD_WIDE") != nullptr || !score_head_supported(m->dec.back().out_p, m->Gp);   // the training kernel's direct-operand form (A/B)
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
      m->score_wimg = nullptr; m->score_wimg_floats = 0;
      SMX_CHECK(dmalloc(&m->score_wimg, wneed));
      m->score_wimg_floats = wneed;
    }
    __bf16* at = reinterpret_cast<__bf16*>(m->score_wimg);
    for (int np = 2; np <= 3; ++np) {
      if (!use_np[np]) continue;
      ScoreSplitWArgs sw;
      sw.W = P_(m, m->t_outW[0]); sw.ldw = m->tensors[m->t_outW[0]].ld; sw.Gp = m->Gp; sw.n_gt = n_gt; sw.nslab = nslab; sw.NP = np;
      sw.img = at;
      SMX_CHECK(launch_score_split_w(m->st, sw));
      wimg[np] = at;
      at += per_plane * np;
    }
  }
  // the encoders and the latent heads
  SMX_CHECK(forward_pass(m, ps, false, false, 3));
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
    SMX_CHECK(launch_score_draws(m->st, d));
    const float* in = nullptr;
    int ld = 0;
    SMX_CHECK(stacked_decoder(m, z, rows, hb, m->scvi ? 0 : (wide_head ? 1 : 2), ht, &in, &ld));
    if (m->scvi) {
      for (int ch = 0; ch < m->k; ++ch) {
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

int smx_marginal_llk(smx_model* m, const int32_t* row_ids, const float* host_x, const float* host_library, int32_t batch,
                     int32_t n_samples, float* mllk, float* llk_mean) {
  SMX_REQUIRE(m && mllk && n_samples > 0, "bad arguments");
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
    a.klmc = m->scale ? m->kl : nullptr;
    hipLaunchKernelGGL(iw_accum_kernel, dim3((batch + 3) / 4), dim3(256), 0, m->st, a);
  }
  if (rc == SMX_OK) {
    std::vector<float> h((size_t)3 * batch);
    hipError_t e = hipMemcpyAsync(h.data(), run, h.size() * sizeof(float), hipMemcpyDeviceToHost, m->st);
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
    std::vector<float> h((size_t)n_targets * 2 * 2 * batch);
    e = hipMemcpyAsync(h.data(), run, h.size() * sizeof(float), hipMemcpyDeviceToHost, m->st);
    if (e == hipSuccess) e = hipStreamSynchronize(m->st);
    if (e != hipSuccess) { set_error(std::string("score_llk readback failed: ") + hipGetErrorString(e)); rc = SMX_ERR_HIP; }
    else
      for (int t = 0; t < n_targets; ++t)
        for (int j = 0; j < 2; ++j) {
          const float* r = h.data() + ((size_t)t * 2 + (j < n_dist ? j : 0)) * 2 * batch;
          for (int b = 0; b < batch; ++b)
            out[((size_t)t * 2 + j) * batch + b] = r[b] + logf(r[batch + b]) - logf((float)n_samples);
        }
  } else {
    hipStreamSynchronize(m->st);
  }
  return rc;
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

int smx_comm_unique_id(uint8_t id[128]) {
  SMX_CHECK(load_rccl());
  ncclUniqueId uid;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclResult_t r = g_rccl.GetUniqueId(&uid);
  if (r != ncclSuccess) { set_error("ncclGetUniqueId failed"); return SMX_ERR_COMM; }
  memcpy(id, &uid, 128);
  return SMX_OK;
}

static int comm_detach(smx_model* m) {   // leave whatever communicator the model is in
  if (m->st) SMX_HIP(hipStreamSynchronize(m->st));
  if (m->st_comm) SMX_HIP(hipStreamSynchronize(m->st_comm));
  if (m->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(m->comm);
  m->comm = nullptr;
  m->local.reset();
  m->rank = 0; m->world = 1;
  drop_graphs(m);
  return SMX_OK;
}

static int ensure_sync_buf(smx_model* m) {
  int wmax = 0;
  for (int w : m->bn_wp) wmax = std::max(wmax, w);
  const size_t need = (size_t)m->world * 2 * (size_t)wmax;
  if (!m->sync_bn || need <= m->sync_cap) return SMX_OK;
  if (m->sync_buf) hipFree(m->sync_buf);
  m->sync_buf = nullptr; m->sync_cap = 0;
  SMX_CHECK(dmalloc(&m->sync_buf, need));
  m->sync_cap = need;
  return SMX_OK;
}

int smx_comm_init(smx_model* m, int rank, int world, const uint8_t id[128]) {
  SMX_REQUIRE(m && id && world >= 1 && rank >= 0 && rank < world, "bad rank/world");
  SMX_CHECK(load_rccl());
  SMX_CHECK(comm_detach(m));
  ncclUniqueId uid;
  memcpy(&uid, id, 128);
  ncclComm_t comm = nullptr;
  ncclResult_t r = g_rccl.CommInitRank(&comm, world, uid, rank);
  if (r != ncclSuccess) {
    // RCCL may leave a half-built handle behind: it is NOT kept (smx_model_destroy must not hand it to CommDestroy)
    set_error(std::string("ncclCommInitRank failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
    return SMX_ERR_COMM;
  }
  m->comm = comm;
  m->rank = rank; m->world = world;
  m->dp_force = getenv("SMX_FORCE_ALLREDUCE") != nullptr;
  m->dp_two_buckets = getenv("SMX_DP_BUCKETS") != nullptr && atoi(getenv("SMX_DP_BUCKETS")) == 2;
  if (!m->st_comm) {
    if (hipStreamCreateWithFlags(&m->st_comm, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_c1, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_c2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_c3, hipEventDisableTiming) != hipSuccess) {
      set_error("communication stream creation failed");
      return SMX_ERR_HIP;
    }
  }
  SMX_CHECK(ensure_sync_buf(m));
  drop_graphs(m);
  return SMX_OK;
}

int smx_comm_init_local(smx_model* const* models, int n) {
  SMX_REQUIRE(models && n >= 1 && n <= SMX_LOCAL_MAX, "loopback communicator: 1..8 models");
  for (int r = 0; r < n; ++r) {
    SMX_REQUIRE(models[r], "null model");
    SMX_REQUIRE(models[r]->device == models[0]->device, "loopback communicator: all models on one device");
    SMX_REQUIRE(models[r]->grads_count == models[0]->grads_count, "loopback communicator: models differ");
    for (int q = 0; q < r; ++q) SMX_REQUIRE(models[q] != models[r], "loopback communicator: a model is listed twice");
  }
  auto g = std::make_shared<LocalGroup>();
  g->world = n;
  for (int r = 0; r < n; ++r) {
    SMX_HIP(hipEventCreateWithFlags(&g->ready[r], hipEventDisableTiming));
    SMX_HIP(hipEventCreateWithFlags(&g->done[r], hipEventDisableTiming));
  }
  for (int r = 0; r < n; ++r) {
    smx_model* m = models[r];
    SMX_CHECK(comm_detach(m));
    m->rank = r; m->world = n; m->local = g;
    int wmax = 0;
    for (int w : m->bn_wp) wmax = std::max(wmax, w);
    const size_t need = std::max(m->grads_count, (size_t)n * 2 * (size_t)wmax);
    if (need > m->local_scratch_cap) {
      if (m->local_scratch) hipFree(m->local_scratch);
      m->local_scratch = nullptr; m->local_scratch_cap = 0;
      SMX_CHECK(dmalloc(&m->local_scratch, need));
      m->local_scratch_cap = need;
    }
    SMX_CHECK(ensure_sync_buf(m));
  }
  return SMX_OK;
}

int smx_comm_set_sync_bn(smx_model* m, int on) {
  SMX_REQUIRE(m, "null model");
  SMX_HIP(hipStreamSynchronize(m->st));
  m->sync_bn = on != 0;
  SMX_CHECK(ensure_sync_buf(m));
  drop_graphs(m);
  return SMX_OK;
}

int smx_comm_library(char* rccl_path, int rccl_cap, char* hip_path, int hip_cap, int32_t* rccl_version) {
  SMX_CHECK(load_rccl());
  if (rccl_path && rccl_cap > 0) { strncpy(rccl_path, g_rccl.path.c_str(), rccl_cap - 1); rccl_path[rccl_cap - 1] = 0; }
  if (hip_path && hip_cap > 0) { strncpy(hip_path, g_rccl.hip_path.c_str(), hip_cap - 1); hip_path[hip_cap - 1] = 0; }
  if (rccl_version) { int v = 0; if (g_rccl.GetVersion) g_rccl.GetVersion(&v); *rccl_version = v; }
  return SMX_OK;
}

int smx_comm_rank(const smx_model* m) { return m ? m->rank : 0; }
int smx_comm_world(const smx_model* m) { return m ? m->world : 0; }

int smx_set_flag(smx_model* m, const char* name, int value) {
  SMX_REQUIRE(m && name, "null argument");
  SMX_HIP(hipStreamSynchronize(m->st));
  const std::string n(name);
  int* f = n == "head_loss" ? &m->flags.head_loss : n == "front" ? &m->flags.front : n == "bwd_front" ? &m->flags.bwd_front
         : n == "head_bwd" ? &m->flags.head_bwd : n == "wgrad" ? &m->flags.wgrad : n == "scvi_fused" ? &m->flags.scvi_fused
         : n == "twin" ? &m->flags.twin : n == "label_ride" ? &m->flags.label_ride : n == "act_epilogue" ? &m->flags.act_epilogue
         : n == "stacked_scoring" ? &m->flags.stacked_scoring : nullptr;
  SMX_REQUIRE(f, "unknown flag (head_loss, front, bwd_front, head_bwd, wgrad, scvi_fused, twin, label_ride, act_epilogue, stacked_scoring)");
  *f = value ? 1 : 0;
  drop_graphs(m);   // a captured step bakes the launch sequence in
  return SMX_OK;
}

int smx_timing_enable(smx_model* m, const char* kernel) {
  SMX_REQUIRE(m, "null model");
  SMX_HIP(hipStreamSynchronize(m->st));
  m->timing_label = kernel ? kernel : "";
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

// ---- kernel-level entry points ------------------------------------------------
int smx_k_count_llk(int likelihood, int direct, const float* x, const float* planes, int32_t B, int32_t G, float* llk,
                    float* grads) {
  SMX_REQUIRE(x && planes && llk && B > 0 && G > 0, "bad arguments");
  const int k = (likelihood == SMX_LLK_ZINB || likelihood == SMX_LLK_ZINBD) ? 3 : 2;
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
  // pad to the library's internal conventions: feature axes to 32, batch axes free
  const int Np = round_up(N, 32);
  const int Kp = round_up(K, 4), Mp = round_up(M, 4);
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
  rc = launch_gemm(nullptr, g, &eff);
  if (rc == SMX_OK && getenv("SMX_KGEMM_REPS")) {  // diagnostic: average launch time of this shape / tile
    const int reps = atoi(getenv("SMX_KGEMM_REPS"));
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
