// smx_comm.hip -- data parallel: RCCL resolution, the loopback communicator of the tests, the all-reduce of the flat buffer.
#include "smx_model.h"

namespace smx {

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
  g_rccl.CommSplit = (decltype(g_rccl.CommSplit))dlsym(h, "ncclCommSplit");
  g_rccl.ReduceScatter = (decltype(g_rccl.ReduceScatter))dlsym(h, "ncclReduceScatter");
  g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
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

struct LocalSrc { const float* p[SMX_LOCAL_MAX]; int n; };
__global__ void local_sum_kernel(LocalSrc s, float* dst, size_t count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    float acc = s.p[0][i];
    for (int r = 1; r < s.n; ++r) acc += s.p[r][i];
    dst[i] = acc;
  }
}

// The hand-written exchange (smx_p2p.hip) takes the step's collectives when it is attached and either asked for (form 3) or the only one
// there is; under the library's own rule (form 0) it keeps the precedence over RCCL it has had since round 3 (SMX_ALLREDUCE=p2p).
bool p2p_selected(const smx_model* m) {
  if (!(m->p2p && m->p2p->error)) return false;
  return m->dp_form == 0 || m->dp_form == 3 || !m->comm;
}
// data-parallel overlap: two buckets on a communication stream (eager launches only)
bool dp_active(const smx_model* m) {
  // dp_force: exercise RCCL on a 1-rank communicator (tests); local: the loopback communicator of the tests
  return (m->comm && (m->world > 1 || m->dp_force)) || (m->local && m->world > 1) || (m->p2p && m->p2p->error && (m->world > 1 || m->dp_force));
}
// Two buckets: the heads' gradients (3/4 of the bytes, final long before the rest) on the communication stream, the rest on the
// model's stream.  Taken from SMX_DP_BUCKETS_MIN_BYTES of head gradients (3 MB: BASELINE configs[1] sits right at it) unless
// SMX_DP_BUCKETS says 1 or 2.  Round 4's form -- BOTH buckets on the communication stream, the optimiser behind an event of that stream --
// cost +32-35 us on ONE rank (profiles/r04_dp_overhead_one_rank.txt): two cross-queue hops (main -> comm -> main) on the critical path of
// every step.  Since round 5 the two-bucket step is a CHAIN (smx_step.hip: dp_chain_start): head bucket all-reduce -> its norms -> the
// heads' clip + Adam, all on the communication stream and joined in front of the NEXT step's output head; the main stream all-reduces the
// front bucket itself and never waits for the other queue inside a step.
bool dp_overlap(const smx_model* m) {
  // Not with the hand-written exchange under SyncBatchNorm: the head bucket's exchange on st_comm would run beside the
  // SyncBatchNorm-backward exchanges on the model's stream, and both go through ONE staging buffer, ONE done counter and ONE
  // monotonic flag set (REDUCED(e + 1) satisfies waiters on e; staging overwritten mid-gather -- ADVICE r03).  RCCL serialises per
  // communicator and keeps the overlap.
  if (p2p_selected(m) && m->sync_bn) return false;
  return dp_active(m) && m->dp_two_buckets && !m->capturing && m->st_comm != nullptr && !m->local;
}
// the chained form: RCCL (the heads' bucket on its own communicator when ncclCommSplit gave one) or the tests' loopback communicator; the
// hand-written exchange keeps round 4's form (one staging buffer, one flag set: its two buckets cannot be in flight together)
bool dp_chain_ok(const smx_model* m) {
  if (!dp_active(m) || !(m->dp_two_buckets || m->flags.opt_shard) || m->capturing || !m->st_comm || p2p_selected(m)) return false;
  return m->bucket1_count > 0 && m->chunk_first_head < m->n_chunks;
}
int local_allreduce(smx_model* m, float* buf, size_t count, hipStream_t st) {
  LocalGroup& g = *m->local;
  const int me = m->rank;
  // (a bucket of the flat gradients takes the same stretch of the scratch: the heads' bucket on the communication stream and a collective
  // of the main stream may be under way together)
  size_t soff = (buf >= m->grads && buf < m->grads + m->grads_count) ? (size_t)(buf - m->grads) : 0;
  if (buf >= m->shard_partial && buf < m->shard_partial + m->n_chunks) soff = m->grads_count + (size_t)(buf - m->shard_partial);   // (flag opt_shard: the chunks' partial sums, on the communication stream)
  SMX_REQUIRE(soff + count <= m->local_scratch_cap, "loopback all-reduce: scratch too small");
  float* const scratch = m->local_scratch + soff;
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
  hipLaunchKernelGGL(local_sum_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, scratch, count);
  SMX_HIP(hipEventRecord(g.done[me], st));
  if (!g.barrier()) { set_error("loopback communicator: a member did not arrive (timeout)"); return SMX_ERR_COMM; }
  for (int r = 0; r < g.world; ++r)
    if (r != me) SMX_HIP(hipStreamWaitEvent(st, g.done[r], 0));   // nobody still reads this rank's buffer
  SMX_HIP(hipMemcpyAsync(buf, scratch, count * sizeof(float), hipMemcpyDeviceToDevice, st));
  return SMX_OK;
}
int dp_allreduce_buf(smx_model* m, float* buf, size_t count, hipStream_t st, bool second) {
  if (m->local) return local_allreduce(m, buf, count, st);
  if (p2p_selected(m)) return p2p_allreduce(m, buf, count, st);
  ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclFloat32, ncclSum, (second && m->comm2) ? m->comm2 : m->comm, st);
  if (r != ncclSuccess) {
    set_error(std::string("ncclAllReduce failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
    return SMX_ERR_COMM;
  }
  return SMX_OK;
}
int dp_allreduce(smx_model* m, size_t off, size_t count, hipStream_t st, bool second) { return dp_allreduce_buf(m, m->grads + off, count, st, second); }

// ---- flag opt_shard: the head bucket as world slices of `slice` floats each (in place: rank r's slice is buf + r * slice) -------------------
// loopback: every rank sums ITS slice over the ranks' buffers in rank order (the all-reduce's order: the same sums) into its own buffer --
// the other ranks read only their own slices of it --, and gathers by copying the owners' slices
static int local_slices(smx_model* m, float* buf, size_t slice, hipStream_t st, bool gather) {
  LocalGroup& g = *m->local;
  const int me = m->rank;
  { std::lock_guard<std::mutex> lk(g.mu); g.src[me] = buf; }
  SMX_HIP(hipEventRecord(g.ready[me], st));
  if (!g.barrier()) { set_error("loopback communicator: a member did not arrive (timeout)"); return SMX_ERR_COMM; }
  for (int r = 0; r < g.world; ++r)
    if (r != me) SMX_HIP(hipStreamWaitEvent(st, g.ready[r], 0));
  if (gather) {
    for (int r = 0; r < g.world; ++r)
      if (r != me) SMX_HIP(hipMemcpyAsync(buf + (size_t)r * slice, g.src[r] + (size_t)r * slice, slice * sizeof(float), hipMemcpyDeviceToDevice, st));
  } else {
    LocalSrc src;
    src.n = g.world;
    for (int r = 0; r < g.world; ++r) src.p[r] = g.src[r] + (size_t)me * slice;
    const unsigned blocks = (unsigned)std::min<size_t>((slice + 255) / 256, 2048);
    hipLaunchKernelGGL(local_sum_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, buf + (size_t)me * slice, slice);   // (in place: element i of this rank's slice is read and written by one thread)
  }
  SMX_HIP(hipEventRecord(g.done[me], st));
  if (!g.barrier()) { set_error("loopback communicator: a member did not arrive (timeout)"); return SMX_ERR_COMM; }
  for (int r = 0; r < g.world; ++r)
    if (r != me) SMX_HIP(hipStreamWaitEvent(st, g.done[r], 0));   // nobody still reads this rank's buffer
  return SMX_OK;
}
bool dp_shard_available(const smx_model* m) {
  if (m->local) return true;
  return m->comm && g_rccl.ReduceScatter && g_rccl.AllGather && !p2p_selected(m);
}
int dp_reduce_scatter(smx_model* m, float* buf, size_t slice, hipStream_t st, bool second) {
  if (m->local) return local_slices(m, buf, slice, st, false);
  ncclResult_t r = g_rccl.ReduceScatter(buf, buf + (size_t)m->rank * slice, slice, ncclFloat32, ncclSum, (second && m->comm2) ? m->comm2 : m->comm, st);
  if (r != ncclSuccess) { set_error(std::string("ncclReduceScatter failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?")); return SMX_ERR_COMM; }
  return SMX_OK;
}
int dp_all_gather(smx_model* m, float* buf, size_t slice, hipStream_t st, bool second) {
  if (m->local) return local_slices(m, buf, slice, st, true);
  ncclResult_t r = g_rccl.AllGather(buf + (size_t)m->rank * slice, buf, slice, ncclFloat32, (second && m->comm2) ? m->comm2 : m->comm, st);
  if (r != ncclSuccess) { set_error(std::string("ncclAllGather failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?")); return SMX_ERR_COMM; }
  return SMX_OK;
}

}  // namespace smx

extern "C" {

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
  p2p_release(m);
  if (m->comm2 && g_rccl.CommDestroy) g_rccl.CommDestroy(m->comm2);
  m->comm2 = nullptr;
  if (m->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(m->comm);
  m->comm = nullptr;
  m->local.reset();
  m->rank = 0; m->world = 1;
  m->dp_form = 0;
  drop_graphs(m);
  return SMX_OK;
}

}  // extern "C"
namespace smx {
// the communication stream of the two-bucket form (the heads' gradients, 3/4 of the bytes and final long before the rest, are reduced
// -- and applied -- beside the rest of the step) and the switches read when a communicator is attached
// Which exchange the steps take.  Asked for (smx_comm_set_form 1 / 2 / 3: what sisua_amd/parallel.py's calibration measured fastest on
// every rank, or SMX_DP_FORM) -- or, with nothing asked (form 0), the rule of rounds 4-5: SMX_DP_BUCKETS if set, else two buckets from
// SMX_DP_BUCKETS_MIN_BYTES of head gradients.  That constant is a guess that BASELINE configs[1] sits right on: it only
// decides when nobody measured.
void apply_dp_form(smx_model* m) {
  const char* nb = getenv("SMX_DP_BUCKETS");
  if (m->dp_form == 1 || m->dp_form == 3) m->dp_two_buckets = false;
  else if (m->dp_form == 2) m->dp_two_buckets = true;
  else m->dp_two_buckets = nb ? atoi(nb) == 2 : m->bucket1_count * sizeof(float) >= SMX_DP_BUCKETS_MIN_BYTES;
}
int ensure_comm_stream(smx_model* m) {
  m->dp_force = getenv("SMX_FORCE_ALLREDUCE") != nullptr;
  apply_dp_form(m);
  if (!m->st_comm) {
    if (hipStreamCreateWithFlags(&m->st_comm, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_c1, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_c2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_c3, hipEventDisableTiming) != hipSuccess) {
      set_error("communication stream creation failed");
      return SMX_ERR_HIP;
    }
  }
  return SMX_OK;
}
// the heads' bucket's own communicator (a COLLECTIVE call: every rank, in the same order)
void ensure_comm2(smx_model* m) {
  if (!m->comm || m->comm2 || !m->dp_two_buckets || !g_rccl.CommSplit || getenv("SMX_DP_ONE_COMM")) return;
  ncclComm_t c2 = nullptr;
  if (g_rccl.CommSplit(m->comm, 0, m->rank, &c2, nullptr) == ncclSuccess && c2) m->comm2 = c2;
}
int ensure_sync_buf(smx_model* m) {
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
}  // namespace smx
extern "C" {

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
  SMX_CHECK(ensure_comm_stream(m));
  // the heads' bucket on a communicator of its own (same ranks, same order): RCCL serialises the operations of ONE communicator in issue
  // order whatever their streams, which would put the front bucket's all-reduce behind the heads' 30 MB.  Without ncclCommSplit (or when it
  // fails) both buckets share `comm`: still correct, the overlap is then what the backward pass covers.
  ensure_comm2(m);
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
    const size_t need = std::max(m->grads_count + (size_t)m->n_chunks, (size_t)n * 2 * (size_t)wmax);
    if (need > m->local_scratch_cap) {
      if (m->local_scratch) hipFree(m->local_scratch);
      m->local_scratch = nullptr; m->local_scratch_cap = 0;
      SMX_CHECK(dmalloc(&m->local_scratch, need));
      m->local_scratch_cap = need;
    }
    SMX_CHECK(ensure_sync_buf(m));
    SMX_CHECK(ensure_comm_stream(m));   // (the chained two-bucket form runs on the loopback too: smx_step.hip, dp_chain_start)
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

int smx_comm_time_allreduce(smx_model* m, int iters, float* us_per_call, int64_t* floats) {
  SMX_REQUIRE(m && us_per_call && iters >= 1 && iters <= 100000, "bad arguments");
  SMX_REQUIRE(smx::dp_active(m), "no communicator attached (or a world of one without SMX_FORCE_ALLREDUCE)");
  SMX_HIP(hipStreamSynchronize(m->st));
  SMX_HIP(hipMemsetAsync(m->grads, 0, m->grads_count * sizeof(float), m->st));   // (sums of zeros: no overflow however often it runs)
  for (int i = 0; i < 3; ++i) SMX_CHECK(smx::dp_allreduce(m, 0, m->grads_count, m->st));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  SMX_HIP(hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); smx::set_error("hipEventCreate failed"); return SMX_ERR_HIP; }
  int rc = SMX_OK;
  float ms = 0.f;
  if (hipEventRecord(e0, m->st) != hipSuccess) rc = SMX_ERR_HIP;
  for (int i = 0; i < iters && rc == SMX_OK; ++i) rc = smx::dp_allreduce(m, 0, m->grads_count, m->st);
  if (rc == SMX_OK && (hipEventRecord(e1, m->st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)) {
    smx::set_error("event timing of the all-reduce failed");
    rc = SMX_ERR_HIP;
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  if (rc == SMX_OK) { *us_per_call = 1e3f * ms / (float)iters; if (floats) *floats = (int64_t)m->grads_count; }
  return rc;
}

int smx_comm_rank(const smx_model* m) { return m ? m->rank : 0; }
int smx_opt_gather(smx_model* m) {
  SMX_REQUIRE(m, "null model");
  SMX_HIP(hipStreamSynchronize(m->st));
  if (m->st_comm) SMX_HIP(hipStreamSynchronize(m->st_comm));
  if (!m->opt_stale) return SMX_OK;
  SMX_REQUIRE(smx::dp_active(m) && smx::dp_shard_available(m), "opt_gather: the communicator the sharded steps ran on is gone");
  const size_t slice = (((size_t)m->bucket1_count + m->world - 1) / m->world + 63) / 64 * 64;
  SMX_CHECK(smx::dp_all_gather(m, m->adam_m + m->bucket1_off, slice, m->st));
  SMX_CHECK(smx::dp_all_gather(m, m->adam_v + m->bucket1_off, slice, m->st));
  SMX_HIP(hipStreamSynchronize(m->st));
  m->opt_stale = false;
  return SMX_OK;
}
int smx_comm_form(const smx_model* m) {
  if (!m || !smx::dp_active(m)) return 0;
  if (smx::dp_chain_ok(m)) return 2;
  if (smx::p2p_selected(m)) return smx::dp_overlap(m) ? 4 : 3;
  return 1;
}
int smx_comm_set_form(smx_model* m, int form) {
  SMX_REQUIRE(m && form >= 0 && form <= 3, "comm_set_form: 0 (the library's rule), 1 (one all-reduce), 2 (two-bucket chain) or 3 (hand-written exchange)");
  SMX_REQUIRE(form != 3 || (m->p2p && m->p2p->error), "comm_set_form: the hand-written exchange is not attached (smx_comm_p2p_export / _init)");
  SMX_REQUIRE(form == 0 || form == 3 || m->comm || m->local, "comm_set_form: no RCCL / loopback communicator attached");
  if (m->st) SMX_HIP(hipStreamSynchronize(m->st));
  if (m->st_comm) SMX_HIP(hipStreamSynchronize(m->st_comm));
  if (m->opt_stale) SMX_CHECK(smx_opt_gather(m));   // (sharded moments are whole again before the form that shards them can go)
  m->dp_form = form;
  smx::apply_dp_form(m);
  smx::ensure_comm2(m);
  drop_graphs(m);
  return SMX_OK;
}
int smx_comm_world(const smx_model* m) { return m ? m->world : 0; }

}  // extern "C"
