// smx_p2p.hip -- the data-parallel all-reduce as a hand-written two-shot exchange over peer-mapped buffers (SURVEY.md 5,
// "Distributed comm backend": the flat-buffer plan; BASELINE.json north_star: "a single all-reduce of gradients over xGMI per
// step").  The reference has no counterpart (sisua/train.py:20 pins one device).
//
// One process per GPU.  Every rank exports two allocations through HIP IPC -- its flat gradient buffer G_r and a small
// communication region R_r = [flags | staging S_r | scratch X_r] -- and maps its peers' (xGMI: a peer pointer is a load /
// store target like local memory).  An all-reduce of G over `world` ranks, chunk c = the c-th 1 / world of the buffer:
//
//   phase A   signal READY(e) to every peer; wait for every peer's READY(e)        -- all backward passes have written G
//             S_r = G_r[chunk r] = sum over q (rank order) of G_q[chunk r]         -- reduce-scatter: reads 1/world of every peer
//             the last workgroup to finish signals REDUCED(e) to every peer
//   phase B   for every q: wait for REDUCED(e) of q; G_r[chunk q] = S_q            -- all-gather: reads every peer's staging
//
// ONE launch (round 6: the phases were two launches before), two flag rounds, each byte crosses a link twice (in, as 1/world pieces from 7 peers at once: the bandwidth of
// all links together).  No exit round: S_r is rewritten only in phase A(e + 1), behind READY(e + 1) of every peer -- which a
// peer raises after its own launch of epoch e has completed on its stream; G_r is read by peers only between READY(e) and
// REDUCED(e).  Summation in rank order on every rank: bitwise the same result everywhere, run to run.
// Flags are monotonic epochs written with system-scope release stores and polled with system-scope acquire loads; every wait
// is BOUNDED (a peer that died must not hang the device: after ~2 s the wait gives up and raises an error word the host
// reads at the next synchronising call).  Buffers other than G (SyncBatchNorm's statistics) take the same path through X_r.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "smx_model.h"

namespace smx {

struct P2PArgs {
  const float* g[SMX_P2P_MAX];       // every rank's buffer to reduce (own: local pointer), same length
  float* s[SMX_P2P_MAX];             // every rank's staging (own: local)
  unsigned* flags[SMX_P2P_MAX];      // every rank's flag block [2][SMX_P2P_MAX] (READY | REDUCED by source rank), own: local
  float* out;                        // where the sum goes (== g[rank] for the in-place form)
  unsigned* done;                    // local: workgroups of launch A that have finished
  unsigned* error;                   // local: non-zero after a timed-out wait of this rank (1) or a peer's report of one (2); sticky
  unsigned* peer_error[SMX_P2P_MAX]; // every rank's error word (peers: in their IPC-mapped regions): a failing rank poisons them all
  long long timeout_ticks;           // bound of every wait on the device's 100 MHz wall clock (SMX_P2P_TIMEOUT_S, default 30 s)
  long count, chunk;                 // floats in all / per rank (chunk a multiple of 4)
  int rank, world; unsigned epoch;
};

// A wait gives up when its bound passes (error word := 1) or as soon as the rank's error word is set by somebody else -- a peer whose
// own wait failed poisons every rank's word (2), so one straggler / dead peer fails the step on EVERY rank instead of leaving the others
// to gather stale staging.  The word is sticky: the host reads it at the next metrics read-back (SMX_ERR_COMM).
__device__ inline bool wait_flag(const unsigned* f, unsigned epoch, unsigned* error, long long timeout_ticks) {
  const long long t0 = wall_clock64();   // 100 MHz on gfx9
  while ((int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
    __builtin_amdgcn_s_sleep(8);
    if (__hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return false;
    if (wall_clock64() - t0 > timeout_ticks) { __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return false; }
  }
  return true;
}
__device__ inline void poison_peers(const P2PArgs& a) {   // (threads [0, world) of a workgroup)
  if (threadIdx.x < (unsigned)a.world && (int)threadIdx.x != a.rank)
    __hip_atomic_store(a.peer_error[threadIdx.x], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ONE launch per all-reduce (round 6; rounds 3-5 ran the two phases as two launches: +22 us per C2 step on one rank before a byte crossed a
// link).  The protocol is the header's, flag for flag -- only the kernel boundary between the reduce-scatter and the all-gather is gone:
// a workgroup that has finished its part of S_r goes on to fetch the peers' reduced chunks, each behind that peer's REDUCED(e).  Nothing is
// read or written earlier or by anybody else than before: G_r[chunk q] is overwritten only after REDUCED(e) of q, which q raises when ALL
// its reads of chunk q are done; this rank's own chunk is written in place by the thread that read it (no peer reads G_r[chunk r]).
// Every rank's launch must be resident while its peers' are (as before: <= 128 workgroups); the waits stay bounded.
template <int W>   // W = world (compile-time: the per-rank load arrays stay in registers)
__global__ __launch_bounds__(256) void p2p_exchange_kernel(P2PArgs a) {
  __shared__ int ok_s;
  if (threadIdx.x < (unsigned)a.world) {
    const int q = threadIdx.x;
    if (blockIdx.x == 0 && q != a.rank)   // READY(e): "my G is final" into peer q's block, slot [0][rank]
      __hip_atomic_store(a.flags[q] + a.rank, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (threadIdx.x == 0) ok_s = 1;
  __syncthreads();
  if (threadIdx.x < (unsigned)a.world && (int)threadIdx.x != a.rank)
    if (!wait_flag(a.flags[a.rank] + threadIdx.x, a.epoch, a.error, a.timeout_ticks)) ok_s = 0;
  __syncthreads();
  if (!ok_s) poison_peers(a);
  const long stride = (long)gridDim.x * 256;
  if (ok_s && W > 1) {   // (a world of one: the sum over one rank is the buffer itself)
    const long base = (long)a.rank * a.chunk;
    const long n4 = std::max<long>(0, std::min(a.chunk, a.count - base)) >> 2;
    // four positions x every rank in flight per thread (a link's latency x bandwidth is ~150 KB: one 16-byte load per thread of
    // a small grid would leave the links idle most of the time); the sum itself in rank order
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 4 * stride) {
      float4 v[W][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < W; ++q) {
          const long i = i0 + u * stride;
          v[q][u] = i < n4 ? reinterpret_cast<const float4*>(a.g[q] + base)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = i0 + u * stride;
        if (i >= n4) break;
        float4 acc = v[0][u];
#pragma unroll
        for (int q = 1; q < W; ++q) { acc.x += v[q][u].x; acc.y += v[q][u].y; acc.z += v[q][u].z; acc.w += v[q][u].w; }
        reinterpret_cast<float4*>(a.s[a.rank])[i] = acc;          // for the peers
        reinterpret_cast<float4*>(a.out + base)[i] = acc;         // this rank's own copy of its chunk
      }
    }
  }
  // REDUCED(e) once every workgroup's part of S is visible system-wide
  __threadfence_system();
  __syncthreads();
  __shared__ int last_s;
  if (threadIdx.x == 0) last_s = (atomicAdd(a.done, 1u) == gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (last_s) {
    if (threadIdx.x == 0) *a.done = 0;
    // (a workgroup whose wait failed set the error word before it counted itself done: S_r is incomplete, REDUCED(e) is NOT raised)
    const bool failed = __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
    if (!failed && threadIdx.x < (unsigned)a.world && (int)threadIdx.x != a.rank)
      __hip_atomic_store(a.flags[threadIdx.x] + SMX_P2P_MAX + a.rank, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (!ok_s) return;
  // all-gather: the peers' reduced chunks, nearest rank first (rank + 1, rank + 2, ...: the ranks do not all start on the same source)
  for (int j = 1; j < W; ++j) {
    const int q = (a.rank + j) % W;
    __syncthreads();
    if (threadIdx.x == 0) ok_s = wait_flag(a.flags[a.rank] + SMX_P2P_MAX + q, a.epoch, a.error, a.timeout_ticks) ? 1 : 0;
    __syncthreads();
    if (!ok_s) { poison_peers(a); return; }
    const long base = (long)q * a.chunk;
    const long n4 = std::max<long>(0, std::min(a.chunk, a.count - base)) >> 2;
    const float4* src = reinterpret_cast<const float4*>(a.s[q]);
    float4* dst = reinterpret_cast<float4*>(a.out + base);
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 4 * stride) {   // four loads in flight per thread
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const long i = i0 + u * stride; v[u] = i < n4 ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const long i = i0 + u * stride; if (i < n4) dst[i] = v[u]; }
    }
  }
}

// the all-reduce of `count` floats at `buf` (the registered gradient buffer or anything that fits the scratch) on stream st
int p2p_allreduce(smx_model* m, float* buf, size_t count, hipStream_t st) {
  P2PState& p = *m->p2p;
  SMX_REQUIRE((count % 4) == 0, "p2p all-reduce: the length must be a multiple of 4 floats");
  P2PArgs a;
  memset(&a, 0, sizeof(a));
  const bool is_g = buf >= m->grads && buf + count <= m->grads + m->grads_count;
  if (!is_g) {   // a small buffer (SyncBatchNorm statistics): through the exported scratch
    SMX_REQUIRE(count <= p.scratch_floats, "p2p all-reduce: buffer exceeds the exported scratch");
    SMX_HIP(hipMemcpyAsync(p.scratch[p.rank], buf, count * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  const size_t off = is_g ? (size_t)(buf - m->grads) : 0;
  for (int q = 0; q < p.world; ++q) {
    a.g[q] = is_g ? p.grads[q] + off : p.scratch[q];
    a.s[q] = p.staging[q];
    a.flags[q] = p.flags[q];
    a.peer_error[q] = p.flags[q] + 2 * SMX_P2P_MAX + 1;
  }
  a.timeout_ticks = p.timeout_ticks;
  a.out = is_g ? buf : p.scratch[p.rank];
  a.done = p.done; a.error = p.error;
  a.count = (long)count;
  a.chunk = (long)(((count + p.world - 1) / p.world + 3) / 4 * 4);
  SMX_REQUIRE((size_t)a.chunk <= p.staging_floats, "p2p all-reduce: staging too small");
  a.rank = p.rank; a.world = p.world; a.epoch = ++p.epoch;
  // few, fat workgroups: the exchange is link-bound and the flag rounds must be co-resident with whatever else runs
  // a fraction of the chip: the exchange is link-bound, and every rank's launches must be resident together (the flag rounds)
  const unsigned nb = (unsigned)std::max<long>(1, std::min<long>(128, (a.chunk / 4 + 1023) / 1024));
  switch (p.world) {
    case 1: hipLaunchKernelGGL(p2p_exchange_kernel<1>, dim3(1), dim3(256), 0, st, a); break;   // (flags only: one workgroup)
    case 2: hipLaunchKernelGGL(p2p_exchange_kernel<2>, dim3(nb), dim3(256), 0, st, a); break;
    case 3: hipLaunchKernelGGL(p2p_exchange_kernel<3>, dim3(nb), dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL(p2p_exchange_kernel<4>, dim3(nb), dim3(256), 0, st, a); break;
    case 5: hipLaunchKernelGGL(p2p_exchange_kernel<5>, dim3(nb), dim3(256), 0, st, a); break;
    case 6: hipLaunchKernelGGL(p2p_exchange_kernel<6>, dim3(nb), dim3(256), 0, st, a); break;
    case 7: hipLaunchKernelGGL(p2p_exchange_kernel<7>, dim3(nb), dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL(p2p_exchange_kernel<8>, dim3(nb), dim3(256), 0, st, a); break;
  }
  SMX_HIP(hipGetLastError());
  if (!is_g) SMX_HIP(hipMemcpyAsync(buf, p.scratch[p.rank], count * sizeof(float), hipMemcpyDeviceToDevice, st));
  return SMX_OK;
}

void p2p_release(smx_model* m) {
  if (!m->p2p) return;
  P2PState& p = *m->p2p;
  for (int q = 0; q < p.world; ++q) {
    if (q == p.rank) continue;
    if (p.grads[q]) hipIpcCloseMemHandle(p.grads[q]);
    if (p.region_base[q]) hipIpcCloseMemHandle(p.region_base[q]);
  }
  if (p.region_base[p.rank] && p.owns_region) hipFree(p.region_base[p.rank]);
  m->p2p.reset();
}

static void region_layout(smx_model* m, int world, size_t* staging_floats, size_t* scratch_floats, size_t* bytes) {
  const size_t chunk = ((m->grads_count + world - 1) / world + 3) / 4 * 4 + 64;
  size_t scratch = 4096;
  for (size_t i = 0; i < m->bn_wp.size(); ++i) scratch = std::max(scratch, (size_t)SMX_P2P_MAX * 2 * m->bn_wp[i]);
  *staging_floats = chunk; *scratch_floats = scratch;
  *bytes = 1024 + (chunk + scratch) * sizeof(float);   // [flags 2 x SMX_P2P_MAX words | done | error | pad to 1 KB][staging][scratch]
}

}  // namespace smx

extern "C" {

// this rank's two IPC handles: [0, 64) the flat gradient buffer, [64, 128) the communication region (allocated here)
int smx_comm_p2p_export(smx_model* m, int world, uint8_t handles[128]) {
  SMX_REQUIRE(m && handles && world >= 1 && world <= SMX_P2P_MAX, "bad arguments (world <= 8)");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  SMX_HIP(hipStreamSynchronize(m->st));
  p2p_release(m);
  m->p2p = std::make_shared<P2PState>();
  P2PState& p = *m->p2p;
  size_t bytes;
  region_layout(m, world, &p.staging_floats, &p.scratch_floats, &bytes);
  // Fine-grained device memory: peers STORE their flags into this region while this rank's waves poll it.  A coarse-grained
  // allocation is only promised to be coherent across agents at kernel boundaries (the owner's L2 may keep serving a polled line);
  // on one device -- where the tests run -- every process shares the L2 and either kind works, so the kind that is right on a
  // multi-GPU node is the one used everywhere.
  void* region = nullptr;
  if (hipExtMallocWithFlags(&region, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    region = nullptr;
    SMX_HIP(hipMalloc(&region, bytes));
  }
  SMX_HIP(hipMemset(region, 0, bytes));
  p.owns_region = true; p.pending_world = world;
  p.region_base[0] = region;   // (parked until the rank is known: smx_comm_p2p_init moves it to slot `rank`)
  hipIpcMemHandle_t hg, hr;
  SMX_HIP(hipIpcGetMemHandle(&hg, m->grads));
  SMX_HIP(hipIpcGetMemHandle(&hr, region));
  memcpy(handles, &hg, 64);
  memcpy(handles + 64, &hr, 64);
  return SMX_OK;
}

// all_handles [world][128]: what every rank's smx_comm_p2p_export returned, in rank order (gathered by the control plane)
int smx_comm_p2p_init(smx_model* m, int rank, int world, const uint8_t* all_handles) {
  SMX_REQUIRE(m && all_handles && world >= 1 && world <= SMX_P2P_MAX && rank >= 0 && rank < world, "bad arguments");
  SMX_REQUIRE(m->p2p && m->p2p->pending_world == world, "call smx_comm_p2p_export(world) first");
  SMX_HIP(hipStreamSynchronize(m->st));
  P2PState& p = *m->p2p;
  void* mine = p.region_base[0];
  p.region_base[0] = nullptr;
  p.rank = rank; p.world = world;
  auto carve = [&](int q, void* base) {
    p.region_base[q] = base;
    p.flags[q] = reinterpret_cast<unsigned*>(base);
    p.staging[q] = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(base) + 1024);
    p.scratch[q] = p.staging[q] + p.staging_floats;
  };
  for (int q = 0; q < world; ++q) {
    if (q == rank) { carve(q, mine); p.grads[q] = m->grads; continue; }
    hipIpcMemHandle_t hg, hr;
    memcpy(&hg, all_handles + (size_t)q * 128, 64);
    memcpy(&hr, all_handles + (size_t)q * 128 + 64, 64);
    void *pg = nullptr, *pr = nullptr;
    hipError_t e1 = hipIpcOpenMemHandle(&pg, hg, hipIpcMemLazyEnablePeerAccess);
    hipError_t e2 = e1 == hipSuccess ? hipIpcOpenMemHandle(&pr, hr, hipIpcMemLazyEnablePeerAccess) : e1;
    if (e1 != hipSuccess || e2 != hipSuccess) {
      set_error(std::string("hipIpcOpenMemHandle of rank ") + std::to_string(q) + " failed: " + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
      if (pg) hipIpcCloseMemHandle(pg);
      carve(rank, mine);
      p2p_release(m);
      return SMX_ERR_COMM;
    }
    p.grads[q] = reinterpret_cast<float*>(pg);
    carve(q, pr);
  }
  p.done = p.flags[rank] + 2 * SMX_P2P_MAX;
  p.error = p.done + 1;
  {   // bound of every wait: generous by default (ordinary rank skew -- a slow upload, a validation pass -- must not trip it)
    const char* ts = getenv("SMX_P2P_TIMEOUT_S");
    const double sec = ts ? atof(ts) : 30.0;
    p.timeout_ticks = (long long)(std::min(std::max(sec, 0.05), 3600.0) * 1e8);
  }
  m->rank = rank; m->world = world;
  SMX_CHECK(smx::ensure_comm_stream(m));
  if (!m->comm) SMX_CHECK(smx::ensure_sync_buf(m));
  drop_graphs(m);
  return SMX_OK;
}

// non-zero after a wait on a peer timed out (1) or a peer reported that its wait did (2): the results of that step are garbage on
// every rank.  Reads AND clears the word (smx_train_step* / smx_eval_step with a metrics read-back report it as SMX_ERR_COMM without clearing)
int smx_comm_p2p_error(smx_model* m, int32_t* error) {
  SMX_REQUIRE(m && error, "null argument");
  *error = 0;
  if (!m->p2p || !m->p2p->error) return SMX_OK;
  SMX_HIP(hipStreamSynchronize(m->st));
  unsigned e = 0;
  SMX_HIP(hipMemcpy(&e, m->p2p->error, sizeof(e), hipMemcpyDeviceToHost));
  if (e) SMX_HIP(hipMemset(m->p2p->error, 0, sizeof(e)));
  *error = (int32_t)e;
  return SMX_OK;
}

}  // extern "C"
