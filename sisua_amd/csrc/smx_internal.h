// smx_internal.h -- host-side declarations shared by the translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "smx_device.h"
#include "../../include/sisua_hip.h"

namespace smx {

void set_error(const std::string& msg);
#define SMX_HIP(call)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      smx::set_error(std::string(#call) + " failed: " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                     std::to_string(__LINE__) + ")");                                          \
      return SMX_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// --------------------------------------------------------------------------
// GEMM (smx_gemm.hip): C[M,N] (+ slabs) = op(A)[M,K] * op(B)[K,N], fp32 MFMA.
// --------------------------------------------------------------------------
enum GemmTile { TILE_AUTO = 0, TILE_128x32 = 1, TILE_64x64 = 2, TILE_32x128 = 3, TILE_32x32_K4 = 4, TILE_64x32_K2 = 5 };

// Optional transform applied to A while it is staged: row gather from the
// resident cells x genes matrix, log1p, input dropout (Philox or injected).
struct AXform {
  const int32_t* rows = nullptr;  // batch index -> row of A (nullptr: identity)
  int log1p = 0;
  float drop_p = 0.f, drop_scale = 1.f;  // input dropout (0 disables)
  NoiseKey nk{0, 0, 0, 0, nullptr};
  uint32_t cell_base = 0;
  const float* inj_mask = nullptr;  // injected multipliers [batch][ld]
  int inj_ld = 0;
  int u16 = 0;                      // A is the compact resident store: uint16 counts, same row pitch in elements
};

// Epilogue of the d z = dpre * W_dec^T product: the latent head's backward is elementwise in (cell, dim),
// so it runs on the accumulator tile and writes d lat directly (d z itself is never stored).
struct EpiLatentBwd {
  const float* lat = nullptr; int ld = 0;            // [B][ld]: mu | s_raw
  const float* sig = nullptr; const float* eps = nullptr;  // [B][Dp]
  float kl_scale = 0.f; int D = 0, Dp = 0, stochastic = 1, relu = 0;
  float* dlat = nullptr;                             // [B][ld]
  // SCALE (Monte-Carlo KL against the mixture prior): d(-log p)/dz [B][Dp] from scale_prior_fwd; nullptr: analytic KL
  const float* dklz = nullptr;
  // FactorVAE: gradient of the total-correlation term with respect to z [B][Dp], added to the product's d z
  const float* dz_add = nullptr;
};

struct AdamArgs;
struct GemmArgs {
  const float* A = nullptr; int lda = 0; int a_kmajor = 0;  // a_kmajor: A stored [K][M]
  const float* B = nullptr; int ldb = 0; int b_nmajor = 0;  // b_nmajor: B stored [N][K]
  float* C = nullptr; int ldc = 0; long slab_stride = 0;
  int M = 0, N = 0, K = 0;
  int split_k = 1;                 // gridDim.z; slab z written at C + z*slab_stride
  int k_chunk = 0;                 // K range per slice (set by the launcher)
  const float* bias = nullptr;     // [N], only with split_k == 1
  float* colsum = nullptr;         // [N] column sums of op(B) over K (bias gradient); split_k == 1, B k-major
  int use_xform = 0;
  AXform xf;
  int tile = TILE_AUTO;
  // weight-gradient products: every wave adds the sum of squares of its part of the tile to
  // sq_part[(by * sq_gx + bx) * 4 + wave] (per-tensor clipnorm without a separate pass over the gradient);
  // the launcher fills sq_gx and *sq_count (= slots written)
  float* sq_part = nullptr; int sq_gx = 0; int* sq_count = nullptr;
  int wide_store = 0;   // set by the launcher: 16-byte stores through an LDS transpose (large outputs)
  int c_colmajor = 0;   // split-K slabs stored [slice][N][128 rows] (32 x 32-K4 tile, M <= 128; rows beyond M zero): summed by a BatchNorm launch (BnFwdArgs::wide)
  int epi = 0;                     // 0: store C; 2: latent-head backward (EpiLatentBwd), split_k == 1
  EpiLatentBwd lb;
  // activation in the store path (layers without BatchNorm and dropout; split_k == 1): act = 1: C = act(acc + bias) with
  // act(v) = max(v, 0) + leak min(v, 0); act = 2 (backward): C = acc * act'(y) with y = act_out[row][col] the layer's
  // forward output (1 where y > 0, leak elsewhere) -- the bias / activation launches of such layers disappear
  int act = 0; float leak = 0.f; const float* act_out = nullptr; int act_ld = 0;
  int act_wrap = 0;     // act = 2, > 0: output rows >= act_wrap take the forward output of row - act_wrap (two backward sweeps stacked as rows of one product)
  int panel_hint = 0;   // launch_wgrad_group: take the panel form (smx_panel.h) for this problem whatever its M (N <= 128)
  // the latent head's backward product (epi = 2, its own kernel) can carry optimiser chunks [ride_first, ride_first + ride_count) of
  // *ride_adam (a HOST pointer, copied into the launch) as extra workgroups: the launch has four workgroups of its own
  const AdamArgs* ride_adam = nullptr; int ride_first = 0, ride_count = 0;
};
// Returns 0 or a negative smx_status.  N, lda, ldb, ldc multiples of 4; N multiple of 32.
// eff_split (optional) receives the number of slabs actually written.
// Developer knobs (tile / split-K / share sweeps, A/B of launch forms, test hooks): ONE registry instead of an environment variable per
// knob.  smx_set_tuning(name, value) sets one; the environment variable SMX_TUNING="name=value,name=value" presets them for a process
// (the scripts under tools/).  A knob that was never set reads its default.  docs/LAB_NOTES.md lists them; INTEGRATION.md lists the few switches a
// USER needs (those stay environment variables).
double tuning(const char* name, double dflt);
unsigned long long tuning_epoch();   // moves on with every smx_set_tuning / smx_clear_tuning
inline bool tuning_on(const char* name) { return tuning(name, 0.0) != 0.0; }
int launch_gemm(hipStream_t st, const GemmArgs& g, int* eff_split = nullptr);
// Several independent products in ONE launch (tiles 128x32 / 32x32-K4 only); eff_splits[i] receives
// the slab count of problem i.
#define SMX_GROUP_MAX 8
int launch_gemm_group(hipStream_t st, const GemmArgs* list, int n, int* eff_splits = nullptr);
// Two products that share the A operand (same M, K, layout, gather transform) and the split factor, side by side along N
// in ONE launch of the 32x32-K4 kernel (scvi: first layers of both encoders; the two / three output heads).
int launch_gemm_dual(hipStream_t st, const GemmArgs& g1, const GemmArgs& g2, int* eff_split = nullptr);
// Direct-operand bf16 x 3 form for mid-size products with a deep contraction axis (smx_dgemm.hip): A [M][K], B either layout,
// bias / activation store paths as launch_gemm; split_k must be 1
bool dgemm_supported(const GemmArgs& g);
int launch_dgemm(hipStream_t st, const GemmArgs& g);
// Heuristic split-K factor used by the model for K-heavy products.
int suggest_split_k(int M, int N, int K);

// --------------------------------------------------------------------------
// Fused kernels (smx_kernels.hip)
// --------------------------------------------------------------------------
struct LossArgs {
  int likelihood = 0;   // smx_likelihood
  int direct = 0;       // planes already activated (scvi)
  int backward = 1;
  const float* X = nullptr; int ldx = 0;  // counts [rows][ldx]
  int x_u16 = 0;                          // X is uint16 (compact resident store)
  const int32_t* rows = nullptr;          // gather (nullptr: identity)
  const float* P = nullptr;               // planes: plane c of cell b at P + b*ldp + c*plane_stride
  long ldp = 0; long plane_stride = 0;
  float* dP = nullptr;                    // same layout as P
  float* llk_part = nullptr;              // [B][n_chunks] partial sums (without the -lgamma(x+1) constant)
  int B = 0, G = 0, Gp = 0;
  float grad_scale = 1.f;                 // d loss / d llk (= -1/B_global); 1 for the test entry
};
inline int llk_planes(int likelihood) { return likelihood == SMX_LLK_MSE ? 1 : (likelihood == SMX_LLK_ZINB || likelihood == SMX_LLK_ZINBD) ? 3 : 2; }
int loss_chunks(int Gp, int B);   // partial sums per cell written by a launch over B cells
int loss_chunks_max(int Gp);      // upper bound over every batch size (allocation)
int launch_count_loss(hipStream_t st, const LossArgs& a);

// Noise for a LATER launch, generated by extra workgroups of the first BatchNorm launch (they run on idle CUs beside
// the 32 BN workgroups): the decoder's front launch computes the whole latent tile in every workgroup and reads its
// eps / dropout multipliers instead of evaluating Philox redundantly.
struct NoiseJob {
  float* dst = nullptr; int ld = 0; int width = 0;   // [B][ld], columns >= width left untouched
  int normal = 0;                                    // 0: dropout multipliers, 1: standard normals
  float p = 0.f; uint32_t stream = 0;                // stream | sample << 8
};
#define SMX_NOISE_JOBS 6
#define SMX_NOISE_BLOCKS_PER_JOB 8

struct LatentArgs {
  int stochastic = 1, relu = 0, training = 1;
  const float* lat = nullptr; int ld = 0;   // [B][2*Dp] (mu | s_raw) or [B][Dp]
  int B = 0, D = 0, Dp = 0;
  NoiseKey nk{0, 0, 0, 0, nullptr};
  const int32_t* rows = nullptr; uint32_t cell_base = 0;
  const float* inj_eps = nullptr; int inj_ld = 0;
  float* z = nullptr; float* sig = nullptr; float* eps = nullptr;  // [B][Dp]
  float* kl = nullptr;                                             // [B]
  // backward
  const float* dz = nullptr; int dz_slabs = 1; long dz_slab_stride = 0;
  float kl_scale = 0.f;       // beta / B_global
  float* dlat = nullptr;      // [B][2*Dp] or [B][Dp]
};

// SCALE read literally (scale.py:26,38-47; SMX_MODEL_SCALE_POST): q(z|x) = sum_c pi_c N(mu_c, diag sigma_c^2) from a latent head of
// 1 + 2 C planes of width Dp (logits in the first C columns of plane 0 | mu_1 .. mu_C | raw sigma_1 .. sigma_C), C <= 8, D <= 64.
enum { ST_MIX_PICK = 67 };   // Philox stream of the uniform that picks a cell's component
struct MixLatArgs {
  const float* lat = nullptr; int ld = 0; int B = 0, D = 0, Dp = 0, C = 0;
  NoiseKey nk{0, 0, 0, 0, nullptr}; NoiseKey nk_pick{0, 0, 0, 0, nullptr};
  const int32_t* rows = nullptr; uint32_t cell_base = 0;
  const float* inj_eps = nullptr; int inj_ld = 0;
  float* z = nullptr; float* eps = nullptr;     // [B][Dp] the draw and its noise
  float* zmean = nullptr; float* zstd = nullptr;   // [B][Dp] the mixture's mean / standard deviation (what predict and encode report)
  float* kl = nullptr;        // [B] log q(z|x) - log N(z; 0, I)
  float* resp = nullptr;      // [B][32] responsibilities of the components under q at z
  int32_t* pick = nullptr;    // [B] the picked component
  // backward
  const float* dz = nullptr; int dz_slabs = 1; long dz_slab_stride = 0; int ldz = 0;
  float kl_scale = 0.f;       // beta / B_global
  float* dlat = nullptr;      // [B][ld]
};
int launch_mixlat_fwd(hipStream_t st, const MixLatArgs& a);
int launch_mixlat_bwd(hipStream_t st, const MixLatArgs& a);

struct BnFwdArgs {
  const float* pre = nullptr; int n_slabs = 1; long slab_stride = 0; int ld = 0;  // pre-activation slabs [S][B][ld]
  int B = 0, H = 0, Hp = 0;
  int batchnorm = 1, training = 1;
  const float* gamma = nullptr; const float* beta = nullptr; const float* bias = nullptr;
  float* moving_mean = nullptr; float* moving_var = nullptr;   // updated in place when update_moving
  float* batch_mean = nullptr; float* batch_var = nullptr;     // [Hp] outputs (training)
  int update_moving = 1;
  float momentum = 0.99f, eps = 1e-3f;
  float* xhat = nullptr;     // [B][Hp] normalised pre-activation (saved for backward)
  float* inv_std = nullptr;  // [Hp]
  float* out = nullptr;      // [B][Hp] relu + dropout
  float drop_p = 0.f;
  NoiseKey nk{0, 0, 0, 0, nullptr};
  const int32_t* rows = nullptr; uint32_t cell_base = 0;  // cell id = cell_base + rows[b]
  const float* inj_mask = nullptr; int inj_ld = 0;
  int n_jobs = 0; NoiseJob jobs[SMX_NOISE_JOBS];
  // front != 0 (first decoder layer): the layer's INPUT is produced by this launch too -- the latent sample
  // z = mu + sigma eps with its KL (`lat`; every workgroup computes the whole [B][Dp] tile into LDS, workgroup 0
  // also stores z / sigma / eps / KL for the backward pass) and the layer's product pre = z W as dot products
  // (K = Dp <= 64: 64 fmas per output row and thread), so the latent kernel and the product kernel disappear
  int front = 0; LatentArgs lat; const float* W = nullptr; int ldw = 0;
  float leak = 0.f;   // activation max(y, 0) + leak min(y, 0): 0 = ReLU; the FactorVAE discriminator's leaky ReLU uses 0.2
  // wide != 0: `pre` holds HUNDREDS of slabs (one per workgroup of a wide-panel product: smx_bigk.hip), each COLUMN-major [Hp][128 rows]; the
  // launch sums them itself, one workgroup per column, in the order of bigk_reduce_kernel (bn_wide_fwd_kernel: no reduce launch)
  int wide = 0;
};
bool bn_wide_supported(int B, int Hp, int n_slabs);
int launch_bn_act_fwd(hipStream_t st, const BnFwdArgs& a);
bool bn_front_supported(int B, int Dp);

// Per-step device-resident scalars.
struct StepState {
  uint32_t step;      // index of the step in flight (Philox counter word 2)
  uint32_t next;      // master copy only: optimiser steps completed so far
  float lr_t;         // bias-corrected Adam step size of the step in flight
  uint32_t cursor;    // position (in steps) of the step in flight inside the uploaded row-id order
};

// ELBO scalars of a step; computed by one workgroup that rides along with another launch
struct MetricsArgs {
  const float* llk_part = nullptr; int n_chunks = 0;   // [B][n_chunks]
  const float* lgx1 = nullptr; const int32_t* rows = nullptr;  // per-cell sum lgamma(x+1), gathered
  const float* llk_y = nullptr;   // [B] masked label llk or nullptr
  const float* llk_o = nullptr;   // [B] log-likelihood of the observed extra outputs (weight 1) or nullptr -> out[7]
  const float* kl = nullptr; const float* kl_l = nullptr;
  int B = 0; float alpha = 0.f, beta = 1.f; float inv_global_batch = 0.f;
  float* out = nullptr;           // [8]: loss, nllk_x, nllk_y, kl, kl_l, tc, dtc_loss, nllk_o
  // per-step history of a train_steps call (single GPU: written here; data parallel: by the optimiser launch,
  // after the all-reduce): hist[cursor * 8 + i] = out[i]
  float* hist = nullptr; const StepState* state = nullptr;
  // FactorVAE: tc [B] = d(z_b); dl [2B] = the discriminator's per-row loss; out[5] = mean tc, out[6] = discriminator
  // objective (mean dl pairs - alpha mean llk_y), out[0] += gamma mean tc
  const float* tc = nullptr; const float* dl = nullptr; float gamma = 0.f;
};
int launch_metrics(hipStream_t st, const MetricsArgs& a);

// Optimiser over the flat parameter buffer.
#define SMX_MAX_TENSORS 48
#define SMX_SQR_MAX 16             // reduce riders per BatchNorm-backward launch
#define SMX_SQR_PER_TENSOR 8       // ... and per tensor
#define SMX_SQR_MIN_SLOTS 4096     // tensors with fewer sum-of-squares slots are summed by the optimiser's workgroups themselves
#define SMX_SQ_SMALL_TENSOR 131072  // floats: below this a workgroup re-derives the tensor's norm by itself (the output bias of three planes up to 43 000 genes)
struct OptChunk { int32_t tensor; int32_t offset; int32_t count; int32_t first_chunk; int32_t n_chunks; int32_t tensor_count; int32_t pad[2]; };
struct AdamArgs {
  // ELBO scalars ride along as one extra workgroup of the gradient-norm kernel
  MetricsArgs metrics; int with_metrics = 0;
  // the optimiser's workgroup 0 finishes the step: master counter, and the state + row ids of the next step
  StepState* master = nullptr; StepState* next_state = nullptr; const int32_t* order = nullptr; int32_t* next_rows = nullptr;
  int batch = 0, prepare_next = 0; float lr = 1e-3f;
  float* params = nullptr; float* grads = nullptr; float* m = nullptr; float* v = nullptr;
  const OptChunk* chunks = nullptr; int n_chunks = 0;
  int n_launch = 0;             // workgroups of the optimiser launch: every chunk but [gap_from, gap_from + gap_len) (those rode along earlier)
  int gap_from = 0, gap_len = 0;
  float* partial = nullptr;     // [n_chunks] sum of squares per chunk
  int sq_chunks = -1;           // chunks the norm pass of launch_adam covers (-1: all; the chained data-parallel form: the front chunks -- the heads' went on the communication stream)
  // norms without the separate pass (use_sq): per tensor either the slots the weight-gradient products wrote
  // (sq_count > 0) or, for small tensors, a sweep of the tensor's gradient by every workgroup that needs it
  int use_sq = 0; const float* sq_slots = nullptr; int sq_first[SMX_MAX_TENSORS]; int sq_count[SMX_MAX_TENSORS];
  float* tensor_norm = nullptr; // [n_tensors] written by the update kernel (pre-clip norms)
  // flag opt_shard: this rank's slice [shard_lo, shard_hi) of the flat buffer (floats, multiples of 64); the sharded kernels touch nothing outside it
  long shard_lo = 0, shard_hi = 0;
  const StepState* state = nullptr;
  float b1 = 0.9f, b2 = 0.999f, eps = 1e-7f, clipnorm = 100.f;
  float grad_scale = 1.f;       // extra factor on the gradient (1: the loss is already scaled by 1 / global batch)
  // SCALE's tied tensors (scale.py:29-33: ONE shared location / scale vector stored as C identical rows, every row holding the shared
  // variable's gradient): the clip norm is the shared variable's, i.e. the stored tensor's sum of squares / C
  int tied_t0 = -1, tied_t1 = -1; float tied_inv = 1.f;
  float* hist_dp = nullptr; const float* tail_metrics = nullptr;   // data parallel: the reduced scalars go to the history here
  // data parallel, world > 1: the moving BatchNorm statistics take the all-reduced batch statistics (mean over the ranks)
  // in extra workgroups of the gradient-norm launch (it was a launch of its own)
  float* bn_moving = nullptr; const float* bn_batch = nullptr; int bn_total = 0; float bn_inv_world = 1.f, bn_momentum = 0.99f;
};
int launch_adam(hipStream_t st, const AdamArgs& a);
int launch_adam_sweep(hipStream_t st, const AdamArgs& a, int first, int count, int wgs);   // chunks [first, first + count) by `wgs` persistent workgroups
int launch_grad_sqsum_range(hipStream_t st, const AdamArgs& a, int first, int count);   // a.partial[chunk] = the chunk's sum of squares, chunks [first, first + count)

struct BnBwdArgs {
  const float* dout = nullptr; int n_slabs = 1; long slab_stride = 0; int ld = 0;  // d loss / d out slabs
  const float* out = nullptr;    // forward output (relu/dropout mask: out > 0)
  const float* xhat = nullptr; const float* inv_std = nullptr; const float* gamma = nullptr;
  int B = 0, H = 0, Hp = 0;
  int batchnorm = 1, training = 1;
  float drop_scale = 1.f;        // 1/(1-p) (1 in eval)
  float* dpre = nullptr;         // [B][Hp]
  float* dgamma = nullptr; float* dbeta = nullptr; float* dbias = nullptr;
  // data parallel: the ELBO scalars must be in the flat buffer before the all-reduce -- one extra workgroup here
  MetricsArgs metrics; int with_metrics = 0;
  // single GPU: the output / label heads' gradients are final before this launch, and this launch leaves most CUs
  // idle -- adam_count extra workgroups apply the optimiser to chunks [adam_first, adam_first + adam_count)
  AdamArgs adam; int adam_first = 0, adam_count = 0;
  // ... or, when that update is too large to ride along, sqr_count workgroups that only sum sum-of-squares slots:
  // rider i leaves the sum of sq_slots[sqr_first[i] .. + sqr_n[i]) in sq_total[sqr_dst[i]] (up to SMX_SQR_PER_TENSOR
  // riders per tensor; the optimiser's workgroups then read that many numbers instead of every slot)
  int sqr_count = 0; int sqr_first[SMX_SQR_MAX], sqr_n[SMX_SQR_MAX], sqr_dst[SMX_SQR_MAX]; float* sq_total = nullptr;
  // front != 0 (last encoder layer): the incoming gradient d h = d lat W_lat^T is produced here as dot products
  // (K = width of the latent head <= 64; d lat [B][fK] staged in LDS, this thread's row of W_lat in registers)
  // instead of being read from a slab another launch wrote
  int front = 0; const float* fD = nullptr; int fld = 0; const float* fW = nullptr; int fldw = 0; int fK = 0;
  // fold_dz (round 6; with front, fK = 64 = 2 Dp, at most 128 cells, a first decoder layer of 128 units): the launch computes fD ITSELF --
  // d z = zD zW^T (zD [B][128]: d pre-activation of the first decoder layer, zW [Dp = 32][128]: its weights) as bf16 x 3 MFMAs and the
  // latent head's backward (zlb: what gemm_latent_bwd_kernel's epilogue did) on it, in every workgroup; workgroup 0 also stores d lat for
  // the weight-gradient launch.  One launch of the chain less (smx_step.hip: backward_pass).
  int fold_dz = 0; const float* zD = nullptr; int zld = 0; const float* zW = nullptr; int zldw = 0; EpiLatentBwd zlb;
  int diag = 0;   // SMX_BN_DIAG bits 16 / 32 / 64: skip the front's dot products / tile load / W row load (timing only)
  float leak = 0.f;   // slope of the activation for out <= 0 (layers without dropout only)
  int wide = 0;       // as BnFwdArgs::wide: `dout` = column-major slabs [n_slabs][Hp][128], summed here (bn_wide_bwd_kernel)
};
bool bn_bwd_front_supported(int B, int K);
bool bn_bwd_fold_supported(int B, int fK, int Dp);   // BnBwdArgs::fold_dz
int launch_bn_act_bwd(hipStream_t st, const BnBwdArgs& a);
// two layers over the same minibatch in ONE launch (scvi: encoder + library encoder)
bool bn_dual_supported(int B);
int launch_bn_act_fwd_dual(hipStream_t st, const BnFwdArgs& a, const BnFwdArgs& b);
int launch_bn_act_bwd_dual(hipStream_t st, const BnBwdArgs& a, const BnBwdArgs& b);   // both with the gradient front

// SyncBatchNorm (smx_kernels.hip): phase 0 leaves this rank's column statistics in `gather` [world][2][Hp],
// the caller all-reduces it, phase 1 finishes the pass with the global statistics.
struct BnSyncArgs { float* gather = nullptr; int rank = 0, world = 1; };
int launch_bn_sync_fwd(hipStream_t st, const BnFwdArgs& a, const BnSyncArgs& y, int phase);
int launch_bn_sync_bwd(hipStream_t st, const BnBwdArgs& a, const BnSyncArgs& y, int phase);

int launch_latent_fwd(hipStream_t st, const LatentArgs& a);
int launch_latent_bwd(hipStream_t st, const LatentArgs& a);

// SCALE (sisua/models/scale.py:13-49): trainable Gaussian-mixture prior over z, one-sample Monte-Carlo KL
struct ScalePriorArgs {
  const float* z = nullptr; const float* sig = nullptr; const float* eps = nullptr; int B = 0, D = 0, Dp = 0, C = 0;
  const float* logits = nullptr; const float* loc = nullptr; const float* scale_raw = nullptr;   // [C], [Cp][Dp], [Cp][Dp]
  float* kl = nullptr;        // [B]  log q(z|x) - log p(z)
  float* resp = nullptr;      // [B][32] responsibilities
  float* dklz = nullptr;      // [B][Dp] d(-log p)/dz
  float kl_scale = 0.f;       // beta / B_global
  float* g_logits = nullptr; float* g_loc = nullptr; float* g_scale = nullptr;   // gradients (backward)
  int tie_mixtures = 0, tie_loc = 0, tie_scale = 0;   // scale.py:29-33: a tied tensor's rows all receive the sum of the rows' gradients (logits: none)
  float* tril_part = nullptr; size_t tril_part_floats = 0;   // covariance = 'tril', backward: partial sums [C][ceil(B / 16)][D][D + 2]
  int tril = 0;   // covariance = 'tril': scale_raw / g_scale are [C D][Dp], row c D + p = row p of component c's lower-triangular factor (D <= 32)
};
int launch_scale_prior_fwd(hipStream_t st, const ScalePriorArgs& a);
int launch_scale_prior_bwd(hipStream_t st, const ScalePriorArgs& a);

// scvi library latent: latl [B][ldl] (mu_l, s_raw_l) -> l sample, KL vs N(lib_mean, sqrt(lib_var))
struct LibLatentArgs {
  const float* latl = nullptr; int ld = 0; int B = 0;
  const float* library = nullptr;  // [n_cells][2] resident, gathered by rows; or [B][2] when rows == nullptr
  const int32_t* rows = nullptr; uint32_t cell_base = 0;
  NoiseKey nk{0, 0, 0, 0, nullptr};
  const float* inj_eps = nullptr; int inj_ld = 1;
  float clip_library = 1e3f;
  float* l = nullptr; float* sig = nullptr; float* eps = nullptr; float* kl = nullptr;  // [B]
  const float* dl = nullptr;  // [B] d loss / d l (already masked by the clip)
  float kl_scale = 0.f;
  float* dlatl = nullptr;     // [B][ld]
};
int launch_lib_latent_fwd(hipStream_t st, const LibLatentArgs& a);
int launch_lib_latent_bwd(hipStream_t st, const LibLatentArgs& a);

// scvi head: raw [B][3][Gp] -> planes (rate, theta, gate) and back.
struct ScviHeadArgs {
  const float* raw = nullptr; float* planes = nullptr; long ld = 0; long plane_stride = 0;
  int B = 0, G = 0, Gp = 0, k = 2;
  const float* l = nullptr; float clip_library = 1e3f;
  float* rho_raw = nullptr;          // [B][Gp] saved softmax
  const float* dplanes = nullptr; float* draw = nullptr; float* dl = nullptr;
};
int launch_scvi_head_fwd(hipStream_t st, const ScviHeadArgs& a);
int launch_scvi_head_bwd(hipStream_t st, const ScviHeadArgs& a);

// scvi head of a training step as ONE row-local launch (smx_scvi.hip): library latent (its head as dot products,
// sample, KL), softmax-rate head, NBD / ZINBD likelihood + gradient, backward through the head and the library latent
struct ScviTrainArgs {
  const float* raw = nullptr; long ld = 0; long plane_stride = 0;   // raw head outputs [B][k][Gp] (bias added)
  int B = 0, G = 0, Gp = 0; int likelihood = 0;
  const float* X = nullptr; int ldx = 0; int x_u16 = 0; const int32_t* rows = nullptr;
  int x_identity = 0;          // X holds the minibatch's rows already (row b = cell b): `rows` then serves the library / noise keys only
  float clip_library = 1e3f; float grad_scale = 1.f;
  float* draw = nullptr;       // [B][ld] d loss / d raw
  float* llk_part = nullptr;   // [B] one partial per cell (without the -lgamma(x+1) constant)
  const float* hl = nullptr; int ldh = 0; int Kl = 0;                    // library encoder output [B][ldh], Kl columns
  const float* Wl = nullptr; int ldwl = 0; const float* bl = nullptr;    // library latent head [Kl][ldwl] (columns mu, s), bias
  const float* library = nullptr; uint32_t cell_base = 0;
  NoiseKey nk{0, 0, 0, 0, nullptr};
  const float* inj_eps = nullptr; int inj_ld = 1;
  float kl_scale = 0.f;
  float* latl = nullptr; int ldl = 0;                                    // [B][ldl]: (mu_l, s_l) kept for inspection
  float* l = nullptr; float* sig = nullptr; float* eps = nullptr; float* kl = nullptr;   // [B]
  float* dlatl = nullptr;      // [B][ldl] gradient wrt (mu_l, s_l), zeros beyond
  float* dl = nullptr;         // [B]
};
bool scvi_head_train_supported(const ScviTrainArgs& a);
int launch_scvi_head_train(hipStream_t st, const ScviTrainArgs& a);
// a plane of scvi's gene output without a Dense head (dispersion / inflation = 'share', scvi.py:66-86): its per-gene vector copied into
// every row of the raw plane / the column sum of the plane's d raw (row order)
int launch_plane_fill(hipStream_t st, float* dst, long ld, const float* v, int B, int Np, int single = 0);   // single: v is ONE scalar
int launch_plane_colsum(hipStream_t st, const float* src, long ld, float* dst, int B, int Np, int single_G = 0);   // single_G > 0: the sum over the G live genes too -> dst[0]

struct LabelArgs {
  int kind = 0;                  // smx_label_likelihood
  int C = 1;                     // mixture components (SMX_LABEL_MIXNB)
  const float* raw = nullptr; int ld = 0;          // [B][ky*Pp]
  const float* Y = nullptr; int ldy = 0;           // labels [rows][ldy]
  const int32_t* rows = nullptr;
  const uint8_t* mask = nullptr;                   // resident [n_cells] or nullptr (all unlabeled)
  int observed = 0;                                // an observed OUTPUT variable (outputs[1:]): every cell counts, the mask is not consulted
  int B = 0, P = 0, Pp = 0;
  float grad_scale = 0.f;        // -alpha / B_global
  float* draw = nullptr;         // [B][ky*Pp]
  float* llk = nullptr;          // [B]  mask * llk_y  (accumulated across label heads: add != 0)
  int add = 0;
  int backward = 1;
};
int launch_label_loss(hipStream_t st, const LabelArgs& a);

// Prepares the per-step state `dst` (+ row ids) of the step at order position `cursor` from the master
// counter.  Eager mode runs it once per train_steps call (later steps are prepared by the optimiser
// kernel of the step before); graph mode runs it as the first node of every step.
int launch_step_begin(hipStream_t st, StepState* master, StepState* dst, const int32_t* order, int32_t* rows,
                      int batch, int cursor_from_master, uint32_t cursor, float lr, float b1, float b2);



// Test helper: Philox multipliers / normals exactly as kernels draw them.
int launch_noise_probe(hipStream_t st, NoiseKey nk, const int64_t* cell_ids, int B, int width, float p, float* mult,
                       float* normal);

// ---- output product fused with the count likelihood, wide (smx_headloss.hip) -------------------
struct HeadLossArgs {
  const float* H = nullptr; int ldh = 0;          // decoder output [B][ldh] (after BN / ReLU / dropout)
  const float* W = nullptr; int ldw = 0;          // [Hp][k * Gp]
  const float* bias = nullptr;                    // [k * Gp]
  const float* X = nullptr; int ldx = 0; const int32_t* rows = nullptr; int x_u16 = 0;
  float* dP = nullptr; long ldp = 0; long plane_stride = 0;   // d loss / d P (product_only: P itself)
  float* llk_part = nullptr;                      // [B][Gp / 32]
  int B = 0, G = 0, Gp = 0, Hp = 0, likelihood = 0;
  float grad_scale = 1.f;
  int product_only = 0;                           // timing variant: the product alone, P stored
  // scoring (smx_marginal_llk, stacked draws): likelihood partials only, nothing else stored; row r of H is draw
  // r / row_mod of cell r % row_mod, whose counts are row (r % row_mod) of X / rows; H is then K-MAJOR ([Hp][ldh], row r
  // of the stacked batch in column r)
  int llk_only = 0, row_mod = 0;
  int n_ct = 0, n_gt = 0;                         // set by the launcher
  int bf16x3 = 0;                                 // the product from bf16 MFMAs on three-way split operands (smx_device.h)
};
// cells x padded genes x planes from which the output head's products run as bf16 x 3 by default (flag "bf16x3" / SMX_BF16X3):
// every size -- measured A/B on one box, us per step f32 -> bf16 x 3: 8kly 84.7 -> 83.2, 8kly-scvi 115.3 -> 114.3, eccly-sisua
// 125.9 -> 122.3, 8kly-2layer 125.9 -> 124.1, c5-shard 263 -> 222 (with the gene-axis products of smx_bigk.hip)
#define SMX_BF16X3_MIN_WORK 1L
bool use_bf16x3(long work);
bool head_loss_supported(int B, int Hp, int Gp);
int head_loss_chunks(int Gp);
int launch_out_head_loss(hipStream_t st, const HeadLossArgs& a);

// ---- importance-weighted log p(x) over stacked posterior draws (smx_score.hip) ------------------
#define SMX_SCORE_MAX_DRAWS 1024   // draws per stacked pass (chunks of them fold into the running log-sum-exp)
struct ScoreDrawArgs {
  const float* lat = nullptr; int ld = 0;   // [B][2 * Dp] (mu | s_raw)
  int B = 0, D = 0, Dp = 0, S = 0, s0 = 0;  // S draws starting at sample index s0
  NoiseKey nk{0, 0, 0, 0, nullptr};         // stream word without the sample index
  const int32_t* rows = nullptr; uint32_t cell_base = 0;
  float* z = nullptr;                       // [S * B][Dp]
  float* lw = nullptr;                      // [S * B]: log N(z; 0, I) - log q(z | x)  (+ the library latent's terms, scvi)
  // scvi: the library latent l = mu_l + sigma_l eps_l of every row as well (scvi.py:37-45, 88-106)
  const float* latl = nullptr; int ld_l = 0;            // [B][ld_l]: (mu_l, s_raw_l) in columns 0, 1
  const float* library = nullptr; const int32_t* lib_rows = nullptr;   // prior (mean, variance) per cell, indexed like lgx1
  NoiseKey nk_l{0, 0, 0, 0, nullptr};
  float* l = nullptr;                       // [S * B]
  // SCALE: the trainable Gaussian-mixture prior replaces N(0, I) in log w (scale.py:13-49); Dp <= 64
  const float* pr_logits = nullptr; const float* pr_loc = nullptr; const float* pr_scale_raw = nullptr; int C = 0;   // [C], [C][Dp], [C][Dp]
};
int launch_score_draws(hipStream_t st, const ScoreDrawArgs& a);
struct ScoreBnArgs {
  float* h = nullptr; long R = 0; int H = 0, Hp = 0;
  const float* gamma = nullptr; const float* beta = nullptr; const float* moving_mean = nullptr; const float* moving_var = nullptr;
  float eps = 1e-3f, leak = 0.f;
  float* out_t = nullptr; long ldt = 0;   // non-null: write the result transposed, out_t [Hp][ldt] (gamma may then be null: plain transpose)
  __bf16* out3 = nullptr;                 // non-null: write the result as its three-way bf16 split [3][R][Hp] (gamma may be null)
};
// a one-layer decoder with BatchNorm over the stacked rows in one launch (draws + product + BatchNorm + activation + split): score_decoder1_kernel
struct ScoreDec1Args {
  ScoreDrawArgs d;                                 // (z is not written; no mixture prior, no library latent)
  const float* W = nullptr; int ldw = 0;           // [Dp][Hp]
  int H = 0, Hp = 0;
  const float* gamma = nullptr; const float* beta = nullptr; const float* moving_mean = nullptr; const float* moving_var = nullptr;
  float eps = 1e-3f, leak = 0.f;
  __bf16* out3 = nullptr;                          // [3][S * B][Hp]
};
bool score_decoder1_supported(int Dp, int Hp);
int launch_score_decoder1(hipStream_t st, const ScoreDec1Args& a);
struct ScoreSplitWArgs {
  const float* W = nullptr; int ldw = 0, Gp = 0;   // [Hp][k * Gp]
  int n_gt = 0, nslab = 0, NP = 0;                 // gene tiles of 32, slabs of 32 k, parameter planes
  __bf16* img = nullptr;                           // [n_gt][nslab][3][NP][2][2][32][8]
};
int launch_score_split_w(hipStream_t st, const ScoreSplitWArgs& a);
int launch_score_bn_act(hipStream_t st, const ScoreBnArgs& a);
struct ScoreHeadArgs {
  const __bf16* A3 = nullptr;                     // last decoder output as its three-way bf16 split [3][R][Hp]
  const __bf16* Wimg = nullptr;                   // W as slab images (score_split_w_kernel)
  const float* bias = nullptr;                    // [k * Gp]
  const float* X = nullptr; int ldx = 0; const int32_t* rows = nullptr; int x_u16 = 0;   // counts of cell (row % row_mod)
  float* llk_part = nullptr;                      // [R][Gp / 32]
  int R = 0, row_mod = 0, G = 0, Gp = 0, Hp = 0, likelihood = 0;
  int n_rb = 0, n_gt = 0, rb_group = 0, gt_per_xcd = 0;   // set by the launcher
  int no_queue = 0;                               // set by the launcher (knob no_score_queue): lgamma differences straight-line instead of through the non-zero queue
  int n_split = 0, wg_per_xcd = 0;                // set by the launcher for score_walk_kernel: row ranges per gene tile, workgroups per XCD
};
bool score_head_supported(int Hp, int Gp);
int launch_score_head(hipStream_t st, const ScoreHeadArgs& a);
// scvi head of stacked rows: softmax-rate head + NBD / ZINBD log-likelihood of one row per workgroup, from the raw planes
struct ScviScoreArgs {
  const float* raw = nullptr; long ld = 0; long plane_stride = 0;   // [R][k * Gp]
  int R = 0, G = 0, Gp = 0, k = 2, likelihood = 0, row_mod = 0;
  const float* l = nullptr; float clip_library = 1e3f;             // [R]
  const float* X = nullptr; int ldx = 0; const int32_t* rows = nullptr; int x_u16 = 0;
  float* llk = nullptr;                                             // [R]
};
bool scvi_score_supported(int Gp);
int launch_scvi_score_rows(hipStream_t st, const ScviScoreArgs& a);
struct IwStackArgs {
  const float* llk_part = nullptr; int n_chunks = 0;   // [S * B][n_chunks]
  const float* lw = nullptr;                            // [S * B] latent part of log w (nullptr: plain log-mean-exp of the likelihoods)
  const float* lgx1 = nullptr; const int32_t* rows = nullptr;
  float* run_max = nullptr; float* run_sum = nullptr; float* llk_sum = nullptr;   // [B] (llk_sum may be null)
  int B = 0, S = 0, first = 1;
};
int launch_iw_stack(hipStream_t st, const IwStackArgs& a);

// ---- both backward products of the output head in one launch (smx_headbwd.hip) ------------------
struct HeadBwdArgs {
  const float* D = nullptr; int ldd = 0;           // decoder output [B][ldd = Hp]
  const float* dP = nullptr; long ldp = 0;         // [B][k * Gp]
  const float* W = nullptr; int ldw = 0;           // [Hp][k * Gp]
  float* dW = nullptr; float* db = nullptr;        // gradients, laid out as W / bias
  float* slab = nullptr; long slab_stride = 0;     // dd slabs [n_slices][B][Hp]
  int dd_colmajor = 0;                             // ... stored [n_slices][Hp][128 cells] instead (B <= 128, no label slabs): BnBwdArgs::wide
  int B = 0, Hp = 0, Gp = 0, n_planes = 0;
  int n_slices = 0, k_chunk = 0;                   // from head_bwd_slices
  float* sq_part = nullptr; int* sq_count = nullptr;
  // sep != 0 (scvi): every plane is a tensor of its own -- W / dW [Hp][ldw = Gp], bias gradient [Gp], sum-of-squares slots
  int sep = 0; const float* Wp[3] = {nullptr, nullptr, nullptr}; float* dWp[3] = {nullptr, nullptr, nullptr};
  float* dbp[3] = {nullptr, nullptr, nullptr}; float* sqp[3] = {nullptr, nullptr, nullptr}; int* sq_countp[3] = {nullptr, nullptr, nullptr};
  // label heads riding along (SISUA / MISA): n_extra more d d slabs, slab n_slices + e = xA[e] xW[e]^T with xA [B][xlda]
  // (the head's d Y), xW [Hp][xldw] (its weights), K = xK[e] (a multiple of 32) -- what a grouped launch of its own did
  int n_extra = 0; const float* xA[SMX_MAX_LABELS] = {nullptr, nullptr, nullptr, nullptr}; int xlda[SMX_MAX_LABELS] = {0, 0, 0, 0};
  const float* xW[SMX_MAX_LABELS] = {nullptr, nullptr, nullptr, nullptr}; int xldw[SMX_MAX_LABELS] = {0, 0, 0, 0}; int xK[SMX_MAX_LABELS] = {0, 0, 0, 0};
  int n_ht = 0, n_gt = 0, n_ct = 0, n_w = 0;       // set by the launcher
  int diag = 0;                                    // SMX_HEADBWD_DIAG bit 1 / 2: role-0 / role-1 workgroups return at once (timing only)
  int bf16x3 = 0;                                  // both products from bf16 MFMAs on three-way split operands (smx_device.h)
  int skip_dw = 0;                                 // role 1 (d d) only
  int skip_dd = 0;                                 // role 0 only: d d came from launch_bigk
};
bool head_bwd_supported(int B, int Hp, int Gp);
int head_bwd_slices(long ldp, int max_slabs, int* k_chunk);
int launch_out_head_bwd(hipStream_t st, const HeadBwdArgs& a);

// ---- products contracting over the gene axis of a wide panel: one workgroup per K slice + a reduce launch (smx_bigk.hip) ----
struct BigKArgs {
  const float* A = nullptr; long lda = 0;          // [M][K] rows, k contiguous (float32, or the uint16 store: a_u16)
  int a_u16 = 0, log1p = 0; const int32_t* rows = nullptr;   // gather by row id (nullptr: identity), log1p on the way
  const float* Bm = nullptr; long ldb = 0; int b_kmajor = 0; // b_kmajor: B stored [K][N] (n contiguous); else [N][K] (k contiguous)
  float* part = nullptr; long slab_stride = 0;     // [n_slices][M][ldc] partial slabs (scratch)
  float* out = nullptr; int ldc = 0;               // [M][ldc]: their sum in slice order
  int M = 0, N = 0, K = 0;
  int n_slices = 0, k_chunk = 0;                   // from bigk_slices
  int stages = 0;                                  // set by the launcher: 32-deep stages in flight + 1 (SMX_BIGK_STAGES)
  // colmajor: the slabs are stored [slice][N columns][128 rows] (M <= 128; rows beyond M zero) and NOT summed by this launch:
  // the consumer is a BatchNorm launch that sums them itself (BnFwdArgs::wide)
  int colmajor = 0;
};
// measured at 128 x 20 000 (tools/bigk_stages_ab.sh, same box): 4 / 3 / 2 stage buffers -> c5-shard 205.4 / 203.8 / 202.9 us per step, bit-identical
// results: the operands of these launches come from the last-level cache, one stage ahead covers their latency, and 64 KB of LDS
// instead of 128 KB lets the next workgroup in while one drains
#define SMX_BIGK_STAGES_DEFAULT 2
int bigk_slices(long K, int max_slices, int* k_chunk);
bool bigk_supported(const BigKArgs& a);
int launch_bigk(hipStream_t st, const BigKArgs& a);

int launch_grad_sqsum_shard(hipStream_t st, const AdamArgs& a, int first, int count);   // partial[chunk] = sum of squares of the chunk's part inside the slice
int launch_head_norms(hipStream_t st, const AdamArgs& a, int first, int count);        // tensor_norm of every tensor of the chunk range from partial[]
int launch_adam_shard(hipStream_t st, const AdamArgs& a, int first, int count, int wgs);   // clip + Adam of the chunk range's elements inside the slice
int launch_bigk_reduce(hipStream_t st, const float* part, long slab_stride, int n_slices, long n4, float* out);

// ---- the whole output head of a training step at a wide panel in one launch (smx_headfused.hip) ----
#define SMX_HEAD_FUSED_TAB_BYTES (8 * 12 * 64 * 16)
#define SMX_SHARD_SLACK 4096          // floats behind the flat buffers: world x round_up(bucket / world, 64) <= bucket + 64 world (world <= 64)
#define SMX_DP_BUCKETS_MIN_BYTES 3000000   // data parallel: two buckets from this many bytes of head gradients
#define SMX_HEAD_FUSED_MIN_GENES 4096
int head_fused_min_genes();   // (knob head_fused_min_genes; smx_headfused.hip)
#define SMX_HEAD_FUSED_MAX_CELLS 256   // (one launch per 128 cells)
// the heads' background optimiser sweep (smx_step.hip: head_sweep_start): from this many 4096-float chunks of head parameters, one persistent
// workgroup per so many chunks.  Measured at 128 x 20 000 (1880 chunks; c5-shard, us per step; 182.3 without), by workgroups: 48 -> 245,
// 64 -> 176-215 (the next output head waits for the sweep), 96 -> 178-182, 128 -> 175.3-176.3, 160 -> 177.5-181, 192 -> 176.8, 256 -> 182.2,
// 512 -> 185.7, 1024 -> 191.1 (the small dependent launches beside it slow down by more than it hides)
#define SMX_HEAD_SWEEP_MIN_CHUNKS 1536
// (end of round 5, the window between two output heads ~25 us shorter: 126 -> 144.8 us, 104 -> 143.1, 96 -> 144.2, 88 -> 142.8-146.4, 80 -> 147.5, 72 -> 152-154:
// one workgroup per 18 chunks = 105)
#define SMX_HEAD_SWEEP_CHUNKS_PER_WG 18
struct HeadFusedArgs {
  const float* D = nullptr; int ldd = 0;            // decoder output [B][ldd], 128 columns
  const float* W = nullptr; long ldw = 0;           // [128][k * Gp]
  const float* bias = nullptr;                      // [k * Gp]
  const void* X = nullptr; long ldx = 0; const int32_t* rows = nullptr; int x_u16 = 0;   // counts (gathered by row id)
  float* dW = nullptr; float* db = nullptr;         // gradients, laid out as W / bias
  float* part = nullptr; long slab_stride = 0;      // [workgroups][B][128]: per-workgroup slabs of d d
  int part_colmajor = 0;                            // the slabs as [workgroups][128 columns][128 cells] (B <= 128): summed by bn_wide_bwd_kernel
  float* llk_part = nullptr;                        // [B][head_fused_chunks(Gp)]
  float* sq_part = nullptr;                         // 8 sum-of-squares slots of dW per workgroup, or nullptr
  void* dtab = nullptr;                             // SMX_HEAD_FUSED_TAB_BYTES of scratch: the waves' split view of d for dW (written and read by the launch)
  int B = 0, G = 0, Gp = 0, likelihood = 0;
  float grad_scale = 1.f;
  int n_gt = 0, per_wg = 0, n_chunks = 0;           // set by the launcher: units (gene tiles), units per workgroup, likelihood partials per cell
  long long* dbg = nullptr;                         // development builds (SMX_HF_STAMPS): 128 cycle stamps
};
bool head_fused_supported(int B, int Hp, int Gp, int k);
int head_fused_grid(int Gp);
int head_fused_chunks(int Gp);   // likelihood partials per cell of a launch (<= 256)
// the launch (*n_slabs workgroups leave a slab of d d each, *n_sq sum-of-squares slots) and the ordered sum of the slabs into dd_out [B][128]
int launch_head_fused(hipStream_t st, const HeadFusedArgs& a, int* n_slabs, int* n_sq, hipEvent_t stop = nullptr);   // stop: recorded by the (last) launch's own packet
int launch_head_fused_reduce(hipStream_t st, const HeadFusedArgs& a, int n_slabs, float* dd_out);
long head_fused_bytes(int B, int G, int Gp, int k);
int head_fused_prepare();   // the kernels' dynamic-LDS limits (model creation; idempotent)

// ---- grouped weight gradients with K = the minibatch (smx_headbwd.hip) ---------------------------
struct WgradProblem {
  const float* A; int lda; int a_mode; int log1p;   // a_mode 0: A [K][M]; 1: rows of the float32 count store gathered by `rows`; 2: uint16 store
  const int32_t* rows;
  const float* Bm; int ldb;
  float* C; int ldc; int M, N;
  float* colsum; float* sq_part;
  int start, n_mt, n_nt;
  int panel;   // > 0: a wide M (a gene panel) with N <= 128 in the panel form (smx_panel.h, role 0): this many workgroups walk its 32-row tiles
};
struct WgradGroup { int n; int B; int b3; int pad; int starts[SMX_GROUP_MAX]; WgradProblem p[SMX_GROUP_MAX]; };   // starts[k] = p[k].start (INT_MAX beyond n): ONE scalar load finds a workgroup's problem;   // b3: bf16 x 3 MFMAs (smx_device.h)
bool wgrad_supported(const GemmArgs& g, int B);

// ---- minibatch-contracted weight gradients of a wide panel: one workgroup per 32 entries of the wide axis (smx_panel.h) ----
#define SMX_PANEL_MIN_WIDE 4096   // entries of the wide axis from which the panel form replaces the 32 x 32-tile kernels
struct PanelProblem {
  const void* big = nullptr; long ld_big = 0;      // the panel [cells][wide entries] (float32, or the uint16 count store: big_mode 2)
  int big_mode = 0, log1p = 0; const int32_t* rows = nullptr;   // gather by row id (nullptr: identity), log1p on the way
  long sub_stride = 0; int n_sub = 1;              // role 1: planes of the head (offset of plane p in the panel's AND the output's columns)
  const float* S = nullptr; int ldS = 0; int n_st = 0;          // narrow operand [cells][ldS]: n_st <= 4 tiles of 32 columns
  float* out = nullptr; long ld_out = 0;           // role 0: [wide][ld_out]; role 1: [32 n_st][ld_out]
  float* big_colsum = nullptr;                     // column sums of the panel over the cells (role 1: the bias gradient) or nullptr
  float* s_colsum = nullptr;                       // column sums of S (role 0: the layer's bias gradient) or nullptr
  float* sq_part = nullptr;                        // 8 sum-of-squares slots per workgroup or nullptr
  int n_wt = 0; int B = 0;                         // tiles of the wide axis; cells
};
int panel_grid(int units);   // workgroups for `units` tiles: whole rounds of at most two per CU, evenly filled
bool panel_dw_supported(const HeadBwdArgs& a);
int launch_panel_dw(hipStream_t st, const HeadBwdArgs& a);   // the output head's dW / db (role 1) in place of launch_out_head_bwd's role 0
// beside: one product of smx_dgemm.hip's form (b_nmajor) that depends on nothing the group writes -- the same launch carries it when
// the group holds a panel problem, a launch of its own follows otherwise
int launch_wgrad_group(hipStream_t st, const GemmArgs* list, int n, int B, int bf16x3 = 0, const GemmArgs* beside = nullptr);

// ---- FactorVAE discriminator (smx_factor.hip; sisua/models/fvae.py:9-18, Kim & Mnih 2018 Algorithm 2) -------------
enum { ST_PERMUTE = 66 };   // Philox stream of the permute_dims uniforms
struct PermuteArgs {
  const float* z = nullptr; int ldz = 0;     // latent sample [B][ldz]
  float* zz = nullptr; int ld = 0;           // [2B][ld]: rows [0, B) = z, rows [B, 2B) = z with every column permuted over the batch
  int B = 0, D = 0;
  NoiseKey nk{0, 0, 0, 0, nullptr};
  const int32_t* rows = nullptr; uint32_t cell_base = 0;
  const float* inj_u = nullptr; int inj_ld = 0;   // injected uniforms [B][inj_ld] (smx_set_noise stream 66)
};
int launch_permute_dims(hipStream_t st, const PermuteArgs& a);

#define SMX_DISC_MAX_GROUPS 8   // label variables of SemiFVAE (their classes: 32 in all, the logit layer's width)
struct DiscHeadArgs {
  const float* logits = nullptr; int n_slabs = 1; long slab_stride = 0; int ld = 0;   // [S][2B][ld] split-K slabs of h W_out
  const float* bias = nullptr; int n_out = 1; int B = 0;
  float gamma = 0.f, alpha = 0.f, inv_gb = 0.f;
  int backward = 1;
  // SemiFVAE: the resident one-hot label variables; variable g owns the logits gstart[g] .. gstart[g + 1] - 1 (n_groups = 0: unsupervised)
  int n_groups = 0; int gstart[SMX_DISC_MAX_GROUPS + 1] = {0}; const float* Y[SMX_DISC_MAX_GROUPS] = {nullptr}; int ldy[SMX_DISC_MAX_GROUPS] = {0};
  const int32_t* rows = nullptr; const uint8_t* mask = nullptr;
  float* u_tc = nullptr;     // [B][32]  d J_vae / d logits (rows of z)
  float* u_d = nullptr;      // [2B][32] d J_d / d logits
  float* tc_cell = nullptr;  // [B]  d(z_b)
  float* dl_cell = nullptr;  // [2B] 1/2 softplus(-d(z_b)) | 1/2 softplus(d(z_perm_b))
  float* llk_y = nullptr;    // [B] -mask CE (SemiFVAE) or nullptr
};
int launch_disc_head(hipStream_t st, const DiscHeadArgs& a);

// ---- dataset kernels (smx_data.hip) ------------------------------------------------------------
enum { ST_CORRUPT_SELECT = 80, ST_CORRUPT_BINOMIAL = 81 };   // Philox streams of the on-device corruption
enum { ST_GENERATE = 90, ST_GENERATE_MU = 91 };               // ... of the on-device generator (smx_dataset_generate_lognormal)
int launch_generate_lognormal(hipStream_t st, void* X, int u16, long ld, long N, int G, uint64_t seed, uint32_t cell_base, float density,
                              float* mu);
struct CorruptArgs {
  float* X = nullptr; long ld = 0; long N = 0; int G = 0; int u16 = 0;   // u16: X is uint16_t*
  uint32_t k0 = 0, k1 = 0, cell_base = 0;
  uint64_t prefix = 0;                  // radix-select prefix (hist passes) / threshold key (apply)
  uint64_t thr_binom = 0;               // floor(retain_rate * 2^32)
  unsigned long long* hist = nullptr;   // [256] byte histogram (hist passes) / [1] corrupted-entry counter (apply)
};
int launch_row_stats(hipStream_t st, const float* X, int u16, long ld, long N, int G, float* lgx1, double* logcount);
// compact sparse store: per-row constants, and the per-step reader (minibatch rows -> dense float32 tile)
int launch_csr_row_stats(hipStream_t st, const int64_t* indptr, const float* vals, long N, float* lgx1);
int launch_csr_expand(hipStream_t st, const int64_t* indptr, const int32_t* cols, const float* vals, const int32_t* rows, long row0,
                      int B, long ld, float* out);
int launch_library_stats(hipStream_t st, const double* logcount, long N, double* stats, float* library);
int launch_corrupt_hist(hipStream_t st, const CorruptArgs& a, int pass);
int launch_corrupt_apply(hipStream_t st, const CorruptArgs& a);

}  // namespace smx
