/* sisua_hip.h -- C-ABI of the MI355X-native SISUA training hot path.
 *
 * The reference (trungnt13/sisua) has no FFI: its hot path is reached through the
 * Python class surface SingleCellModel.fit/predict/encode/decode
 * (sisua/models/single_cell_model.py:67-306), which hands every minibatch to
 * odin-ai / TensorFlow.  This library replaces everything below that class
 * surface.  Each entry point cites the reference interface it replaces; the
 * ctypes binding a maintainer would add is sisua_amd/_hip.py (see
 * INTEGRATION.md).
 *
 * Conventions: plain pointers and sizes, no torch / HIP types.  Every function
 * returns 0 on success and a negative smx_status otherwise; smx_last_error()
 * returns a thread-local message.  The caller owns every host buffer, the
 * library owns every device buffer.  One host thread (process) per GPU; all
 * kernels of a model run on one HIP stream owned by the model.  Host arrays
 * are dense row-major with LOGICAL shapes (the padded HBM layout is private).
 */
#ifndef SISUA_HIP_H_
#define SISUA_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMX_ABI_VERSION 4
#define SMX_MAX_LAYERS 8
#define SMX_MAX_LABELS 4

typedef enum {
  SMX_OK = 0,
  SMX_ERR_INVALID = -1,   /* bad argument / shape / state            */
  SMX_ERR_HIP = -2,       /* a HIP runtime call failed               */
  SMX_ERR_NOMEM = -3,
  SMX_ERR_COMM = -4,      /* RCCL not loadable or a collective failed */
  SMX_ERR_NAN = -5        /* terminate_on_nan (configs/base.yaml:59)  */
} smx_status;

/* Model families: sisua/models/vae.py:15-16 (VAE), dca.py:13-28, scvi.py:20-171,
 * vae.py:19-44 (SISUA = VAE + label heads; MISA = SISUA with mixture label heads, vae.py:47-98). */
/* SMX_MODEL_SCALE: scale.py:13-49 (SCALE, Xiong et al. 2019): VAE whose prior over z is a trainable mixture of
 * n_components diagonal Gaussians, KL term by one-sample Monte Carlo (`analytic=False`).  With label heads (n_labels > 0) it
 * is SCALAR (scale.py:52-59: SCALE + SISUA's semi-supervised heads). */
/* SMX_MODEL_FVAE: fvae.py:9-18 (FVAE / SemiFVAE; Kim & Mnih 2018): VAE + a discriminator on z whose logit estimates the
 * total correlation; the VAE's tensors follow -ELBO + gamma TC, the discriminator's tensors its own classification loss
 * (z against z with every dimension permuted over the minibatch), both in the same step.  Label variables (behind the
 * observed outputs; SMX_LABEL_ONEHOT, 32 classes in all) make it the semi-supervised form: one logit per class of every
 * variable, TC logit = the logsumexp of all of them, each variable's masked cross-entropy under the softmax of its own. */
typedef enum { SMX_MODEL_VAE = 0, SMX_MODEL_DCA = 1, SMX_MODEL_SCVI = 2, SMX_MODEL_SISUA = 3, SMX_MODEL_SCALE = 4,
               SMX_MODEL_FVAE = 5,
               SMX_MODEL_SCALE_TRIL = 6 /* SCALE with covariance = 'tril' / 'full' (scale.py:28,35): a lower-triangular scale factor per
                                          component -- prior/scale is [n_components * latent_dim][latent_dim] (diag = softplus + 1e-5) */,
               SMX_MODEL_SCALE_POST = 7 /* scale.py:26,38-47 read literally: the latent POSTERIOR is a mixture of n_components (2 .. min(latent_dim,
                                          8)) diagonal Gaussians from a (1 + 2 n_components) * latent_dim wide latent head (logits in the first columns
                                          of plane 0 | locations | raw scales), standard-normal prior, Monte-Carlo KL; no prior tensors */ } smx_model_kind;
/* Count likelihoods selected by RVmeta.posterior (configs/base.yaml:32-40,
 * data/_single_cell_base.py:518-533). */
/* SMX_LLK_MSE: RVmeta(dim, 'mse') (the reference's tests/test_singlecell_models.py:82-91, 97-100): a deterministic output, ONE
 * parameter plane (the mean), log p(x) := -mean_g (x - mean)^2 -- exactly minus tf.losses.mse.  No marginal-likelihood scoring. */
typedef enum { SMX_LLK_NB = 0, SMX_LLK_ZINB = 1, SMX_LLK_NBD = 2, SMX_LLK_ZINBD = 3, SMX_LLK_MSE = 4 } smx_likelihood;
/* Label heads of SISUA (vae.py:19-44): NB (ADT counts), one-hot categorical (cell types); of MISA (vae.py:47-98): every
 * label dimension a mixture of label_components (2..4) negative binomials (SMX_LABEL_MIXNB) or, for continuous labels, normals
 * (SMX_LABEL_MIXGAUSS, 'mixgaussian', vae.py:86-92); SMX_LABEL_MIXTRIL ('mixtril', the class's docstring example vae.py:58): ONE
 * mixture of label_components full-covariance Gaussians over the whole label vector (lower-triangular scale factors; label_dim <= 64;
 * the head then has label_components * (2 + label_dim) planes); SMX_LABEL_MIXZINB: MISA(zero_inflated=True), vae.py:76-84 -- the
 * components of SMX_LABEL_MIXNB zero-inflated, a fourth group of label_components gate-logit planes. */
/* SMX_LABEL_NBD / ZINB / ZINBD: the remaining count posteriors of RVmeta as heads (vae.py:30 "'onehot'/'nbd'/'nb'"; the second OUTPUT of
 * tests/test_singlecell_models.py:133-134 is 'nbd'): planes as for the gene output -- (mean, dispersion) through softplus / softplus1,
 * (log total_count, logits, gate logits), (mean, dispersion, gate logits). */
typedef enum { SMX_LABEL_NB = 0, SMX_LABEL_ONEHOT = 1, SMX_LABEL_MIXNB = 2, SMX_LABEL_MIXGAUSS = 3, SMX_LABEL_MIXTRIL = 4, SMX_LABEL_MIXZINB = 5,
               SMX_LABEL_NBD = 6, SMX_LABEL_ZINB = 7, SMX_LABEL_ZINBD = 8 } smx_label_likelihood;
typedef enum { SMX_ACT_RELU = 0, SMX_ACT_LINEAR = 1 } smx_activation;

/* Constructor arguments of SingleCellModel / SCVI / SISUA / DeepCountAutoencoder
 * (single_cell_model.py:74-97, scvi.py:33-48, vae.py:40-44, dca.py:16-28) plus
 * the optimiser block of configs/base.yaml:45-50. */
typedef struct {
  int32_t abi_version;                 /* must be SMX_ABI_VERSION */
  int32_t model;                       /* smx_model_kind */
  int32_t likelihood;                  /* smx_likelihood */
  int32_t n_genes;                     /* G */
  int32_t latent_dim;                  /* D */
  int32_t n_enc, enc_units[SMX_MAX_LAYERS];
  int32_t n_dec, dec_units[SMX_MAX_LAYERS];
  int32_t n_encl, encl_units[SMX_MAX_LAYERS];   /* scvi library encoder */
  int32_t n_labels, label_dim[SMX_MAX_LABELS], label_llk[SMX_MAX_LABELS];
  int32_t label_components[SMX_MAX_LABELS];   /* SMX_LABEL_MIXNB / MIXGAUSS / MIXTRIL: mixture components (MISA n_components, vae.py:77) */
  /* outputs[1:] of the reference's constructors (single_cell_model.py:74-97; tests/test_singlecell_models.py:129-141
   * `VAE(outputs=[RVmeta(G, 'zinb'), RVmeta(P, 'nbd')])`; scvi.py:168-169 `pY = [p(d) for p in self.posteriors[1:]]`): a head with
   * label_observed[j] != 0 is a further OUTPUT variable -- fully observed (the label mask is not consulted), weight 1 instead of alpha,
   * its negative log-likelihood reported as smx_metrics.nllk_o.  Observed heads come first; any model kind but SMX_MODEL_FVAE /
   * SMX_MODEL_SCALE_POST takes them (label heads proper keep their model-kind rule). */
  int32_t label_observed[SMX_MAX_LABELS];
  /* scvi.py:55-56,66-86,136-160 `dispersion` / `inflation` of the gene output: 0 = 'full' (a Dense head per cell and gene), 1 = 'share'
   * (no head: ONE trainable vector [n_genes] shared by every cell -- tensor out1/b resp. out2/b without out1/W resp. out2/W; theta = exp of
   * it, gate logits = it), 2 = 'single' (ONE trainable scalar for every cell and gene: out1/b resp. out2/b of one element).  SMX_MODEL_SCVI only. */
  int32_t scvi_dispersion, scvi_inflation;
  int32_t n_components;                /* SMX_MODEL_SCALE: components of the mixture prior (scale.py:27), 1..32 */
  int32_t disc_units, disc_layers;     /* SMX_MODEL_FVAE: hidden width / hidden layers of the discriminator (odin: 1000, 5) */
  float gamma, disc_leak;              /* SMX_MODEL_FVAE: weight of the TC term (6.0); leaky-ReLU slope (0.2) */
  int32_t batchnorm;                   /* NetConf.batchnorm */
  int32_t log_norm;                    /* single_cell_model.py:82 */
  int32_t latent_activation;           /* dca only */
  float dropout_enc, dropout_dec, input_dropout;
  float beta, alpha;                   /* base.yaml:6-7 */
  float clip_library;                  /* scvi.py:47 */
  float bn_momentum, bn_eps;
  float lr, adam_beta1, adam_beta2, adam_eps, clipnorm;
  int32_t max_batch;                   /* largest minibatch a step will see */
  uint64_t seed;                       /* Philox key (dropout masks, eps) */
} smx_config;

/* Scalars of one step, mean over the (global) minibatch; names follow the
 * reference's logged scalars (tutorials/notebook/...ipynb:238: loss, nllk_x, KLqp). */
typedef struct {
  float loss, nllk_x, nllk_y, kl, kl_l;
  float grad_norm_max;   /* largest per-tensor gradient norm before clipping */
  int32_t nan_flag;      /* non-zero if any of the above is not finite */
  int32_t step;          /* optimiser step count after this call */
  float tc, dtc_loss;    /* SMX_MODEL_FVAE: total-correlation estimate mean d(z); the discriminator's loss (0 otherwise) */
  float nllk_o;          /* the observed extra outputs (label_observed): -mean sum of their log-likelihoods (0 without them) */
} smx_metrics;

typedef struct smx_model smx_model;

/* ---- process / device ---------------------------------------------------- */
const char* smx_last_error(void);
int smx_abi_version(void);
/* Number of visible HIP devices (0 when there is no GPU). */
int smx_device_count(void);
/* Bind this process to a device.  Replaces `CUDA_VISIBLE_DEVICES='0'` (train.py:20). */
int smx_init(int device);
int smx_synchronize(void);

/* ---- model lifetime ------------------------------------------------------ */
/* SingleCellModel.__init__ (single_cell_model.py:74-101). Glorot-uniform weights
 * are NOT drawn here: the host sets them with smx_set_tensor (keeps init RNG on
 * the Python side, shared with the oracle). */
int smx_model_create(const smx_config* cfg, smx_model** out);
int smx_model_destroy(smx_model* m);

/* Manifest of trainable tensors, in oracle order (oracle/sisua_oracle.py:manifest). */
int smx_num_tensors(const smx_model* m);
int smx_tensor_info(const smx_model* m, int index, char* name, int name_cap, int32_t* rows, int32_t* cols);
/* which: 0 = parameters, 1 = gradients of the last step (after all-reduce, before
 * clipping), 2 = Adam m, 3 = Adam v.  Replaces save_weights/load_weights
 * (single_cell_model.py:283-306) and serves the parity tests. */
int smx_get_tensor(smx_model* m, int which, int index, float* host);
int smx_set_tensor(smx_model* m, int which, int index, const float* host);
/* Batch-norm moving statistics, layer order = oracle bn_manifest; which: 0 mean, 1 var. */
int smx_num_bn_layers(const smx_model* m);
int smx_get_bn(smx_model* m, int layer, int which, float* host, int32_t* width);
int smx_set_bn(smx_model* m, int layer, int which, const float* host);
int smx_get_step(const smx_model* m, int32_t* step);
int smx_set_step(smx_model* m, int32_t step);

/* ---- data ---------------------------------------------------------------- */
/* Upload the (already split / corrupted) cells x genes matrix once; it stays
 * resident in HBM (replaces the per-step tf.data H2D copy of
 * data/_single_cell_base.py:539-602).  X [n_cells, G] dense float32 (the
 * reference's on-disk format, data/utils.py:427-431).  labels[j] is
 * [n_cells, label_dim[j]] or NULL; library [n_cells,2] = (local_mean, local_var)
 * (:568-570) or NULL; label_mask [n_cells] 0/1 or NULL (:580-591).
 * cell_id_base offsets the Philox cell ids (rank shard offset). */
int smx_dataset_upload(smx_model* m, const float* X, int64_t n_cells, const float* const* labels,
                       const float* library, const uint8_t* label_mask, int64_t cell_id_base);
/* The same with the counts stored compactly as uint16 (every count must be <= 65535): half the
 * HBM footprint and half the gather traffic of the dense float32 memmap of the reference
 * (sisua/data/utils.py:401-452), identical results -- the kernels widen on load.  Labels stay float32. */
int smx_dataset_upload_u16(smx_model* m, const uint16_t* X, int64_t n_cells, const float* const* labels,
                           const float* library, const uint8_t* label_mask, int64_t cell_id_base);

/* Compact sparse store (the dense float32 memmap semantics of sisua/data/utils.py:401-452 over the
 * non-zeros only): the counts as CSR -- indptr [n_cells + 1] (indptr[0] = 0), then column indices (< n_genes) and
 * float32 values of the non-zeros row by row; 8 bytes per non-zero on the device.  Every pass expands its minibatch's
 * rows into a dense float32 tile first, so every result is bit-identical to the float32 store.  smx_dataset_library /
 * smx_dataset_corrupt need a dense store; input dropout too (its noise is keyed by the dense store's rows). */
int smx_dataset_upload_csr(smx_model* m, const int64_t* indptr, const int32_t* cols, const float* vals, int64_t n_cells,
                           const float* const* labels, const float* library, const uint8_t* label_mask, int64_t cell_id_base);

/* The rank's shard of BASELINE.json configs[4] ("synthetic 1e6 cells x 20k genes log-normal counts"), generated ON the
 * device ("generated on-device per shard from (seed, rank)"; the reference's scaling test draws its matrix
 * with NumPy on the host, tests/test_scalability.py:22-27): n_cells rows of ONE virtual matrix whose entry (cell, gene) is
 * a function of (seed, global cell id = rank * n_cells + row, gene) only -- x = floor(LogNormal(mu_gene, 1)) kept with
 * probability `density` (0.14 leaves ~93 % zeros), capped at 65535, gene 0 >= 1.  storage_u16 != 0: the compact store
 * (40 GB for 1e6 x 20 000).  Replaces the resident matrix; no labels / library prior (VAE / DCA models).  Bit-for-bit the
 * oracle's generate_lognormal_rows up to exp() rounding at integer boundaries. */
int smx_dataset_generate_lognormal(smx_model* m, uint64_t seed, int32_t rank, int64_t n_cells, int32_t storage_u16, double density);

int64_t smx_dataset_size(const smx_model* m);

/* Library-size statistics of the RESIDENT matrix, get_library_size (sisua/data/utils.py:231-263) as the
 * reference recomputes them after corrupt() (sisua/data/_single_cell_analysis.py:110): log_counts = log(sum_g x
 * + 1e-8) per cell, local_mean / local_var over cells.  Fills the resident [n_cells, 2] library prior with
 * (local_mean, local_var) and returns them in stats[0..1] (stats may be NULL). */
int smx_dataset_library(smx_model* m, float stats[2]);

/* apply_artificial_corruption(distribution='binomial') (sisua/data/utils.py:168-228; SingleCellOMIC.corrupt,
 * _single_cell_analysis.py:78-111) IN PLACE on the resident matrix: exactly floor(dropout * nnz) of the non-zero
 * entries, chosen without replacement, become Binomial(n = x, p = retain_rate).  The draws come from the Philox
 * counter RNG keyed by (seed, cell id, gene), not from NumPy's MT19937 stream: same law, order independent,
 * bit-identical to oracle/sisua_oracle.py:corrupt_philox (the host path sisua_amd.data.corrupt keeps the
 * reference's RandomState stream bit-exactly).  Refreshes the per-row likelihood constants; call
 * smx_dataset_library afterwards as the reference does.  Same argument checks as the reference: dropout in
 * [0, 1); no-op unless 0 < dropout < 1 or 0 < retain_rate < 1. */
int smx_dataset_corrupt(smx_model* m, double dropout, double retain_rate, uint64_t seed, int64_t* n_corrupted);

/* Read back rows of the resident matrix (tests / checkpoints): X [n_rows, n_genes], the per-row constant
 * sum_g lgamma(x+1) [n_rows], the library prior [n_rows, 2]; any output may be NULL. */
int smx_dataset_read(smx_model* m, int64_t row0, int64_t n_rows, float* X, float* row_const, float* library);

/* ---- the hot path -------------------------------------------------------- */
/* One optimiser step on the cells `row_ids` of the resident matrix: forward,
 * ELBO, backward, (all-reduce), per-tensor clipnorm, Adam.  Replaces one
 * iteration of the odin Trainer loop under SingleCellModel.fit
 * (single_cell_model.py:213-236).  `out` may be NULL (no host
 * sync).  batch <= cfg.max_batch. */
int smx_train_step(smx_model* m, const int32_t* row_ids, int32_t batch, smx_metrics* out);
/* Same arithmetic, launched as one captured hipGraph (fixed batch size). */
int smx_train_step_graph(smx_model* m, const int32_t* row_ids, int32_t batch, smx_metrics* out);
/* Queue `n_steps` steps whose row ids are order[s*batch .. (s+1)*batch); no host
 * sync between steps (order = NULL: the ids staged by smx_train_stage).  `out` (may be NULL) receives the last step's metrics; a non-finite loss or gradient
 * norm is reported through out->nan_flag with status SMX_OK (terminate_on_nan is the caller's decision). */
int smx_train_steps(smx_model* m, const int32_t* order, int32_t n_steps, int32_t batch, int use_graph,
                    smx_metrics* out);
/* The row ids of the NEXT smx_train_steps call made resident ahead of it: that call is then given order = NULL (same
 * n_steps and batch) and queues its steps without the host-to-device copy of the ids -- the minibatch schedule of an epoch is
 * input data like the matrix itself (an input pipeline's prefetch; single_cell_model.py:213-236 iterates a prepared
 * tf.data stream).  The staged ids serve ONE call. */
int smx_train_stage(smx_model* m, const int32_t* order, int32_t n_steps, int32_t batch);
/* ELBO scalars of EVERY step of the last smx_train_steps call, kept on the device while the steps ran (no host
 * sync between them): host [n_steps][8] = (loss, nllk_x, nllk_y, kl, kl_l, 0, 0, 0) per step, global-minibatch
 * means.  Feeds the per-epoch train_history of SingleCellModel.fit (tests/test_singlecell_models.py:28-32 of the
 * reference reads train_history['loss'] per epoch). */
int smx_metrics_history(smx_model* m, int32_t n_steps, float* host);
/* Validation loss: eval-mode forward (moving BN stats, no dropout) + ELBO, no
 * update (valid_freq loop of BetaVAE.fit). */
int smx_eval_step(smx_model* m, const int32_t* row_ids, int32_t batch, smx_metrics* out);

/* Eval-mode forward for predict/encode/decode (single_cell_model.py:119-211):
 * writes distribution parameters into caller-owned buffers (any may be NULL).
 * Input cells: row_ids into the resident matrix, or host_x [batch,G] when
 * row_ids == NULL (host_library [batch,2] for scvi).  sample_index selects the
 * Monte-Carlo draw of eps.  Outputs: z_mean/z_scale/z_sample [batch,D];
 * l_mean/l_scale/l_sample [batch] (scvi); x_params [k,batch,G] planes in
 * likelihood order (nb/zinb: log total_count, logits, gate logits;
 * nbd/zinbd: mean, dispersion, gate logits -- already activated);
 * y_params[j] [batch, ky*P_j] raw head outputs (ky = 2 NB: log total_count | logits; 1 one-hot: logits; 3 C mixture of
 * C NB: C mixture logits | C log total_counts | C logits, each P_j wide). */
int smx_forward(smx_model* m, const int32_t* row_ids, const float* host_x, const float* host_library,
                int32_t batch, int32_t sample_index, int32_t training, float* z_mean, float* z_scale,
                float* z_sample, float* l_mean, float* l_scale, float* l_sample, float* x_params,
                float* const* y_params);

/* n_samples Monte-Carlo draws of one batch in ONE call: SingleCellModel.predict(sample_shape=n) as
 * Posterior._initialize uses it (sisua/analysis/posterior.py:172-182, sample_shape = 10).  The encoders run once (eval
 * mode); draw s re-samples the latents with Philox sample index s and decodes -- the same draws as n_samples calls of
 * smx_forward(sample_index = s), and the same numbers to rounding (a host batch's draws are decoded as rows of one pass,
 * as in smx_predict; bit for bit with flag "stacked_scoring" = 0 and for resident rows).  z_mean / z_scale [batch, D] and l_mean / l_scale [batch] once; z_samples
 * [n_samples, batch, D]; l_samples [n_samples, batch]; x_params [n_samples, k, batch, n_genes]; y_params[j]
 * [n_samples, batch, ky * P_j].  Any output may be NULL. */
int smx_forward_samples(smx_model* m, const int32_t* row_ids, const float* host_x, const float* host_library, int32_t batch,
                        int32_t n_samples, float* z_mean, float* z_scale, float* z_samples, float* l_mean, float* l_scale,
                        float* l_samples, float* x_params, float* const* y_params);

/* SingleCellModel.predict(inputs, sample_shape, batch_size) over a whole host matrix in ONE call
 * (sisua/models/single_cell_model.py:153-211: "predict on minibatches then return a single distribution by
 * concatenation").  host_x [n_cells, n_genes] (host_library [n_cells, 2] for scvi) is walked in minibatches of `batch`
 * (<= max_batch; the last one may be smaller) and every result is written straight to its final place:
 * z_mean / z_scale [n_cells, D]; l_mean / l_scale [n_cells]; z_samples [n_samples, n_cells, D]; l_samples
 * [n_samples, n_cells]; x_params [n_samples, k, n_cells, n_genes]; y_params[j] [n_samples, n_cells, ky * P_j].  Same draws
 * as smx_forward_samples batch by batch (draw s uses Philox sample index s; the noise of a cell is keyed by its index
 * within its minibatch, as there); the same numbers bit for bit at one draw, to rounding with several (their decodes
 * then run as rows of one pass, flag "stacked_scoring").  Any output may be NULL. */
int smx_predict(smx_model* m, const float* host_x, const float* host_library, int64_t n_cells, int32_t batch,
                int32_t n_samples, float* z_mean, float* z_scale, float* z_samples, float* l_mean, float* l_scale,
                float* l_samples, float* x_params, float* const* y_params);

/* The same walk over a host matrix, but what leaves the device is a STATISTIC of the gene output instead of its parameter planes -- what
 * the reference's callers ask of predict()'s result: `y.mean()`, `.variance()`, `.log_prob(x)` of the output distribution or of its
 * count distribution without the zero-inflation wrapper (sisua/analysis/posterior.py:187-255: 'reconstructed' / 'imputed').  The planes
 * are 4 k G bytes per cell and draw (24 KB at 1998 genes, zinb) and a fresh result array of that size is first-touched by the copy;
 * a mean is a third of it, the mean over the draws 1 / (3 n_samples), a log_prob 4 bytes.  Same passes, same draws as smx_predict.
 *   stat 0 mean, 1 variance: out [n_samples, n_cells, n_genes];  2 mean averaged over the draws: out [n_cells, n_genes];
 *   stat 3 log_prob (summed over the genes: Independent(..., 1)): out [n_samples, n_cells], of `target` [n_cells, n_genes] or, target = NULL,
 *          of host_x itself;  count_only != 0: the count distribution (NB / NBD) of a zero-inflated output. */
int smx_predict_stat(smx_model* m, const float* host_x, const float* host_library, int64_t n_cells, int32_t batch, int32_t n_samples,
                     int32_t stat, int32_t count_only, const float* target, float* out);

/* Decoder only (SingleCellModel.decode, single_cell_model.py:141-151; scvi.py:108-171):
 * z [batch,D] (and l [batch] for scvi) -> the same x_params / y_params as smx_forward,
 * eval mode. */
int smx_decode(smx_model* m, const float* z, const float* l, int32_t batch, float* x_params, float* const* y_params);

/* Importance-weighted marginal log-likelihood, the scoring path of Posterior.cal_marginal_llk ->
 * scm.marginal_log_prob(**Xs, sample_shape=100) (sisua/analysis/posterior.py:941-976): the encoder runs once,
 * then the n_samples draws (z of draw s from Philox sample index s) go through decoder, output head and forward-only
 * likelihood as rows of one pass (up to 16 384 rows at a time; flag "stacked_scoring" = 0: draw by draw) with
 * a running log-sum-exp per cell on the device.  mllk[batch] = log mean_s p(x|z_s) p(z_s) / q(z_s|x);
 * llk_mean[batch] = mean_s log p(x|z_s) (may be NULL).  Cells: row_ids or host_x (+ host_library for scvi). */
int smx_marginal_llk(smx_model* m, const int32_t* row_ids, const float* host_x, const float* host_library, int32_t batch,
                     int32_t n_samples, float* mllk, float* llk_mean);

/* Posterior-predictive scoring, Posterior.cal_llk (sisua/analysis/posterior.py:919-938): the cells given by
 * row_ids / host_x are encoded once, then n_samples posterior draws are decoded and the output distribution is
 * scored against each target matrix (host [batch, n_genes]; a NULL entry = the input cells themselves) with a
 * running log-sum-exp over the draws on the device.  out[(t*2 + j)*batch + b] = logsumexp_s log p_j(target_t[b] |
 * z_s) - log n_samples, j = 0 the model's output distribution ("reconstructed"), j = 1 its count distribution
 * without the zero-inflation gate ("imputed", posterior.py:218-225; equals j = 0 for nb / nbd). n_targets <= 4. */
int smx_score_llk(smx_model* m, const int32_t* row_ids, const float* host_x, const float* host_library,
                  const float* const* targets, int32_t n_targets, int32_t batch, int32_t n_samples, float* out);

/* Test hook: inject noise for the NEXT step instead of Philox.  stream ids as in
 * oracle/sisua_oracle.py (STREAM_*); data [batch, width] holds eps values or
 * dropout multipliers.  smx_clear_noise() returns to Philox. */
int smx_set_noise(smx_model* m, int32_t stream, const float* data, int32_t batch, int32_t width);
int smx_clear_noise(smx_model* m);

/* ---- data parallel (one process per GPU) ---------------------------------- */
/* 128-byte RCCL unique id, created on rank 0 and handed to every rank by the
 * host (torch.distributed / a file store). */
int smx_comm_unique_id(uint8_t id[128]);
/* Join the communicator; afterwards every train step all-reduces the flat
 * gradient buffer (+ BN batch stats + metrics) once over xGMI.  The loss is scaled by 1 / (batch * world), so
 * the summed buffer holds the gradient of the GLOBAL minibatch mean; smx_get_tensor(which = 1) returns it.
 * On failure the model is left without a communicator (world 1) and SMX_ERR_COMM is returned. */
int smx_comm_init(smx_model* m, int rank, int world, const uint8_t id[128]);
int smx_comm_world(const smx_model* m);
int smx_comm_rank(const smx_model* m);
/* How an (eager) training step exchanges its gradients right now: 0 no collective (world 1), 1 ONE all-reduce of the flat buffer between
 * the backward pass and the optimiser (north star; RCCL or the tests' loopback), 2 the two-bucket chain -- the heads' gradients (3/4 of the
 * bytes) all-reduced, normed and applied on a communication stream beside the rest of the step, the front bucket all-reduced on the model's
 * stream --, 3 the hand-written exchange over IPC-mapped peer buffers, one bucket, ONE launch per all-reduce on the model's stream, 4 the
 * hand-written exchange's two-bucket form of round 4 (both buckets on the communication stream; only under the library's own rule). */
int smx_comm_form(const smx_model* m);
/* Ask for a form: 1, 2 or 3 as above (3 needs smx_comm_p2p_init, 1 / 2 a communicator), or 0 = the library's own rule (the hand-written
 * exchange whenever it is attached; two buckets from 3 MB of head gradients, SMX_DP_BUCKETS=1|2 overrides).  A COLLECTIVE call when it
 * asks for 2 (the heads' communicator is split off on first use): every rank, same order.  sisua_amd/parallel.py measures the available
 * forms on the job's own steps and sets the fastest on every rank (calibrate_forms); nothing asked = nothing measured. */
int smx_comm_set_form(smx_model* m, int form);
/* Flag "opt_shard" (smx_set_flag; off by default; data parallel, the chained form -- it is taken whenever the flag is set): the output and
 * label heads' optimiser state is SHARDED over the ranks -- reduce-scatter of their gradient bucket, per-tensor clipnorm + Adam on this rank's
 * 1 / world slice, all-gather of the updated parameters (the same wire bytes as the all-reduce; 1 / world of the optimiser's memory traffic).
 * Parameters stay replicated and bit-identical on every rank; a rank's gradient buffer (smx_get_tensor, which = 1) holds the heads' REDUCED
 * gradient inside its slice only, and the Adam MOMENTS of the heads outside a rank's slice go stale.  smx_opt_gather
 * (a collective: every rank calls it, between training calls) all-gathers both moments, after which smx_get_tensor(which = 2 | 3) of a head
 * tensor -- refused while stale -- returns the job's moments on every rank (checkpoints); smx_train_steps calls it itself in front of steps that will
 * not take the sharded chain (a captured graph, the flag switched off since).  RCCL or the loopback communicator; with the
 * hand-written exchange the flag is ignored (the all-reduce form runs). */
int smx_opt_gather(smx_model* m);
/* The same all-reduce as a hand-written two-shot exchange over peer-mapped buffers instead of RCCL (reduce-scatter
 * + all-gather of the flat buffer through HIP-IPC-mapped peer memory over xGMI; sums in rank order -- bitwise the same on every
 * rank): smx_comm_p2p_export allocates this rank's communication region and returns its two 64-byte IPC handles (the flat
 * gradient buffer | the region); the host gathers the `world` x 128 bytes in rank order (control plane) and hands them to
 * smx_comm_p2p_init, after which every training step's collective (and SyncBatchNorm's small ones) takes this path -- with or
 * without an RCCL communicator (call smx_comm_init first when both are wanted).  world <= 8 (one node).  Waits on peers are
 * bounded (SMX_P2P_TIMEOUT_S, 30 s): a timed-out wait marks the step void on every rank -- smx_train_step* / smx_eval_step return
 * SMX_ERR_COMM on every call with a metrics read-back until smx_comm_p2p_error has read (and cleared) the word; the Python Engine does
 * that when it raises.  Recovery: restore the last checkpoint on every rank, re-attach the communicator. */
int smx_comm_p2p_export(smx_model* m, int world, uint8_t handles[128]);
int smx_comm_p2p_init(smx_model* m, int rank, int world, const uint8_t* all_handles);
int smx_comm_p2p_error(smx_model* m, int32_t* error);
/* Measurement hook (bench.py, N > 1): the step's collective ALONE -- `iters` all-reduces of the (zeroed) flat gradient buffer
 * through whichever path the steps take (RCCL, the peer-to-peer exchange, the loopback communicator), timed with events on the
 * model's stream after three untimed calls; every rank must call it with the same `iters`.  us_per_call: this rank's average;
 * floats (may be NULL): the buffer's length.
 * The gradient buffer is scratch between steps: nothing of the model's state changes. */
int smx_comm_time_allreduce(smx_model* m, int iters, float* us_per_call, int64_t* floats);
/* Which communication library the process is bound to, and the HIP runtime both it and this library run on
 * (RCCL is resolved as the sibling of the loaded libamdhip64: ROCm's, or torch's bundled copy when torch was
 * imported first; SMX_RCCL_PATH overrides).  rccl_version: ncclGetVersion code.  Any output may be NULL. */
int smx_comm_library(char* rccl_path, int rccl_cap, char* hip_path, int hip_cap, int32_t* rccl_version);
/* SyncBatchNorm (SURVEY.md 8e caveat i; opt-in): BatchNorm statistics over the GLOBAL minibatch, as the
 * single-process reference computes them -- one extra all-reduce of [world][2][H] column statistics per
 * BatchNorm layer forward and one backward.  Off (default): per-replica statistics, ONE all-reduce per step.
 * Every rank must run the same batch size. */
int smx_comm_set_sync_bn(smx_model* m, int on);
/* Test hook: join n models of THIS process (one device; each driven by its own host thread) into a loopback
 * communicator -- rank i = models[i].  Their steps all-reduce through events + a summing kernel instead of RCCL,
 * so the world > 1 arithmetic of the step can be checked on a one-GPU box.  Eager launches only. */
int smx_comm_init_local(smx_model* const* models, int n);

/* ---- host-side helper ------------------------------------------------------ */
/* Visit order of one epoch under a streaming shuffle buffer of `buffer` cells (tf.data .shuffle(1000) after .cache,
 * before .batch: sisua/data/_single_cell_base.py:597-600): at step t the element picks[t] % len(buffer) leaves the
 * buffer and the next unseen cell takes its place.  picks[n_obs]: non-negative random integers from the caller's
 * generator (the stream that defines the order); out[n_obs] receives the cell indices.  No device work. */
int smx_shuffle_order(int32_t n_obs, int32_t buffer, const int64_t* picks, int32_t* out);

/* ---- developer knobs -------------------------------------------------------- */
/* Tile / split-K / share sweeps, A/B of launch forms and test hooks (e.g. "score_rows", "predict_stage_floats", "no_sq_partials",
 * "adam_wide_share") live in ONE registry instead of an environment variable each: smx_set_tuning sets a knob for the process,
 * smx_clear_tuning removes one (name = NULL or "": all); the environment variable SMX_TUNING="name=value,name=value" presets them
 * (the scripts under tools/).  docs/LAB_NOTES.md lists the knobs and their defaults; none of them is needed to USE the library -- the switches a user
 * may need are the environment variables of INTEGRATION.md section C.  A model's launch-form flags read "no_<flag>" when it is created. */
int smx_set_tuning(const char* name, double value);
int smx_clear_tuning(const char* name);

/* ---- code-path switches ---------------------------------------------------- */
/* The training step has two forms of several stages: the default wide / fused kernels and the separate-launch forms
 * they replaced (which eval, predict and the scoring paths always use).  name: "head_loss" (output product fused with
 * the likelihood), "front" (latent sample + first decoder product inside BatchNorm-forward), "bwd_front" (d h inside
 * BatchNorm-backward + the weight gradients grouped at the end), "head_bwd" (both backward products of the output
 * head in one launch), "wgrad" (minibatch-contracted weight gradients as the wide kernel), "scvi_fused" (scvi
 * training step: library latent + softmax-rate head + likelihood + their backward as one row-local launch,
 * scvi.py:88-171), "twin" (scvi: the first layers of the encoder and of the library encoder, and pairs of output
 * heads, side by side in one launch), "label_ride" (SISUA / MISA: the label heads' d d as extra slabs of the output
 * head's backward launch, their weight gradients in the grouped launch at the end), "act_epilogue" (layers without
 * BatchNorm and dropout, e.g. the FactorVAE discriminator: bias + activation and the activation's derivative in the
 * products' store paths), "stacked_scoring" (smx_marginal_llk, smx_score_llk and smx_predict with several draws: all
 * posterior draws of a batch as rows of ONE decoder pass, in the scoring calls the output head fused with the
 * likelihood; 0 = one decoder pass per draw; SMX_NO_STACKED_SCORING).
 * "head_fused" (wide panels -- at least 4096 genes, 128 decoder columns, at most 128 cells, no label heads: the output product, the
 * likelihood AND both backward products of the head as one launch that owns a tile of 32 genes from the raw weights to their
 * gradients, smx_headfused.hip; 0 = the fused head + the two wide backward kernels).
 * "head_sweep" (with "head_fused", one GPU, eager steps: clip + Adam of the heads' tensors -- 3/4 of the parameters at 20 000 genes -- as a
 * background sweep of a fixed number of workgroups on a second stream, between this step's output head and the next step's; every
 * smx_train_steps call ends with the sweep joined; 0 = riders of the backward chain + the optimiser launch; same bits either way).
 * "bf16x3": the training products of the output head (the fused head, both products of its backward, the first layer's
 * weight gradient) from bf16 MFMAs on operands split three ways in registers (f32 accuracy to one rounding of a product;
 * 0.375 of the f32 MFMAs' cycles, on the matrix pipe): 1 always, 0 never (exact f32 MFMAs), -1 (default) from the head's
 * width -- SMX_BF16X3 in the environment sets the same default.
 * value 1 = default form, 0 = separate launches.  Results agree to rounding;
 * used for A/B measurements and by the parity tests of both forms.  Defaults may also be set with SMX_NO_HEAD_LOSS /
 * SMX_NO_FRONT / SMX_NO_BWD_FRONT / SMX_NO_HEAD_BWD / SMX_NO_WGRAD / SMX_NO_SCVI_FUSED / SMX_NO_TWIN /
 * SMX_NO_LABEL_RIDE / SMX_NO_ACT_EPILOGUE in the environment. */
int smx_set_flag(smx_model* m, const char* name, int value);

/* ---- measurement ---------------------------------------------------------- */
/* HIP-event timing of one named kernel class inside eager steps, on the model's
 * stream.  kernel: "out_head" (the fused output product + likelihood; 8 idempotent launches per event pair),
 * "out_head_product" (the same kernel without the likelihood: what the fused kernel's time is compared with), "loss"
 * (the standalone likelihood kernel, flag head_loss = 0), "gemm_enc_fwd", "gemm_out_fwd", "gemm_out_bwd" (dW + dX of the head),
 * "gemm_enc_dw", "bn_fwd", "bn_bwd", "adam", "allreduce", "step", or "null" (an event pair around
 * nothing: the overhead to subtract from single-kernel timings).  "out_head@N" / "out_head_product@N" / "loss@N": N
 * launches per event pair instead of 8 (N = 1: the single launch rocprofv3's per-kernel duration is compared with).
 * Enable, run steps, read. */
int smx_timing_enable(smx_model* m, const char* kernel);
int smx_timing_read(smx_model* m, double* total_ms, int64_t* launches);
/* Algorithmic bytes (fwd+bwd loss kernel = (4+8k)G + 16D + 4 per cell). */
int64_t smx_loss_bytes_per_cell(const smx_model* m);
/* Algorithmic bytes of ONE launch of the fused output head (smx_headfused.hip: W_out + bias read, dW_out + db written, decoder output,
 * counts, d d, likelihood partials) when a training step of `batch` cells takes it -- a wide panel (>= 4096 genes), 128 decoder columns,
 * batch <= 128, no label heads --, else 0: bench.py's roofline entry at the C5 width. */
int64_t smx_head_fused_bytes(const smx_model* m, int32_t batch);

/* ---- kernel-level entry points (parity tests of single kernels) ------------ */
/* Fused count log-likelihood forward+backward over host planes [k][B][G]:
 * llk[B] and grads [k][B][G] (d llk / d plane, unscaled).  direct != 0: planes
 * are (mean, dispersion, gate) (scvi.py:151-164). */
int smx_k_count_llk(int likelihood, int direct, const float* x, const float* planes, int32_t B, int32_t G,
                    float* llk, float* grads);
/* The optimiser launch by itself over caller-given tensors (per-tensor clipnorm + Adam, row a-16): n_tensors
 * tensors of sizes[i] floats, concatenated in params / grads / m / v (host arrays; params, m, v updated in
 * place).  step = the 1-based count t of this update (lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) is evaluated on the
 * device as in a training step).  norms[n_tensors] (may be NULL): gradient norms before clipping. */
int smx_k_adam(int32_t n_tensors, const int32_t* sizes, float* params, const float* grads, float* m, float* v,
               int32_t step, float lr, float beta1, float beta2, float eps, float clipnorm, float* norms);
/* C[M,N] = op(A) * op(B) in fp32 on the MFMA path; transA: A given as [K,M];
 * transB: B given as [N,K]; split_k >= 1 (slabs summed on return).  tile_cfg 0: the library's choice of LDS tile; 100: the
 * direct bf16 x 3 form for deep contractions (transA = 0, K >= 512); 101 / 102: the minibatch-contracted weight-gradient forms
 * (transA = 1, transB = 0; 32 x 32 tiles / the gene-tile-owner panel form, N <= 128) -- test entries for those kernels. */
int smx_k_gemm(int transA, int transB, const float* A, const float* B, int32_t M, int32_t N, int32_t K,
               int32_t split_k, int32_t tile_cfg, float* C);
/* The whole output head of a training step at a wide gene panel in ONE launch (smx_headfused.hip; rows a-9, a-10 / a-11 and the head's
 * part of a-16): P = d W + bias (never stored) -> count log-likelihood of x and dP = grad_scale * d llk / d P (never stored) -> dW = d^T dP,
 * db = colsum(dP), dd = dP W^T.  Host arrays: x [B][G] counts (u16 != 0: through the uint16 store), d [B][128], W [128][k][G], bias [k][G]
 * (k = 2 NB / NBD, 3 ZINB / ZINBD); B <= 128, G >= 4096 after padding to 32.  Out: llk [B], dW [128][k][G],
 * db [k][G], dd [B][128], sumsq = sum of squares of dW (may be NULL); us (may be NULL): average device time of `reps` launches. */
int smx_k_head_fused(int likelihood, int u16, const float* x, const float* d, const float* W, const float* bias, int32_t B, int32_t G,
                     float grad_scale, int32_t reps, float* llk, float* dW, float* db, float* dd, float* sumsq, float* us);
/* The same launch `launches` (>= 2) times on the same inputs, every launch after the first compared on the device, bit for bit, with what
 * the first left (dW, db, the d d slabs, the likelihood partials): *n_differ = launches that differed, *first_word (may be NULL) = index of
 * the first differing word within [dW | db | slabs | partials] or -1.  Regression test of two hardware behaviours that made this kernel's
 * results timing-dependent (tools/isa_lint.py rules R1, R2). */
int smx_k_head_fused_stress(int likelihood, int u16, const float* x, const float* d, const float* W, const float* bias, int32_t B, int32_t G,
                            float grad_scale, int32_t launches, int32_t* n_differ, int64_t* first_word);
/* The kernels' noise function beside hiprand's own generator (BASELINE north_star: "sampling from a hiprand state per wavefront"): for
 * every counter quadruple (c0, c1, c2, c3) = (column block, cell id, step, stream | sample << 8) `ours` receives the four words the
 * kernels compute, `hiprand_words` the four words of ONE hiprand4() on a hiprandStatePhilox4_32_10_t set up by
 * hiprand_init(seed, subsequence = c2 | c3 << 32, offset = 4 * (c0 | c1 << 32)); c1 < 2^30 (the offset is 64 bits). */
int smx_k_hiprand(uint64_t seed, int32_t n, const uint32_t* counters, uint32_t* ours, uint32_t* hiprand_words);
/* Philox words / dropout multipliers / normals exactly as the kernels draw them. */
int smx_k_noise(uint64_t seed, int32_t stream, int32_t step, int32_t sample, const int64_t* cell_ids, int32_t B,
                int32_t width, float dropout_p, float* dropout_mult, float* normal);

#ifdef __cplusplus
}
#endif
#endif /* SISUA_HIP_H_ */
