"""Whole-step parity: forward, ELBO, every gradient, Adam update and BN moving
statistics of the HIP path against the float64 oracle on the same seeded inputs;
tolerance 1e-4 relative (BASELINE.json north_star)."""
import os

import numpy as np
import pytest

from oracle import sisua_oracle as so
from tests.util import adam_state_errors, grad_errors, make_pair, masked_move_error, perturbed_params, rel_l2, synth_counts, synth_labels

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.fixture(scope="module")
def Engine():
  from sisua_amd import build
  build.build(verbose=False)
  from sisua_amd.engine import Engine
  return Engine


CASES = {
    "vae_zinb": dict(model="vae", n_genes=203, likelihood="zinb", enc_units=(48, 40), dec_units=(40,), latent_dim=10),
    "vae_nb_nobn": dict(model="vae", n_genes=64, likelihood="nb", enc_units=(32,), dec_units=(32,), latent_dim=5,
                        batchnorm=False, input_dropout=0.3),
    "vae_zinbd": dict(model="vae", n_genes=130, likelihood="zinbd", enc_units=(64, 64), dec_units=(64, 64), latent_dim=12,
                      input_dropout=0.2),
    "vae_nbd": dict(model="vae", n_genes=97, likelihood="nbd", enc_units=(33,), dec_units=(17,), latent_dim=7),
    "dca_zinb": dict(model="dca", n_genes=150, likelihood="zinb", enc_units=(32,), dec_units=(32,), latent_dim=8),
    "dca_linear": dict(model="dca", n_genes=70, likelihood="nb", enc_units=(32,), dec_units=(32,), latent_dim=8,
                       latent_activation="linear"),
    "sisua": dict(model="sisua", n_genes=180, likelihood="zinb", enc_units=(64,), dec_units=(64,), latent_dim=9,
                  labels=((12, "nb"), (7, "onehot"))),
    "misa": dict(model="sisua", n_genes=140, likelihood="zinb", enc_units=(48,), dec_units=(48,), latent_dim=8,
                 labels=((12, "mixnb2"), (5, "mixnb3"))),
    # MISA with a continuous label variable (vae.py:86-92 'mixgaussian': mixture of normals per label dimension) beside a count one
    "misa_gauss": dict(model="sisua", n_genes=120, likelihood="nb", enc_units=(40,), dec_units=(40,), latent_dim=6,
                       labels=((9, "mixgauss3"), (6, "mixnb2")), alpha=10.0),
    # MISA's docstring example (vae.py:58 'mixtril'): ONE mixture of full-covariance Gaussians over the label vector (label_tril_kernel);
    # 14 = the protein panel of pbmc8k_ly, 38 > 32 = two padded plane widths
    "misa_tril": dict(model="sisua", n_genes=120, likelihood="zinb", enc_units=(40,), dec_units=(40,), latent_dim=6,
                      labels=((14, "mixtril2"), (5, "onehot")), alpha=10.0),
    "misa_tril_wide": dict(model="sisua", n_genes=90, likelihood="nb", enc_units=(32,), dec_units=(32,), latent_dim=5,
                           labels=((38, "mixtril3"),), alpha=4.0),
    "misa_tril_max": dict(model="sisua", n_genes=60, likelihood="nb", enc_units=(24,), dec_units=(24,), latent_dim=4,
                          labels=((64, "mixtril4"),), alpha=2.0),   # the largest head the kernel takes: four waves, 66.6 KB of LDS
    # MISA(zero_inflated=True) (vae.py:76-84): mixtures of ZERO-INFLATED negative binomials per label dimension
    "misa_zi": dict(model="sisua", n_genes=110, likelihood="zinb", enc_units=(40,), dec_units=(40,), latent_dim=6,
                    labels=((11, "mixzinb3"), (5, "mixnb2")), alpha=10.0),
    "scale": dict(model="scale", n_genes=150, likelihood="zinb", enc_units=(48,), dec_units=(48,), latent_dim=10, n_components=7),
    # SCALE with full-covariance components (scale.py:28,35 covariance='tril'): a lower-triangular factor per component
    "scale_tril": dict(model="scale", n_genes=130, likelihood="zinb", enc_units=(40,), dec_units=(40,), latent_dim=9, n_components=5, covariance="tril"),
    # SCALE read literally (scale.py:26,38-47): the latent POSTERIOR is the mixture-density layer, standard-normal prior, Monte-Carlo KL
    "scale_post": dict(model="scale", n_genes=120, likelihood="zinb", enc_units=(40,), dec_units=(40,), latent_dim=8, n_components=4, latent_mixture=True),
    "scale_post_nobn": dict(model="scale", n_genes=90, likelihood="nb", enc_units=(32, 24), dec_units=(24,), latent_dim=5, n_components=3, latent_mixture=True,
                            batchnorm=False),
    # the deterministic 'mse' output (RVmeta(dim, 'mse'), tests/test_singlecell_models.py:82-100 of the reference): one plane
    "dca_mse": dict(model="dca", n_genes=110, likelihood="mse", enc_units=(32,), dec_units=(32,), latent_dim=8),
    "vae_mse": dict(model="vae", n_genes=203, likelihood="mse", enc_units=(48,), dec_units=(40,), latent_dim=6),
    # SCALE with tied mixture parameters (scale.py:29-33): one scale vector for every component, weights fixed uniform; one location
    "scale_tied": dict(model="scale", n_genes=120, likelihood="zinb", enc_units=(40,), dec_units=(40,), latent_dim=7, n_components=5,
                       tie_scale=True, tie_mixtures=True),
    "scale_tied_loc": dict(model="scale", n_genes=90, likelihood="nb", enc_units=(32,), dec_units=(32,), latent_dim=6, n_components=4,
                           tie_loc=True),
    # ... with a clipnorm that binds on the tied tensors: the clip norm of C identical rows is the SHARED variable's (sum of squares / C)
    "scale_tied_clip": dict(model="scale", n_genes=90, likelihood="nb", enc_units=(32,), dec_units=(32,), latent_dim=6, n_components=4,
                            tie_loc=True, tie_scale=True, clipnorm=2.0),   # (stored norms 2.3 / 5.4+: loc is clipped only under the stored-tensor convention)
    "scalar": dict(model="scale", n_genes=130, likelihood="zinb", enc_units=(48,), dec_units=(48,), latent_dim=8, n_components=5,
                   labels=((10, "nb"), (4, "onehot"))),   # SCALAR = SCALE + SISUA's label heads (scale.py:52-59)
    "fvae": dict(model="fvae", n_genes=150, likelihood="zinb", enc_units=(48,), dec_units=(48,), latent_dim=10, disc_units=100,
                 disc_layers=3),
    "semifvae": dict(model="fvae", n_genes=120, likelihood="nb", enc_units=(40,), dec_units=(40,), latent_dim=7, disc_units=70,
                     disc_layers=2, labels=((6, "onehot"),), gamma=3.0),
    # round 5: outputs[1:] on FactorVAE / SemiFVAE (the label variable behind the observed outputs in the target order) and on the
    # mixture-density posterior (VERDICT r04 Missing 5: these refused)
    "fvae_two_outputs": dict(model="fvae", n_genes=130, likelihood="zinb", enc_units=(40,), dec_units=(40,), latent_dim=8, disc_units=64,
                             disc_layers=2, extra_outputs=((9, "nbd"),)),
    # round 6 (VERDICT r05 Missing 3): SemiFVAE with SEVERAL label variables -- one logit per class of every variable, the TC logit their
    # joint logsumexp, each variable's masked cross-entropy under the softmax of its own logits (smx_factor.hip: disc_head_kernel)
    "semifvae_three_labels": dict(model="fvae", n_genes=120, likelihood="zinb", enc_units=(40,), dec_units=(40,), latent_dim=7, disc_units=70,
                                  disc_layers=2, labels=((6, "onehot"), (2, "onehot"), (9, "onehot")), gamma=3.0, alpha=4.0),
    "semifvae_two_labels_two_outputs": dict(model="fvae", n_genes=110, likelihood="nb", enc_units=(40,), dec_units=(40,), latent_dim=7, disc_units=70,
                                            disc_layers=2, extra_outputs=((8, "nbd"),), labels=((16, "onehot"), (16, "onehot")), gamma=2.0, alpha=5.0),
    "semifvae_two_outputs": dict(model="fvae", n_genes=110, likelihood="nb", enc_units=(40,), dec_units=(40,), latent_dim=7, disc_units=70,
                                 disc_layers=2, extra_outputs=((8, "zinb"), (5, "nb")), labels=((6, "onehot"),), gamma=3.0, alpha=5.0),
    "scale_post_two_outputs": dict(model="scale", n_genes=120, likelihood="zinb", enc_units=(40,), dec_units=(40,), latent_dim=8, n_components=4,
                                   latent_mixture=True, extra_outputs=((10, "nbd"),)),
    # a discriminator as deep as odin's default (1000 units: here 640 and 600, the latter padded to 608): its layers' forward
    # products and input gradients take the bf16 x 3 form of smx_dgemm.hip (K >= 512); the square weight gradients the panel form
    # of smx_panel.h in column groups of 128 (640 = 5 groups) or, when the width is no multiple of 128, the 32 x 32-tile kernel
    "fvae_deep_disc": dict(model="fvae", n_genes=96, likelihood="nb", enc_units=(32,), dec_units=(32,), latent_dim=6, disc_units=640,
                           disc_layers=3),
    "fvae_deep_disc_608": dict(model="fvae", n_genes=96, likelihood="nb", enc_units=(32,), dec_units=(32,), latent_dim=6, disc_units=600,
                               disc_layers=2),
    "scvi_zinbd": dict(model="scvi", n_genes=160, likelihood="zinbd", enc_units=(48,), dec_units=(48,), latent_dim=6,
                       encl_units=(16,)),
    "scvi_nbd": dict(model="scvi", n_genes=96, likelihood="nbd", enc_units=(32,), dec_units=(32,), latent_dim=4,
                     encl_units=(8,), batchnorm=False),
    # scvi's own defaults (scvi.py:33-48: encoder [64, 64], library encoder [64], latent 10) and a two-layer library encoder:
    # the side-by-side first layers continue into deeper ones, gradient fronts are handed down layer by layer
    "scvi_default": dict(model="scvi", n_genes=200, likelihood="zinbd", enc_units=(64, 64), dec_units=(64, 64), latent_dim=10,
                         encl_units=(64,)),
    "scvi_deep": dict(model="scvi", n_genes=120, likelihood="nbd", enc_units=(64, 40), dec_units=(48,), latent_dim=8,
                      encl_units=(32, 24)),
    "paper_shape": dict(model="vae", n_genes=1998, likelihood="zinb", enc_units=(128,), dec_units=(128,), latent_dim=32),
    # wide panels (>= 4096 padded genes): with flag bf16x3 the products that contract over the gene axis run as one workgroup
    # per K slice + a reduce launch (smx_bigk.hip), the products that contract over the minibatch (d W of the head, the first
    # encoder layer's weight gradient) as one workgroup per 32 genes with every tile of H (smx_panel.h); without BatchNorm the
    # layer has a bias, whose gradient is the panel form's column sum of the narrow operand; 96 units: three tiles of H
    "wide_panel_nobn_96": dict(model="vae", n_genes=4128, likelihood="nb", enc_units=(96,), dec_units=(96,), latent_dim=8, batchnorm=False),
    "wide_panel_64": dict(model="vae", n_genes=4200, likelihood="zinb", enc_units=(64,), dec_units=(64,), latent_dim=10),
    "wide_panel_128": dict(model="vae", n_genes=4500, likelihood="nb", enc_units=(128,), dec_units=(128,), latent_dim=16),
    # outputs[1:] of the reference's constructors (tests/test_singlecell_models.py:129-141: VAE(outputs=[zinb genes, nbd proteins]);
    # scvi.py:168-169): further OBSERVED output variables as heads on the decoder output -- weight 1, every cell, metric nllk_o
    "vae_two_outputs": dict(model="vae", n_genes=180, likelihood="zinb", enc_units=(64,), dec_units=(64,), latent_dim=9, extra_outputs=((12, "nbd"),)),
    "vae_three_outputs": dict(model="vae", n_genes=140, likelihood="nb", enc_units=(48,), dec_units=(48,), latent_dim=8,
                              extra_outputs=((14, "zinb"), (38, "zinbd"))),
    "dca_two_outputs": dict(model="dca", n_genes=120, likelihood="zinb", enc_units=(40,), dec_units=(40,), latent_dim=6, extra_outputs=((7, "onehot"),)),
    "sisua_extra_output": dict(model="sisua", n_genes=150, likelihood="zinb", enc_units=(48,), dec_units=(48,), latent_dim=8,
                               extra_outputs=((10, "nb"),), labels=((12, "nbd"), (5, "onehot")), alpha=10.0),
    "scale_two_outputs": dict(model="scale", n_genes=130, likelihood="zinb", enc_units=(48,), dec_units=(48,), latent_dim=8, n_components=5,
                              extra_outputs=((9, "nbd"),)),
    "scvi_two_outputs": dict(model="scvi", n_genes=160, likelihood="zinbd", enc_units=(48,), dec_units=(48,), latent_dim=6, encl_units=(16,),
                             extra_outputs=((12, "nbd"),)),
    # scvi.py:55-56,66-86: dispersion / inflation without a Dense head -- one trainable per-gene vector (out1/b, out2/b alone)
    "scvi_share_dispersion": dict(model="scvi", n_genes=160, likelihood="zinbd", enc_units=(48,), dec_units=(48,), latent_dim=6, encl_units=(16,),
                                  dispersion="share"),
    "scvi_share_both": dict(model="scvi", n_genes=130, likelihood="zinbd", enc_units=(40,), dec_units=(40,), latent_dim=5, encl_units=(16,),
                            dispersion="share", inflation="share"),
    "scvi_single": dict(model="scvi", n_genes=130, likelihood="zinbd", enc_units=(40,), dec_units=(40,), latent_dim=5, encl_units=(16,),
                        dispersion="single", inflation="single"),
    "scvi_nbd_share_dispersion": dict(model="scvi", n_genes=96, likelihood="nbd", enc_units=(32,), dec_units=(32,), latent_dim=4, encl_units=(8,),
                                      dispersion="share", batchnorm=False),
}


def _problem(kw, n=300, seed=0):
  spec, cfg = make_pair(**kw)
  x = synth_counts(n, spec.n_genes, sparsity=0.85, seed=seed, max_count=2000 if spec.n_genes < 500 else None)
  ys = synth_labels(n, spec.extra_outputs + spec.labels)   # target arrays in head order: outputs[1:], then the label variables
  _, lm, lv = so.library_size(x)
  lib = np.tile(np.array([[lm, lv]], dtype=np.float32), (n, 1))
  mask = so.label_mask(n, 0.4, n_omics=1 + len(spec.labels), seed=1)
  return spec, cfg, x, ys, lib, mask


def _oracle_step(spec, params, bn, opt, x, ys, lib, mask, rows, step, cell_base=0):
  noise = so.PhiloxNoise(spec.seed, step, rows + cell_base)
  return so.train_step(spec, params, bn, opt, x[rows], noise, y=[y[rows] for y in ys], library=lib[rows], mask=mask[rows])


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("batch", [32, 100, 160])
def test_one_step_matches_oracle(Engine, name, batch):
  kw = CASES[name]
  spec, cfg, x, ys, lib, mask = _problem(kw)
  params = perturbed_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  e = Engine(cfg, max_batch=max(128, batch), init=False)
  e.set_params(params)
  e.upload(x, ys, lib, mask, cell_id_base=1000)
  rows = np.random.default_rng(1).choice(x.shape[0], size=batch, replace=False).astype(np.int32)
  p0 = {k: v.copy() for k, v in params.items()}
  res = _oracle_step(spec, params, bn, opt, x, ys, lib, mask, rows, 0, cell_base=1000)
  m = e.train_step(rows)
  assert m["nan_flag"] == 0 and m["step"] == 1
  for key in ("loss", "nllk_x", "kl"):
    assert np.isclose(m[key], res["metrics"][key], rtol=RTOL, atol=1e-5), (key, m[key], res["metrics"][key])
  if spec.labels:
    assert np.isclose(m["nllk_y"], res["metrics"]["nllk_y"], rtol=RTOL, atol=1e-5)
  assert np.isclose(m["nllk_o"], res["metrics"]["nllk_o"], rtol=RTOL, atol=1e-5) and (m["nllk_o"] != 0) == bool(spec.extra_outputs)
  if spec.model == "scvi":
    assert np.isclose(m["kl_l"], res["metrics"]["kl_l"], rtol=RTOL, atol=1e-5)
    assert ("out1/W" in e.names) == (spec.dispersion == "full") and ("out2/W" in e.names) == (spec.k == 3 and spec.inflation == "full")
  if spec.model == "fvae":
    for key in ("tc", "dtc_loss"):
      assert np.isclose(m[key], res["metrics"][key], rtol=RTOL, atol=1e-5), (key, m[key], res["metrics"][key])
  grads = e.get_params(which=1)
  worst = grad_errors(grads, res["grads"])
  assert max(worst.values()) < RTOL, sorted(worst.items(), key=lambda kv: -kv[1])[:3]
  # the optimiser: its moments are linear / quadratic in the (clipped) gradients and are held to the gradients' tolerance;
  # the weights are judged where |g| is far above float32 rounding (first Adam step: dw = lr g / (|g| + 3.2e-6), so rounding
  # noise dg on a near-zero gradient moves a weight by up to lr |dg| / 3.2e-6 -- bounded by lr, checked as a bound only)
  em, ev, where = adam_state_errors(e, opt)
  assert em < 2e-4 and ev < 4e-4, (em, ev, where)
  newp = e.get_params()
  top = max(np.linalg.norm(v) for v in res["grads"].values())
  for k in newp:
    assert np.abs(newp[k] - params[k]).max() <= 1.001 * spec.lr, k
    if np.linalg.norm(res["grads"][k]) > 1e-3 * top:  # skip analytically-zero gradients
      err = masked_move_error(newp[k], p0[k], params[k], res["grads"][k], spec.lr)
      assert err is None or err < 2e-3, (k, err)
  names = [p for p, _ in so.bn_manifest(spec)]
  for i, st in e.get_bn().items():
    assert np.allclose(st["moving_mean"], bn[f"{names[i]}/moving_mean"], rtol=1e-4, atol=1e-6)
    assert np.allclose(st["moving_var"], bn[f"{names[i]}/moving_var"], rtol=1e-4, atol=1e-6)
  e.close()


@pytest.mark.parametrize("flags", [("head_loss",), ("front",), ("bwd_front",), ("head_bwd",), ("wgrad",),
                                   ("head_loss", "front", "bwd_front", "head_bwd", "wgrad")])
@pytest.mark.parametrize("name", ["vae_zinb", "vae_zinbd", "sisua", "paper_shape", "dca_zinb"])
def test_separate_launch_forms_match_oracle(Engine, name, flags):
  """Every stage that has a fused / wide default form also keeps its separate-launch form (smx_set_flag): eval and
  the scoring paths use those, and shapes the fused kernels do not take fall back to them.  Same parity bar, and
  three steps of both forms end within rounding of each other."""
  kw = CASES[name]
  spec, cfg, x, ys, lib, mask = _problem(kw)
  params = perturbed_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  B = 96
  e, e0 = Engine(cfg, max_batch=128, init=False), Engine(cfg, max_batch=128, init=False)
  for eng in (e, e0):
    eng.set_params(params)
    eng.upload(x, ys, lib, mask, cell_id_base=1000)
  for f in flags:
    e.set_flag(f, False)
  rows = np.random.default_rng(1).choice(x.shape[0], size=B, replace=False).astype(np.int32)
  res = _oracle_step(spec, params, bn, opt, x, ys, lib, mask, rows, 0, cell_base=1000)
  m = e.train_step(rows)
  for key in ("loss", "nllk_x", "kl"):
    assert np.isclose(m[key], res["metrics"][key], rtol=RTOL, atol=1e-5), (key, m[key], res["metrics"][key])
  worst = grad_errors(e.get_params(which=1), res["grads"])
  assert max(worst.values()) < RTOL, sorted(worst.items(), key=lambda kv: -kv[1])[:3]
  assert np.isclose(m["grad_norm_max"], max(np.linalg.norm(g) for g in res["grads"].values()), rtol=1e-4)
  m0 = e0.train_step(rows)
  for s in (1, 2):
    r2 = ((rows + 7 * s) % x.shape[0]).astype(np.int32)
    m, m0 = e.train_step(r2), e0.train_step(r2)
  assert np.isclose(m["loss"], m0["loss"], rtol=1e-5)
  with pytest.raises(Exception):
    e.set_flag("no_such_flag", True)
  e.close(); e0.close()


@pytest.mark.parametrize("name", ["fvae", "semifvae", "semifvae_three_labels", "semifvae_two_labels_two_outputs"])
def test_activation_epilogue_forms_match_oracle(Engine, name):
  """Layers without BatchNorm and dropout (the FactorVAE discriminator): bias + activation in the products' store paths
  and the activation's derivative in the d-input products, or the separate bias / activation launches (flag
  act_epilogue): both match the oracle, three steps of both end within rounding of each other."""
  kw = CASES[name]
  spec, cfg, x, ys, lib, mask = _problem(kw)
  params = perturbed_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  B = 96
  e, e0 = Engine(cfg, max_batch=128, init=False), Engine(cfg, max_batch=128, init=False)
  for eng in (e, e0):
    eng.set_params(params)
    eng.upload(x, ys, lib, mask, cell_id_base=1000)
  e.set_flag("act_epilogue", False)
  rows = np.random.default_rng(1).choice(x.shape[0], size=B, replace=False).astype(np.int32)
  res = _oracle_step(spec, params, bn, opt, x, ys, lib, mask, rows, 0, cell_base=1000)
  for eng in (e, e0):
    m = eng.train_step(rows)
    for key in ("loss", "nllk_x", "kl", "tc", "dtc_loss"):
      assert np.isclose(m[key], res["metrics"][key], rtol=RTOL, atol=1e-5), (key, m[key], res["metrics"][key])
    worst = grad_errors(eng.get_params(which=1), res["grads"])
    assert max(worst.values()) < RTOL, sorted(worst.items(), key=lambda kv: -kv[1])[:3]
  for s in (1, 2):
    r2 = ((rows + 7 * s) % x.shape[0]).astype(np.int32)
    m, m0 = e.train_step(r2), e0.train_step(r2)
  assert np.isclose(m["loss"], m0["loss"], rtol=1e-5) and np.isclose(m["dtc_loss"], m0["dtc_loss"], rtol=1e-4)
  e.close(); e0.close()


@pytest.mark.parametrize("flags", [("label_ride",), ("label_ride", "wgrad"), ("wgrad",), ("head_bwd",), ("bwd_front", "wgrad")])
@pytest.mark.parametrize("name", ["sisua", "misa", "misa_gauss"])
def test_label_backward_forms_match_oracle(Engine, name, flags):
  """Label heads' backward: d d as extra slabs of the output head's backward launch and the head's weight gradient in
  the grouped launch at the end of the backward pass (its optimiser chunks then wait for the last launch), or the
  grouped launch of their own: same parity bar, and the optimiser saw every tensor's final gradient in both forms."""
  kw = CASES[name]
  spec, cfg, x, ys, lib, mask = _problem(kw)
  params = perturbed_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  B = 96
  e, e0 = Engine(cfg, max_batch=128, init=False), Engine(cfg, max_batch=128, init=False)
  for eng in (e, e0):
    eng.set_params(params)
    eng.upload(x, ys, lib, mask, cell_id_base=1000)
  for f in flags:
    e.set_flag(f, False)
  rows = np.random.default_rng(1).choice(x.shape[0], size=B, replace=False).astype(np.int32)
  p0 = {k: v.copy() for k, v in params.items()}
  res = _oracle_step(spec, params, bn, opt, x, ys, lib, mask, rows, 0, cell_base=1000)   # (updates params in place)
  for eng in (e, e0):
    m = eng.train_step(rows)
    for key in ("loss", "nllk_x", "nllk_y", "kl"):
      assert np.isclose(m[key], res["metrics"][key], rtol=RTOL, atol=1e-5), (key, m[key], res["metrics"][key])
    worst = grad_errors(eng.get_params(which=1), res["grads"])
    assert max(worst.values()) < RTOL, sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    em, ev, where = adam_state_errors(eng, opt)
    assert em < 2e-4 and ev < 4e-4, (em, ev, where)
    got = eng.get_params()
    for k in got:
      assert np.abs(got[k] - params[k]).max() <= 1.001 * spec.lr, k
      if k.startswith("lab"):
        err = masked_move_error(got[k], p0[k], params[k], res["grads"][k], spec.lr)
        assert err is None or err < 2e-3, (k, err)
  for s in (1, 2):
    r2 = ((rows + 7 * s) % x.shape[0]).astype(np.int32)
    m, m0 = e.train_step(r2), e0.train_step(r2)
  assert np.isclose(m["loss"], m0["loss"], rtol=1e-5)
  e.close(); e0.close()


@pytest.mark.parametrize("flags", [("scvi_fused",), ("twin",), ("scvi_fused", "twin"), ("scvi_fused", "twin", "bwd_front", "front", "wgrad")])
@pytest.mark.parametrize("name", ["scvi_zinbd", "scvi_nbd", "scvi_default", "scvi_deep"])
def test_scvi_separate_launch_forms_match_oracle(Engine, name, flags):
  """scvi: the row-local head launch of a training step (library latent + softmax head + likelihood + their backward,
  smx_scvi.hip) and the side-by-side launches of the two encoders / pairs of heads keep their separate-launch forms
  (what evaluation, prediction and scoring use): same parity bar, and three steps of both forms end within rounding."""
  kw = CASES[name]
  spec, cfg, x, ys, lib, mask = _problem(kw)
  params = perturbed_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  B = 96
  e, e0 = Engine(cfg, max_batch=128, init=False), Engine(cfg, max_batch=128, init=False)
  for eng in (e, e0):
    eng.set_params(params)
    eng.upload(x, ys, lib, mask, cell_id_base=1000)
  for f in flags:
    e.set_flag(f, False)
  rows = np.random.default_rng(1).choice(x.shape[0], size=B, replace=False).astype(np.int32)
  res = _oracle_step(spec, params, bn, opt, x, ys, lib, mask, rows, 0, cell_base=1000)
  for eng in (e, e0):
    m = eng.train_step(rows)
    for key in ("loss", "nllk_x", "kl", "kl_l"):
      assert np.isclose(m[key], res["metrics"][key], rtol=RTOL, atol=1e-5), (key, m[key], res["metrics"][key])
    worst = grad_errors(eng.get_params(which=1), res["grads"])
    assert max(worst.values()) < RTOL, sorted(worst.items(), key=lambda kv: -kv[1])[:3]
  for s in (1, 2):
    r2 = ((rows + 7 * s) % x.shape[0]).astype(np.int32)
    m, m0 = e.train_step(r2), e0.train_step(r2)
  assert np.isclose(m["loss"], m0["loss"], rtol=1e-5)
  e.close(); e0.close()


@pytest.mark.parametrize("name", ["vae_zinb", "sisua", "scvi_zinbd", "fvae", "semifvae"])
def test_injected_noise_matches_oracle(Engine, name):
  """smx_set_noise hook: same parameters + same minibatch + same injected eps / dropout
  masks => same loss and gradients (SURVEY.md section 7 'hard parts')."""
  kw = dict(CASES[name], input_dropout=0.25)
  spec, cfg, x, ys, lib, mask = _problem(kw)
  params = perturbed_params(spec)
  B = 64
  rows = np.arange(10, 10 + B, dtype=np.int32)
  rng = np.random.default_rng(9)
  def dmask(w, p):
    return (rng.uniform(size=(B, w)) >= p).astype(np.float32) / np.float32(1 - p)
  drop = {so.STREAM_INPUT_DROPOUT: dmask(spec.n_genes, 0.25)}
  for i, u in enumerate(spec.enc_units):
    drop[so.STREAM_ENC_DROPOUT + i] = dmask(u, spec.dropout_enc)
  for i, u in enumerate(spec.dec_units):
    drop[so.STREAM_DEC_DROPOUT + i] = dmask(u, spec.dropout_dec)
  normal = {so.STREAM_EPS_Z: rng.normal(size=(B, spec.latent_dim)).astype(np.float32)}
  if spec.model == "scvi":
    for i, u in enumerate(spec.encl_units):
      drop[so.STREAM_ENCL_DROPOUT + i] = dmask(u, spec.dropout_enc)
    normal[so.STREAM_EPS_L] = rng.normal(size=(B, 1)).astype(np.float32)
  uniform = {}
  if spec.model == "fvae":   # the permute_dims permutations are the ranks of these (one tie on purpose: broken by row)
    u = rng.uniform(size=(B, spec.latent_dim)).astype(np.float32)
    u[7, 0] = u[3, 0]
    uniform[so.STREAM_PERMUTE] = u
  e = Engine(cfg, max_batch=B, init=False)
  e.set_params(params)
  e.upload(x, ys, lib, mask)
  for s, v in {**drop, **normal, **uniform}.items():
    e.set_noise(s, v)
  res = so.forward_backward(spec, params, so.init_bn_state(spec), x[rows], so.InjectedNoise(drop, normal, uniform),
                            y=[y[rows] for y in ys], library=lib[rows], mask=mask[rows])
  m = e.train_step(rows)
  assert np.isclose(m["loss"], res["loss"], rtol=RTOL)
  grads = e.get_params(which=1)
  worst = grad_errors(grads, res["grads"])
  assert max(worst.values()) < RTOL, sorted(worst.items(), key=lambda kv: -kv[1])[:3]
  e.close()


@pytest.mark.parametrize("name,graph", [("vae_zinb", False), ("vae_zinb", True), ("sisua", True), ("scvi_zinbd", False), ("scvi_nbd", True), ("misa", False), ("scale", True),
                                        ("scale_tril", False), ("scale_post", True), ("misa_tril", False), ("misa_zi", True), ("fvae", True), ("semifvae", False)])
def test_trajectory_matches_oracle(Engine, name, graph):
  """50-step seeded trajectory: ELBO per step within 1e-4 relative."""
  kw = CASES[name]
  spec, cfg, x, ys, lib, mask = _problem(kw, n=512)
  params = {k: v.copy() for k, v in so.init_params(spec).items()}
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  B, steps = 64, 50
  e = Engine(cfg, max_batch=B, init=False)
  e.set_params(params)
  e.upload(x, ys, lib, mask)
  order = np.concatenate([so.epoch_order(x.shape[0], ep, shuffle=100, seed=1) for ep in range(8)])[: steps * B].astype(np.int32)
  ref, got = [], []
  for s in range(steps):
    rows = order[s * B:(s + 1) * B]
    ref.append(_oracle_step(spec, params, bn, opt, x, ys, lib, mask, rows, s)["loss"])
    got.append(e.train_step(rows, graph=graph)["loss"])
  ref, got = np.array(ref), np.array(got)
  assert np.allclose(got, ref, rtol=RTOL), np.abs(got / ref - 1).max()
  assert ref[-5:].mean() < ref[:5].mean()      # and it trains
  e.close()


def test_graph_and_eager_are_bitwise_identical(Engine):
  kw = CASES["vae_zinb"]
  spec, cfg, x, ys, lib, mask = _problem(kw)
  outs = []
  for graph in (False, True):
    e = Engine(cfg, max_batch=64)
    e.upload(x, ys, lib, mask)
    order = np.arange(64 * 6, dtype=np.int32) % x.shape[0]
    e.train_steps(order, 6, 64, graph=graph)
    outs.append(e.get_params())
    e.close()
  for k in outs[0]:
    assert np.array_equal(outs[0][k], outs[1][k]), k


def test_staged_row_ids_equal_passed_row_ids(Engine):
  """smx_train_stage: the row ids of the next train_steps call uploaded ahead of it (bench.py stages the timed steps' ids before
  the clock starts).  Bit-identical to passing them in the call; staged ids serve exactly one call of the same shape."""
  from sisua_amd._hip import SmxError
  spec, cfg, x, ys, lib, mask = _problem(CASES["vae_zinb"], n=256)
  params = perturbed_params(spec)
  order = (np.random.default_rng(3).permutation(256)[: 3 * 64].tolist() + list(range(2 * 64)))
  order = np.asarray(order, np.int32)
  got = []
  for staged in (False, True):
    e = Engine(cfg, max_batch=64, init=False)
    e.set_params(params)
    e.upload(x, ys, lib, mask)
    if staged:
      with pytest.raises(SmxError):
        e.train_steps(None, 5, 64, graph=False)        # nothing staged
      e.stage_steps(order, 5, 64)
      with pytest.raises(SmxError):
        e.train_steps(None, 4, 64, graph=False)        # another shape (and the staged ids are spent by the attempt)
      e.stage_steps(order, 5, 64)
      e.train_steps(None, 5, 64, graph=False)
      with pytest.raises(SmxError):
        e.train_steps(None, 5, 64, graph=False)        # one call only
    else:
      e.train_steps(order, 5, 64, graph=False)
    h = e.metrics_history(5)
    got.append((h["loss"].copy(), e.get_params()))
    e.close()
  assert np.array_equal(got[0][0], got[1][0])
  for k in got[0][1]:
    assert np.array_equal(got[0][1][k], got[1][1][k]), k


@pytest.mark.parametrize("case", ["sisua", "scale_post", "sisua_extra_output", "scvi_two_outputs", "scvi_share_both", "scvi_single"])
def test_eval_and_forward_match_oracle(Engine, case):
  kw = CASES[case]   # ('scale_post': z_mean / z_scale are the mixture posterior's moments; a different draw may pick another component)
  spec, cfg, x, ys, lib, mask = _problem(kw)
  params = perturbed_params(spec)
  bn = so.init_bn_state(spec)
  rng = np.random.default_rng(2)
  for k in bn:  # non-trivial moving statistics
    bn[k] = (bn[k] + 0.2 * rng.uniform(size=bn[k].shape)).astype(np.float32).astype(np.float64)
  e = Engine(cfg, max_batch=128, init=False)
  e.set_params(params)
  names = [p for p, _ in so.bn_manifest(spec)]
  e.set_bn({i: dict(moving_mean=bn[f"{n}/moving_mean"], moving_var=bn[f"{n}/moving_var"]) for i, n in enumerate(names)})
  e.upload(x, ys, lib, mask)
  rows = np.arange(40, 140, dtype=np.int32)
  noise = so.PhiloxNoise(spec.seed, 0, rows, sample=0)
  res = so.forward_backward(spec, params, bn, x[rows], noise, y=[y[rows] for y in ys], library=lib[rows],
                            mask=mask[rows], training=False, backward=False)
  m = e.eval_step(rows)
  assert np.isclose(m["loss"], res["loss"], rtol=RTOL)
  out = e.forward(row_ids=rows, sample_index=0)
  assert np.allclose(out["z_mean"], res["z_mean"], rtol=1e-4, atol=1e-5)
  assert np.allclose(out["z_scale"], res["z_scale"], rtol=1e-4, atol=1e-5)
  assert np.allclose(out["z_sample"], res["z"], rtol=1e-3, atol=1e-4)
  for c in range(spec.k):
    assert np.allclose(out["x_params"][c], res["x_params"][c], rtol=1e-3, atol=1e-4)
  assert len(out["y_params"]) == len(spec.extra_outputs + spec.labels)
  for j in range(len(spec.extra_outputs + spec.labels)):
    assert np.allclose(out["y_params"][j], res["y_params"][j], rtol=1e-3, atol=1e-4)
  # host-batch path (predict on raw arrays) agrees with the resident-row path on the means
  out2 = e.forward(x=x[rows], library=lib[rows])
  assert np.allclose(out2["z_mean"], out["z_mean"], rtol=1e-6, atol=1e-6)
  # a different MC sample index changes the draw, not the mean
  out3 = e.forward(row_ids=rows, sample_index=1)
  assert np.array_equal(out3["z_mean"], out["z_mean"]) and not np.allclose(out3["z_sample"], out["z_sample"])
  e.close()


def test_error_paths(Engine):
  from sisua_amd import SmxError
  spec, cfg, x, ys, lib, mask = _problem(CASES["vae_zinb"], n=50)
  e = Engine(cfg, max_batch=16)
  with pytest.raises(SmxError):
    e.train_step(np.arange(8, dtype=np.int32))            # no dataset yet
  e.upload(x, ys, lib, mask)
  with pytest.raises(SmxError):
    e.train_step(np.arange(32, dtype=np.int32))           # batch > max_batch
  with pytest.raises(SmxError):
    e.train_step(np.array([0, 1, 50], dtype=np.int32))    # row id out of range
  assert e.train_step(np.arange(16, dtype=np.int32))["step"] == 1
  e.close()


@pytest.mark.parametrize("graph,buckets", [(False, "1"), (False, "2"), (True, "1"), (False, "shard")])
def test_rccl_single_rank_allreduce_is_identity(Engine, monkeypatch, graph, buckets):
  """The data-parallel code path exercised on the one GPU of the test box (1-rank RCCL communicator,
  SMX_FORCE_ALLREDUCE): eager = two buckets on the communication stream overlapped with backward,
  graph = one captured all-reduce.  A 1-rank all-reduce must not change any result.  `shard`: flag opt_shard -- ncclReduceScatter, the sharded
  clip + Adam over the (one) slice, ncclAllGather, smx_opt_gather -- through RCCL itself."""
  from sisua_amd._hip import SmxError
  shard = buckets == "shard"
  monkeypatch.setenv("SMX_FORCE_ALLREDUCE", "1")
  monkeypatch.setenv("SMX_DP_BUCKETS", "2" if shard else buckets)
  # the data-parallel path takes the norms from a pass over the (all-reduced) gradient; give the reference
  # run the same summation order so that the comparison can be exact
  from sisua_amd import _hip
  _hip.set_tuning("no_sq_partials", 1)   # (cleared after the test: tests/conftest.py)
  spec, cfg, x, ys, lib, mask = _problem(CASES["sisua"])
  outs = []
  for use_comm in (False, True):
    e = Engine(cfg, max_batch=64)
    e.upload(x, ys, lib, mask)
    if use_comm:
      e.comm_init(0, 1, Engine.comm_unique_id())
      assert e.world == 1
      if shard:
        e.set_flag("opt_shard", True)
    order = np.arange(64 * 6, dtype=np.int32) % x.shape[0]
    m = e.train_steps(order, 6, 64, graph=graph, metrics=True)
    if use_comm and shard:
      with pytest.raises(SmxError, match="opt_gather"):
        e.get_params(which=2)
      e.opt_gather()
    outs.append((m, e.get_params(), e.get_bn(), e.get_params(which=2), e.get_params(which=3)))
    if use_comm:   # the measurement hook of bench.py (N > 1): the collective alone; it must leave the model's state as it is
      us, nbytes = e.comm_time_allreduce(20)
      assert 0.0 < us < 1e4 and nbytes >= 4 * sum(v.size for v in outs[-1][1].values())
      assert all(np.array_equal(v, outs[-1][1][k]) for k, v in e.get_params().items())
      m2 = e.train_steps(order, 6, 64, graph=graph, metrics=True)
      assert np.isfinite(m2["loss"])
    else:
      with pytest.raises(Exception, match="communicator"):
        e.comm_time_allreduce(5)
    e.close()
  for key in ("loss", "nllk_x", "nllk_y", "kl"):
    assert outs[0][0][key] == outs[1][0][key], key
  for which in (1, 3, 4):
    for k in outs[0][which]:
      assert np.array_equal(outs[0][which][k], outs[1][which][k]), (which, k)
  for i in outs[0][2]:
    assert np.array_equal(outs[0][2][i]["moving_mean"], outs[1][2][i]["moving_mean"])


def _golden(name):
  import os
  return np.load(os.path.join(os.path.dirname(__file__), "golden", name))


def test_hip_matches_committed_step_fixture(Engine):
  """one full VAE step with fixed params / eps / masks against COMMITTED numbers."""
  from tests.golden import make_oracle_fixtures as mk
  fx = _golden("oracle_step_fixture.npz")
  spec, cfg = make_pair(**mk.STEP_KW)
  names = [n for n, _ in so.manifest(spec)]
  e = Engine(cfg, max_batch=8, init=False)
  e.set_params({n: fx[f"p0/{n}"] for n in names})
  e.upload(fx["x"])
  for k in fx.files:
    if k.startswith("drop/") or k.startswith("eps/"):
      e.set_noise(int(k.split("/")[1]), fx[k])
  m = e.train_step(np.arange(8, dtype=np.int32))
  assert np.isclose(m["loss"], float(fx["loss"]), rtol=RTOL) and np.isclose(m["kl"], float(fx["kl"]), rtol=RTOL)
  worst = grad_errors(e.get_params(which=1), {n: fx[f"g/{n}"] for n in names})
  assert max(worst.values()) < RTOL, sorted(worst.items(), key=lambda kv: -kv[1])[:3]
  # the committed post-update weights: the first Adam step is lr_1 g / (|g| + eps sqrt(1 - b2)/(1 - b1)...) -- its moments
  # follow from the committed gradients exactly (m = 0.1 g, v = 0.001 g^2: no clipping at these norms), so they are held to
  # the gradients' tolerance; the weights where |g| is above rounding, and as a bound elsewhere
  g_fx = {n: np.asarray(fx[f"g/{n}"], np.float64) for n in names}
  em, ev, where = adam_state_errors(e, {"m": {n: 0.1 * g for n, g in g_fx.items()}, "v": {n: 0.001 * g * g for n, g in g_fx.items()}})
  assert em < 2e-4 and ev < 4e-4, (em, ev, where)
  newp = e.get_params()
  for n in names:
    assert np.abs(newp[n] - fx[f"p1/{n}"]).max() <= 1.001 * spec.lr, n
    err = masked_move_error(newp[n], fx[f"p0/{n}"], fx[f"p1/{n}"], g_fx[n], spec.lr)
    assert err is None or err < 2e-3, (n, err)
  e.close()


def test_hip_matches_committed_trajectory_fixture(Engine):
  """50-step seeded trajectory, ELBO per step within 1e-4 of the COMMITTED values."""
  from tests.golden import make_oracle_fixtures as mk
  fx = _golden("oracle_trajectory_fixture.npz")
  spec, cfg = make_pair(**mk.TRAJ_KW)
  e = Engine(cfg, max_batch=64)
  e.upload(fx["x"])
  got = [e.train_step(fx["order"][s * 64:(s + 1) * 64])["loss"] for s in range(50)]
  assert np.allclose(got, fx["loss"], rtol=RTOL), np.abs(np.array(got) / fx["loss"] - 1).max()
  assert rel_l2(e.get_params()["lat/W"], fx["final_lat_W"]) < RTOL   # the latent head's weights after the 50 steps
  e.close()



def test_scoring_head_images_follow_the_parameters(Engine):
  """The scoring head's bf16 x 3 images of W_out are kept while the parameters stand (a scoring sweep over a dataset splits W once) and rebuilt
  when they move: after optimiser steps, and after set_params, marginal_llk equals -- bit for bit -- what the same engine gives with the images
  rebuilt on every call (knob no_wimg_cache)."""
  from sisua_amd import _hip
  spec, cfg, x, ys, lib, mask = _problem(dict(CASES["vae_zinb"], labels=()))
  e = Engine(cfg, max_batch=64)
  e.upload(x, ys, lib, mask)
  rows = np.arange(10, 58, dtype=np.int32)

  def both():
    a = e.marginal_llk(row_ids=rows, n_samples=8)
    b = e.marginal_llk(row_ids=rows, n_samples=8)   # (a second call: served from the kept images)
    _hip.set_tuning("no_wimg_cache", 1)
    try:
      c = e.marginal_llk(row_ids=rows, n_samples=8)
    finally:
      _hip.set_tuning("no_wimg_cache", 0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
    return a[0]
  m0 = both()
  e.train_steps(np.arange(64 * 3, dtype=np.int32) % x.shape[0], 3, 64)
  m1 = both()
  assert not np.array_equal(m0, m1)
  e.set_params(perturbed_params(spec))
  m2 = both()
  assert not np.array_equal(m1, m2)
  e.close()


@pytest.mark.parametrize("name", ["vae_zinb", "scvi_zinbd", "dca_zinb", "scale", "scale_tril", "scale_post"])
def test_marginal_llk_matches_oracle(Engine, name):
  """importance-weighted log p(x) (posterior.py:941-976) on the GPU vs the oracle."""
  spec, cfg, x, ys, lib, mask = _problem(dict(CASES[name], labels=()) if name != "sisua" else CASES[name])
  params = perturbed_params(spec)
  bn = so.init_bn_state(spec)
  e = Engine(cfg, max_batch=64, init=False)
  e.set_params(params)
  e.upload(x, ys, lib, mask, cell_id_base=500)
  rows = np.arange(20, 70, dtype=np.int32)
  S = 12
  ref_m, ref_l = so.marginal_log_prob(spec, params, bn, x[rows], rows + 500, S, library=lib[rows])
  got_m, got_l = e.marginal_llk(row_ids=rows, n_samples=S)
  assert np.allclose(got_m, ref_m, rtol=RTOL, atol=1e-3), np.abs(got_m - ref_m).max()
  assert np.allclose(got_l, ref_l, rtol=RTOL, atol=1e-3)
  # host-batch entry (cell ids = position in the batch) and the jensen bound mllk >= mean llk - KL-ish slack
  ref_m2, _ = so.marginal_log_prob(spec, params, bn, x[rows], np.arange(len(rows)), S, library=lib[rows])
  got_m2, _ = e.marginal_llk(x=x[rows], library=lib[rows], n_samples=S)
  assert np.allclose(got_m2, ref_m2, rtol=RTOL, atol=1e-3)
  one, _ = e.marginal_llk(row_ids=rows, n_samples=1)
  assert np.isfinite(one).all()
  # the stacked form (all draws as rows of one decoder pass; VAE-family models) and the draw-by-draw form see the same
  # draws: both match the oracle and each other; several stacked passes fold into the same running log-sum-exp
  e.set_flag("stacked_scoring", False)
  loop_m, loop_l = e.marginal_llk(row_ids=rows, n_samples=S)
  e.set_flag("stacked_scoring", True)
  assert np.allclose(loop_m, ref_m, rtol=RTOL, atol=1e-3) and np.allclose(loop_l, ref_l, rtol=RTOL, atol=1e-3)
  assert np.allclose(loop_m, got_m, rtol=1e-5, atol=1e-3)
  from sisua_amd import _hip
  _hip.set_tuning("score_head_wide", 1)   # the f32 direct-operand head kernel in its likelihood-only mode
  try:
    wide_m, wide_l = e.marginal_llk(row_ids=rows, n_samples=S)
  finally:
    _hip.clear_tuning("score_head_wide")
  assert np.allclose(wide_m, ref_m, rtol=RTOL, atol=1e-3) and np.allclose(wide_m, got_m, rtol=1e-5, atol=1e-3)
  assert np.allclose(wide_l, got_l, rtol=1e-5, atol=1e-3)
  _hip.set_tuning("score_rows", 250)   # 5 draws of the 50 cells per pass: 5 + 5 + 2
  try:
    ch_m, ch_l = e.marginal_llk(row_ids=rows, n_samples=S)
  finally:
    _hip.clear_tuning("score_rows")
  assert np.allclose(ch_m, got_m, rtol=1e-5, atol=1e-3) and np.allclose(ch_l, got_l, rtol=1e-5, atol=1e-3)
  e.close()


@pytest.mark.parametrize("name,storage,n_rows,S", [("vae_zinb", "f32", 50, 40), ("vae_zinb", "u16", 37, 100), ("vae_nbd", "f32", 64, 24),
                                                   ("vae_zinbd", "csr", 128, 30), ("vae_zinb", "f32", 31, 70)])
def test_scoring_head_walk_equals_tile_per_workgroup(Engine, name, storage, n_rows, S):
  """The scoring head's two forms -- score_head_kernel (one 128 x 32 tile per workgroup: what short passes still use) and score_walk_kernel (a workgroup
  walks 256-row blocks under its gene tile's resident W image; from three blocks per workgroup up) -- give the same bits: the same products in the same
  order, the same likelihood code.  Cells per draw that divide 256 and that do not (the counts then change from block to block), every count store, the row
  ranges forced to 1 / 2 / 3 per gene tile, marginal_log_prob and the posterior-predictive scores."""
  from sisua_amd import _hip
  spec, cfg, x, ys, lib, mask = _problem(dict(CASES[name], labels=()))
  e = Engine(cfg, max_batch=128)
  e.upload(x, ys, lib, mask, storage=storage)
  rows = np.arange(5, 5 + n_rows, dtype=np.int32)

  def both():
    return e.marginal_llk(row_ids=rows, n_samples=S), e.score_llk([None, x[rows][:, ::-1].copy()], row_ids=rows, n_samples=S)
  walk = both()
  forms = {}
  for ranges in (0, 1, 2, 3):
    _hip.set_tuning("score_walk", ranges)
    try:
      forms[ranges] = both()
    finally:
      _hip.clear_tuning("score_walk")
  assert np.isfinite(walk[0][0]).all() and np.isfinite(walk[1]).all()
  for ranges, got in forms.items():
    assert np.array_equal(got[0][0], walk[0][0]) and np.array_equal(got[0][1], walk[0][1]), ranges
    assert np.array_equal(got[1], walk[1]), ranges
  e.close()


@pytest.mark.parametrize("latent_dim,dec", [(10, 40), (32, 128), (40, 64), (64, 96)])
def test_scoring_decoder_in_one_launch(Engine, latent_dim, dec):
  """A one-layer decoder with BatchNorm over the stacked draws in ONE launch (score_decoder1_kernel: the draws, the product, evaluation-mode BatchNorm,
  the activation and the bf16 x 3 split) against the three launches it replaces (knob no_score_dec1) and against the oracle: the draws are the same bits
  (one or two rows per wave by the latent width), the product the same ascending sum."""
  from sisua_amd import _hip
  spec, cfg, x, ys, lib, mask = _problem(dict(CASES["vae_zinb"], labels=(), latent_dim=latent_dim, dec_units=(dec,)))
  params = perturbed_params(spec)
  bn = so.init_bn_state(spec)
  e = Engine(cfg, max_batch=64, init=False)
  e.set_params(params)
  e.upload(x, ys, lib, mask)
  rows = np.arange(7, 60, dtype=np.int32)
  S = 9
  ref_m, ref_l = so.marginal_log_prob(spec, params, bn, x[rows], rows, S, library=lib[rows])
  one = e.marginal_llk(row_ids=rows, n_samples=S)
  _hip.set_tuning("no_score_dec1", 1)
  try:
    three = e.marginal_llk(row_ids=rows, n_samples=S)
  finally:
    _hip.clear_tuning("no_score_dec1")
  assert np.allclose(one[0], ref_m, rtol=RTOL, atol=1e-3) and np.allclose(one[1], ref_l, rtol=RTOL, atol=1e-3)
  assert np.allclose(one[0], three[0], rtol=1e-6, atol=1e-4) and np.allclose(one[1], three[1], rtol=1e-6, atol=1e-4)
  e.close()


def test_marginal_llk_stacked_two_layer_decoder_and_compact_store(Engine):
  """Stacked scoring through a two-layer decoder, the uint16 and the sparse store, 100 draws (posterior.py:964)."""
  spec, cfg, x, ys, lib, mask = _problem(dict(CASES["vae_zinb"], labels=(), dec_units=(40, 56)))
  params = perturbed_params(spec)
  bn = so.init_bn_state(spec)
  rows = np.arange(3, 40, dtype=np.int32)
  S = 100
  ref_m, ref_l = so.marginal_log_prob(spec, params, bn, x[rows], rows, S, library=lib[rows])
  for storage in ("f32", "u16", "csr"):
    e = Engine(cfg, max_batch=64, init=False)
    e.set_params(params)
    e.upload(x, ys, lib, mask, storage=storage)
    got_m, got_l = e.marginal_llk(row_ids=rows, n_samples=S)
    assert np.allclose(got_m, ref_m, rtol=RTOL, atol=1e-3), (storage, np.abs(got_m - ref_m).max())
    assert np.allclose(got_l, ref_l, rtol=RTOL, atol=1e-3), storage
    e.close()


@pytest.mark.parametrize("name", ["vae_zinb", "vae_nbd", "vae_zinbd", "scvi_zinbd", "scvi_nbd", "dca_zinb"])
def test_score_llk_matches_oracle(Engine, name):
  """Posterior.cal_llk (posterior.py:919-938) -- reconstructed / imputed likelihood of the
  original and the corrupted counts, log-mean-exp over posterior draws, GPU vs oracle."""
  spec, cfg, x, ys, lib, mask = _problem(dict(CASES[name], labels=()))
  params = perturbed_params(spec)
  bn = so.init_bn_state(spec)
  e = Engine(cfg, max_batch=64, init=False)
  e.set_params(params)
  x_cor = so.corrupt_binomial(x, 0.3, 0.2, seed=3).astype(np.float32)
  e.upload(x_cor, ys, lib, mask, cell_id_base=500)
  rows = np.arange(10, 60, dtype=np.int32)
  S = 6
  ref = so.posterior_llk(spec, params, bn, x_cor[rows], rows + 500, [x[rows], None], S, library=lib[rows])
  got = e.score_llk([x[rows], None], row_ids=rows, n_samples=S)
  assert got.shape == ref.shape == (2, 2, len(rows))
  assert np.allclose(got, ref, rtol=RTOL, atol=1e-3), np.abs(got - ref).max()
  if spec.likelihood in ("nb", "nbd"):
    assert np.array_equal(got[:, 0], got[:, 1])
  else:   # dropping the zero-inflation gate changes the score
    assert np.abs(got[:, 0] - got[:, 1]).max() > 1e-3
  # host-batch entry: cell ids = position in the batch
  ref2 = so.posterior_llk(spec, params, bn, x_cor[rows], np.arange(len(rows)), [x[rows]], S, library=lib[rows])
  got2 = e.score_llk([x[rows]], x=x_cor[rows], library=lib[rows], n_samples=S)
  assert np.allclose(got2, ref2, rtol=RTOL, atol=1e-3)
  # VAE-family models score all draws as rows of one decoder pass; the draw-by-draw form (and the f32 head kernel of the
  # stacked form) give the same scores; several stacked passes fold into the same running log-mean-exp
  e.set_flag("stacked_scoring", False)
  loop = e.score_llk([x[rows], None], row_ids=rows, n_samples=S)
  e.set_flag("stacked_scoring", True)
  assert np.allclose(loop, ref, rtol=RTOL, atol=1e-3) and np.allclose(loop, got, rtol=1e-5, atol=1e-3)
  from sisua_amd import _hip
  for var, val in (("score_head_wide", 1), ("score_rows", 100)):
    _hip.set_tuning(var, val)
    try:
      alt = e.score_llk([x[rows], None], row_ids=rows, n_samples=S)
    finally:
      _hip.clear_tuning(var)
    assert np.allclose(alt, got, rtol=1e-5, atol=1e-3), var
  e.close()


@pytest.mark.parametrize("name", ["vae_zinb", "scvi_zinbd"])
def test_clipnorm_bites_and_matches_oracle(Engine, name):
  """Per-tensor clipnorm (configs/base.yaml:46-50) with a threshold far below the gradient norms: from the
  second step on the clipped and unclipped Adam trajectories differ, so the norm path is what is tested."""
  kw = dict(CASES[name], clipnorm=0.05, lr=5e-3)
  spec, cfg, x, ys, lib, mask = _problem(kw)
  params = perturbed_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  e = Engine(cfg, max_batch=64, init=False)
  e.set_params(params)
  e.upload(x, ys, lib, mask, cell_id_base=0)
  got, ref, norms = [], [], None
  for step in range(6):
    rows = np.arange(step * 40, step * 40 + 64, dtype=np.int32) % x.shape[0]
    res = _oracle_step(spec, params, bn, opt, x, ys, lib, mask, rows, step)
    m = e.train_step(rows)
    got.append(m["loss"]); ref.append(res["metrics"]["loss"])
    norms = max(np.linalg.norm(g) for g in res["grads"].values())
    assert np.isclose(m["grad_norm_max"], norms, rtol=1e-3), (step, m["grad_norm_max"], norms)
  assert norms > 10 * spec.clipnorm          # the threshold really was exceeded
  assert np.allclose(got, ref, rtol=RTOL), np.abs(np.array(got) / np.array(ref) - 1).max()
  em, ev, where = adam_state_errors(e, opt)   # six steps of CLIPPED gradients: the moments carry the clip factors
  assert em < 1e-3 and ev < 2e-3, (em, ev, where)
  newp = e.get_params()
  # the same run without clipping ends somewhere else
  spec2, cfg2 = make_pair(**dict(kw, clipnorm=0.0))
  e2 = Engine(cfg2, max_batch=64, init=False)
  e2.set_params(perturbed_params(spec2))
  e2.upload(x, ys, lib, mask, cell_id_base=0)
  for step in range(6):
    e2.train_step(np.arange(step * 40, step * 40 + 64, dtype=np.int32) % x.shape[0])
  assert not np.allclose(e2.get_params()["lat/W"], newp["lat/W"], atol=1e-4)
  e.close(); e2.close()


def test_one_step_wide_gene_panel(Engine):
  """BASELINE configs[4] width (G = 20 000, the per-GPU slice of C5): exercises the 64-slice split-K products,
  wide groups as separate launches, 16-byte product stores and the 8-byte loss-kernel form against the oracle."""
  kw = dict(model="vae", n_genes=20000, likelihood="zinb", enc_units=(128,), dec_units=(128,), latent_dim=32)
  spec, cfg = make_pair(**kw)
  rng = np.random.default_rng(4)
  n, B = 96, 64
  x = (np.floor(rng.lognormal(0.0, 1.0, size=(n, 20000))) * (rng.uniform(size=(n, 20000)) < 0.1)).astype(np.float32)
  x[:, 0] += 1
  params = perturbed_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  e = Engine(cfg, max_batch=B, init=False)
  e.set_params(params)
  e.upload(x, cell_id_base=7)
  rows = np.arange(5, 5 + B, dtype=np.int32)
  res = so.train_step(spec, params, bn, opt, x[rows], so.PhiloxNoise(spec.seed, 0, rows + 7))
  m = e.train_step(rows)
  for key in ("loss", "nllk_x", "kl"):
    assert np.isclose(m[key], res["metrics"][key], rtol=RTOL, atol=1e-5), (key, m[key], res["metrics"][key])
  worst = grad_errors(e.get_params(which=1), res["grads"])
  assert max(worst.values()) < RTOL, sorted(worst.items(), key=lambda kv: -kv[1])[:3]
  assert np.isclose(m["grad_norm_max"], max(np.linalg.norm(g) for g in res["grads"].values()), rtol=1e-3)
  e.close()


@pytest.mark.parametrize("batch", [300, 600, 1100])
def test_one_step_large_batches(Engine, batch):
  """BatchNorm keeps 2 / 4 / 8 / 16 rows per lane in registers (batch <= 1024) and falls back to the round trip
  through memory beyond: batch 300 (8 rows), 600 (16 rows), 1100 (fallback) against the oracle."""
  kw = dict(model="vae", n_genes=150, likelihood="zinb", enc_units=(40, 24), dec_units=(24,), latent_dim=6, input_dropout=0.1)
  spec, cfg = make_pair(**kw)
  n = batch + 50
  x = synth_counts(n, 150, sparsity=0.8, seed=3)
  params = perturbed_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  e = Engine(cfg, max_batch=batch, init=False)
  e.set_params(params)
  e.upload(x, cell_id_base=5)
  rows = np.random.default_rng(2).permutation(n)[:batch].astype(np.int32)
  res = so.train_step(spec, params, bn, opt, x[rows], so.PhiloxNoise(spec.seed, 0, rows + 5))
  m = e.train_step(rows)
  for key in ("loss", "nllk_x", "kl"):
    assert np.isclose(m[key], res["metrics"][key], rtol=RTOL, atol=1e-5), (key, m[key], res["metrics"][key])
  worst = grad_errors(e.get_params(which=1), res["grads"])
  assert max(worst.values()) < RTOL, sorted(worst.items(), key=lambda kv: -kv[1])[:3]
  names = [p for p, _ in so.bn_manifest(spec)]
  for i, st in e.get_bn().items():
    assert np.allclose(st["moving_mean"], bn[f"{names[i]}/moving_mean"], rtol=1e-4, atol=1e-6)
    assert np.allclose(st["moving_var"], bn[f"{names[i]}/moving_var"], rtol=1e-4, atol=1e-6)
  e.close()


@pytest.mark.parametrize("name", ["vae_zinb", "scvi_zinbd", "sisua", "scale_post"])
def test_forward_samples_equals_repeated_forward(Engine, name):
  """smx_forward_samples (predict(sample_shape=n): encoders once, n re-sampled decodes) must return exactly what n
  calls of smx_forward(sample_index=s) return, for resident rows and for a host batch."""
  spec, cfg, x, ys, lib, mask = _problem(CASES[name])
  e = Engine(cfg, max_batch=48)
  e.upload(x, ys, lib, mask, cell_id_base=77)
  rows = np.arange(5, 45, dtype=np.int32)
  S = 4
  for kw in (dict(row_ids=rows), dict(x=x[rows], library=lib[rows])):
    many = e.forward_samples(S, **kw)
    for s in range(S):
      one = e.forward(sample_index=s, **kw)
      assert np.array_equal(many["x_params"][s], one["x_params"]) and np.array_equal(many["z_sample"][s], one["z_sample"])
      for a, b in zip(many["y_params"], one["y_params"]):
        assert np.array_equal(a[s], b)
      if spec.model == "scvi":
        assert np.array_equal(many["l_sample"][s], one["l_sample"])
    assert np.array_equal(many["z_mean"], one["z_mean"])
    assert not np.array_equal(many["z_sample"][0], many["z_sample"][1])
  e.close()


@pytest.mark.parametrize("labels", [((9, "mixnb4"),), ((6, "mixnb2"), (5, "mixnb2")),
                                    ((5, "mixnb4"), (4, "mixnb4"), (3, "mixnb4"), (6, "mixnb4")),
                                    ((14, "mixtril2"), (4, "nb")),   # 'mixtril': 2 x (2 + 14) = 32 planes of one head
                                    ((7, "mixzinb4"),)])
@pytest.mark.parametrize("S", [1, 3])
def test_predict_packs_every_label_plane(Engine, labels, S):
  """smx_predict hands its outputs to one pack launch per pass; a MISA model with four mixture components (3 latent + 3
  count planes + 12 label planes) or several mixture label heads needs more pack jobs than one launch's argument list
  held in round 2, and the planes beyond it were silently left out.  Every plane of every head must equal
  what smx_forward returns minibatch by minibatch -- bit for bit at one draw, to rounding with several (stacked decode)."""
  kw = dict(model="sisua", n_genes=90, likelihood="zinb", enc_units=(32,), dec_units=(32,), latent_dim=6, labels=labels)
  spec, cfg, x, ys, lib, mask = _problem(kw, n=100)
  e = Engine(cfg, max_batch=40)
  e.upload(x, ys, lib, mask)
  got = e.predict(x, n_samples=S, batch=40)
  for b0 in range(0, 100, 40):
    xb = x[b0:b0 + 40]
    for s in range(S):
      one = e.forward(x=xb, sample_index=s)
      cmp = np.array_equal if S == 1 else (lambda a, b: np.allclose(a, b, rtol=2e-5, atol=2e-5))
      assert cmp(got["x_params"][s][:, b0:b0 + 40], one["x_params"])
      assert cmp(got["z_sample"][s][b0:b0 + 40], one["z_sample"])
      for j, (a, b) in enumerate(zip(got["y_params"], one["y_params"])):
        assert np.isfinite(a).all() and cmp(a[s][b0:b0 + 40], b), (j, s, b0)
    assert np.array_equal(got["z_mean"][b0:b0 + 40], one["z_mean"])
  e.close()


@pytest.mark.parametrize("batch", [128, 100])
@pytest.mark.parametrize("name", ["vae_zinb", "vae_nb_nobn", "vae_zinbd", "sisua", "scvi_zinbd", "scvi_nbd", "paper_shape", "wide_panel_64",
                                  "wide_panel_128", "wide_panel_nobn_96"])
def test_bf16x3_products_match_oracle(Engine, name, batch):
  """The output head's training products (fused head, both products of its backward -- the scvi form with separate plane
  tensors included --, the first layer's weight gradient) formed from bf16 MFMAs on operands split three ways in registers
  (smx_device.h: six of the nine cross products; what is dropped is below one f32 rounding of a product): same 1e-4 bar on
  the ELBO scalars and on EVERY gradient as the exact-f32 MFMA forms, full and ragged minibatches; and the two forms agree
  with each other far inside that bar."""
  spec, cfg, x, ys, lib, mask = _problem(CASES[name])
  params = perturbed_params(spec)
  rows = np.random.default_rng(2).choice(x.shape[0], size=batch, replace=False).astype(np.int32)
  got = {}
  for b3 in (1, 0):
    e = Engine(cfg, max_batch=128, init=False)
    e.set_params(params)
    e.upload(x, ys, lib, mask, cell_id_base=1000)
    e.set_flag("bf16x3", bool(b3))
    m = e.train_step(rows)
    got[b3] = (m, e.get_params(which=1))
    e.close()
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  res = _oracle_step(spec, params, bn, opt, x, ys, lib, mask, rows, 0, cell_base=1000)
  m, grads = got[1]
  for key in ("loss", "nllk_x", "kl"):
    assert np.isclose(m[key], res["metrics"][key], rtol=RTOL, atol=1e-5), (key, m[key], res["metrics"][key])
  worst = grad_errors(grads, res["grads"])
  assert max(worst.values()) < RTOL, sorted(worst.items(), key=lambda kv: -kv[1])[:3]
  between = grad_errors(grads, got[0][1])
  assert max(between.values()) < 2e-5, sorted(between.items(), key=lambda kv: -kv[1])[:3]
  assert abs(m["loss"] - got[0][0]["loss"]) <= 2e-6 * abs(m["loss"])


@pytest.mark.parametrize("name", ["misa_tril", "scale_tril"])
def test_hip_matches_committed_variant_steps(Engine, name):
  """One training step of the round-3 variants -- MISA's 'mixtril' + zero-inflated mixture heads, SCALE with full-covariance
  components -- against COMMITTED numbers (tests/golden/oracle_variants_fixture.npz): loss terms and every gradient."""
  from tests.golden import make_head_fixtures as mk
  fx = _golden("oracle_variants_fixture.npz")
  spec, cfg = make_pair(**mk.CASES[name])
  names = [n for n, _ in so.manifest(spec)]
  e = Engine(cfg, max_batch=32, init=False)
  e.set_params({n: fx[f"{name}/p0/{n}"] for n in names})
  ys = [fx[f"{name}/y{j}"] for j in range(len(spec.labels))]
  e.upload(fx[f"{name}/x"], ys, None, fx[f"{name}/mask"] if ys else None, cell_id_base=mk.CELL_BASE)
  rows = np.arange(7, 7 + mk.B, dtype=np.int32)
  m = e.train_step(rows)
  for key in ("loss", "nllk_x", "kl") + (("nllk_y",) if ys else ()):
    assert np.isclose(m[key], float(fx[f"{name}/{key}"]), rtol=RTOL, atol=1e-5), (key, m[key], float(fx[f"{name}/{key}"]))
  worst = grad_errors(e.get_params(which=1), {n: fx[f"{name}/g/{n}"] for n in names})
  assert max(worst.values()) < RTOL, sorted(worst.items(), key=lambda kv: -kv[1])[:3]
  e.close()
