"""An INDEPENDENT second implementation of the whole training step, checked against the oracle.

`oracle/sisua_oracle.py` derives every gradient by hand.  Here the same step is written once more in torch float64
with nothing shared but the frozen third-party semantics: likelihoods and KL terms come from
`torch.distributions` wherever a distribution exists there (NegativeBinomial, Normal, kl_divergence,
OneHotCategorical), gradients from `torch.autograd`, the optimiser from `torch.optim.Adam` (with the epsilon
re-expressed per step so that it is Keras's `m / (sqrt(v) + eps)`), clipping by hand (torch's helper adds 1e-6 to the
norm).  The noise (dropout masks, eps) is the only thing taken from the oracle: it is an input of the step.
Agreement: loss and updated parameters to 1e-9, every gradient to 1e-8 (relative L2), BatchNorm moving statistics to 1e-12, two steps,
all four model families + the ablations the GPU tests run.  CPU only."""
import numpy as np
import pytest
import torch
import torch.distributions as td

from oracle import sisua_oracle as so
from tests.util import perturbed_params, synth_counts, synth_labels

torch.set_default_dtype(torch.float64)
SP1 = float(np.log(np.expm1(1.0)))


def softplus1(x):
  return torch.nn.functional.softplus(x + SP1)


def mlp(spec, P, bn, prefix, units, h, noise, stream0, p_drop, new_bn):
  """Dense (no bias under BN) -> BatchNorm(batch statistics, biased variance) -> ReLU -> inverted Dropout."""
  for i, _ in enumerate(units):
    pre = h @ P[f"{prefix}{i}/W"]
    if spec.batchnorm:
      mu = pre.mean(0)
      var = ((pre - mu) ** 2).mean(0)
      new_bn[f"{prefix}{i}/moving_mean"] = spec.bn_momentum * bn[f"{prefix}{i}/moving_mean"] + (1 - spec.bn_momentum) * mu.detach().numpy()
      new_bn[f"{prefix}{i}/moving_var"] = spec.bn_momentum * bn[f"{prefix}{i}/moving_var"] + (1 - spec.bn_momentum) * var.detach().numpy()
      y = P[f"{prefix}{i}/gamma"] * (pre - mu) / torch.sqrt(var + spec.bn_eps) + P[f"{prefix}{i}/beta"]
    else:
      y = pre + P[f"{prefix}{i}/b"]
    h = torch.relu(y) * torch.as_tensor(noise.dropout(stream0 + i, y.shape[1], p_drop))
  return h


def count_log_prob(x, planes, likelihood, direct):
  """log p(x) per (cell, gene).  'nb'/'zinb': TFP NegativeBinomial(total_count = exp(a), logits) -- the torch
  distribution of the same name has the same convention; 'nbd'/'zinbd': the scVI formula (there is no distribution
  object with its epsilon terms); zero inflation: log(pi 1[x = 0] + (1 - pi) NB(x)) through logsumexp."""
  if likelihood in ("nb", "zinb"):
    lp = td.NegativeBinomial(total_count=torch.exp(planes[0]), logits=planes[1], validate_args=False).log_prob(x)   # (protein levels are real-valued)
  else:
    mu, th = (planes[0], planes[1]) if direct else (torch.nn.functional.softplus(planes[0]), softplus1(planes[1]))
    e = so.NBD_EPS
    lp = (th * (torch.log(th + e) - torch.log(th + mu + e)) + x * (torch.log(mu + e) - torch.log(th + mu + e))
          + torch.lgamma(x + th) - torch.lgamma(th) - torch.lgamma(x + 1))
  if likelihood in ("zinb", "zinbd"):
    g = planes[2]
    log_pi, log_1mpi = torch.nn.functional.logsigmoid(g), torch.nn.functional.logsigmoid(-g)
    zero = torch.logsumexp(torch.stack([log_pi, log_1mpi + lp]), 0)
    lp = torch.where(x == 0, zero, log_1mpi + lp)
  return lp


def torch_step(spec, params, bn, x, noise, y=(), library=None, mask=None):
  P = {k: torch.tensor(v, requires_grad=True) for k, v in params.items()}
  x = torch.as_tensor(np.asarray(x, np.float64))
  B, G, D = x.shape[0], spec.n_genes, spec.latent_dim
  new_bn = {}
  h0 = torch.log1p(x) if spec.log_norm else x
  h0 = h0 * torch.as_tensor(noise.dropout(so.STREAM_INPUT_DROPOUT, G, spec.input_dropout))
  h = mlp(spec, P, bn, "enc", spec.enc_units, h0, noise, so.STREAM_ENC_DROPOUT, spec.dropout_enc, new_bn)
  lat = h @ P["lat/W"] + P["lat/b"]
  if spec.latent_mixture:   # SCALE read literally (scale.py:26,38-47): a mixture-density POSTERIOR, standard-normal prior, Monte-Carlo KL
    C = spec.n_components
    mu_c, sig_c = lat[:, D:(1 + C) * D].reshape(B, C, D), softplus1(lat[:, (1 + C) * D:].reshape(B, C, D))
    qm = td.MixtureSameFamily(td.Categorical(logits=lat[:, :C]), td.Independent(td.Normal(mu_c, sig_c), 1))
    pi32 = torch.softmax(lat[:, :C], 1).detach().numpy().astype(np.float32)
    u = noise.uniform(so.STREAM_MIX_PICK, 1)[:, 0].astype(np.float32)
    pick = torch.as_tensor(np.minimum((np.cumsum(pi32, axis=1) < u[:, None]).sum(1), C - 1))
    rows = torch.arange(B)
    z = mu_c[rows, pick] + sig_c[rows, pick] * torch.as_tensor(noise.normal(so.STREAM_EPS_Z, D))   # reparameterised through the picked component only
    kl = qm.log_prob(z) - td.Independent(td.Normal(torch.zeros_like(z), torch.ones_like(z)), 1).log_prob(z)
  elif spec.stochastic:
    q = td.Normal(lat[:, :D], softplus1(lat[:, D:]))
    z = q.loc + q.scale * torch.as_tensor(noise.normal(so.STREAM_EPS_Z, D))
    if spec.model == "scale":   # Monte-Carlo KL against the trainable mixture prior: log q(z|x) - log p(z)
      if spec.scale_tril:   # covariance = 'tril' (scale.py:28): full-covariance components, diag(L) = softplus(raw) + 1e-5
        Lr = P["prior/scale"].reshape(spec.n_components, D, D)
        L = torch.tril(Lr, -1) + torch.diag_embed(torch.nn.functional.softplus(torch.diagonal(Lr, dim1=-2, dim2=-1)) + so.TRIL_DIAG_SHIFT)
        comp = td.MultivariateNormal(P["prior/loc"], scale_tril=L)
      else:
        comp = td.Independent(td.Normal(P["prior/loc"], softplus1(P["prior/scale"])), 1)
      prior = td.MixtureSameFamily(td.Categorical(logits=P["prior/logits"]), comp)
      kl = td.Independent(q, 1).log_prob(z) - prior.log_prob(z)
    else:
      kl = td.kl_divergence(q, td.Normal(torch.zeros_like(q.loc), torch.ones_like(q.scale))).sum(1)
  else:
    z = torch.relu(lat) if spec.latent_activation == "relu" else lat
    kl = torch.zeros(B)
  kl_l = torch.zeros(B)
  if spec.model == "scvi":
    hl = mlp(spec, P, bn, "encl", spec.encl_units, h0, noise, so.STREAM_ENCL_DROPOUT, spec.dropout_enc, new_bn)
    latl = hl @ P["latl/W"] + P["latl/b"]
    ql = td.Normal(latl[:, 0], softplus1(latl[:, 1]))
    l = ql.loc + ql.scale * torch.as_tensor(noise.normal(so.STREAM_EPS_L, 1)[:, 0])
    lib = torch.as_tensor(np.asarray(library, np.float64))
    kl_l = td.kl_divergence(ql, td.Normal(lib[:, 0], torch.sqrt(lib[:, 1])))
  d = mlp(spec, P, bn, "dec", spec.dec_units, z, noise, so.STREAM_DEC_DROPOUT, spec.dropout_dec, new_bn)
  if spec.model == "scvi":
    # dispersion / inflation other than 'full' (scvi.py:66-86): no Dense head, one trainable per-gene vector shared by every cell
    raw = [(d @ P[f"out{c}/W"] if f"out{c}/W" in P else 0.0) + P[f"out{c}/b"].expand(B, G) for c in range(spec.k)]
    rho = torch.clamp(torch.softmax(raw[0], dim=1), so.SCVI_RHO_MIN, 1 - so.SCVI_RHO_MIN)
    rate = torch.exp(torch.clamp(l, 0.0, spec.clip_library))[:, None] * rho
    planes = [rate, torch.exp(raw[1])] + ([raw[2]] if spec.k == 3 else [])
    llk_x = count_log_prob(x, planes, spec.likelihood, True).sum(1)
  else:
    raw = d @ P["out/W"] + P["out/b"]
    llk_x = count_log_prob(x, [raw[:, c * G:(c + 1) * G] for c in range(spec.k)], spec.likelihood, False).sum(1)
  llk_y, llk_o = torch.zeros(B), torch.zeros(B)
  m = torch.zeros(B) if mask is None else torch.as_tensor(np.asarray(mask, np.float64))
  extra_vae, j_disc = 0.0, None
  if spec.model == "fvae":
    # FactorVAE (Kim & Mnih 2018, Algorithm 2).  Two objectives, kept apart with detach(): the VAE objective sees the
    # discriminator as a fixed function (frozen copies of its tensors), the discriminator objective sees z as a constant.
    def disc(inp, W):
      hh = inp
      for i in range(spec.disc_layers):
        hh = torch.nn.functional.leaky_relu(hh @ W[f"disc{i}/W"] + W[f"disc{i}/b"], spec.disc_leak)
      return hh @ W["discout/W"] + W["discout/b"]
    frozen = {k: v.detach() for k, v in P.items() if k.startswith("disc")}
    order = torch.argsort(torch.as_tensor(noise.uniform(so.STREAM_PERMUTE, D)), dim=0, stable=True)   # rank of (u, row) per column
    zd = z.detach()
    logits_v, logits_z, logits_p = disc(z, frozen), disc(zd, P), disc(torch.gather(zd, 0, order), P)
    extra_vae = spec.gamma * torch.logsumexp(logits_v, 1).mean()
    j_disc = 0.5 * (torch.nn.functional.softplus(-torch.logsumexp(logits_z, 1)).mean()
                    + torch.nn.functional.softplus(torch.logsumexp(logits_p, 1)).mean())
    c0 = 0
    for jl, (Pj, _) in enumerate(spec.labels):   # every label variable: a categorical over ITS columns of the logit layer (the TC logit above: over all of them)
      yj = torch.as_tensor(np.asarray(y[len(spec.extra_outputs) + jl], np.float64))   # (behind the observed outputs in the target order)
      llk_y = llk_y + td.OneHotCategorical(logits=logits_v[:, c0:c0 + Pj]).log_prob(yj)
      j_disc = j_disc - spec.alpha * (m * td.OneHotCategorical(logits=logits_z[:, c0:c0 + Pj]).log_prob(yj)).mean()
      c0 += Pj
  # heads on the decoder output: the extra OUTPUT variables first (fully observed: weight 1, no mask), then the label variables
  heads = [(Pj, kind, True) for Pj, kind in spec.extra_outputs] + ([] if spec.model == "fvae" else [(Pj, kind, False) for Pj, kind in spec.labels])
  for j, (Pj, kind, observed) in enumerate(heads):
    ry = d @ P[f"lab{j}/W"] + P[f"lab{j}/b"]
    yj = torch.as_tensor(np.asarray(y[j], np.float64))
    llk_prev, llk_y = llk_y, torch.zeros(B)
    if kind == "nb":
      # protein levels are real-valued (dataset.html:187): the same density formula, support check off
      llk_y = llk_y + td.NegativeBinomial(total_count=torch.exp(ry[:, :Pj]), logits=ry[:, Pj:], validate_args=False).log_prob(yj).sum(1)
    elif kind in ("nbd", "zinb", "zinbd"):   # the other count posteriors, planes as for the gene output
      kk = 2 if kind == "nbd" else 3
      llk_y = llk_y + count_log_prob(yj, [ry[:, c * Pj:(c + 1) * Pj] for c in range(kk)], kind, False).sum(1)
    elif kind.startswith("mixtril"):   # MISA's docstring example (vae.py:58): ONE mixture of C full-covariance Gaussians over the label vector
      C = int(kind[-1])
      pl = ry.reshape(B, C * (2 + Pj), Pj)
      Lr = pl[:, 2 * C:].reshape(B, C, Pj, Pj).permute(0, 1, 3, 2)          # planes are the COLUMNS of L
      L = torch.tril(Lr, -1) + torch.diag_embed(torch.nn.functional.softplus(torch.diagonal(Lr, dim1=-2, dim2=-1)) + so.TRIL_DIAG_SHIFT)
      llk_y = llk_y + td.MixtureSameFamily(td.Categorical(logits=pl[:, :C, 0]), td.MultivariateNormal(pl[:, C:2 * C], scale_tril=L)).log_prob(yj)
    elif kind.startswith("mixgauss"):   # MISA, continuous labels: MixtureSameFamily over C normals per label dimension
      C = int(kind[-1])
      pl = ry.reshape(B, 3 * C, Pj)
      comp = td.Normal(pl[:, C:2 * C].permute(0, 2, 1), torch.nn.functional.softplus(pl[:, 2 * C:].permute(0, 2, 1) + so.SOFTPLUS_INV_1))
      llk_y = llk_y + td.MixtureSameFamily(td.Categorical(logits=pl[:, :C].permute(0, 2, 1)), comp).log_prob(yj).sum(1)
    elif kind.startswith("mixzinb"):   # MISA(zero_inflated=True): the components zero-inflated (no torch class: the gate mixed in by hand)
      C = int(kind[-1])
      pl = ry.reshape(B, 4 * C, Pj)
      nb = td.NegativeBinomial(total_count=torch.exp(pl[:, C:2 * C]), logits=pl[:, 2 * C:3 * C], validate_args=False).log_prob(yj[:, None, :])
      g = pl[:, 3 * C:]
      comp = torch.where(yj[:, None, :] == 0, torch.logaddexp(g, nb), nb) - torch.nn.functional.softplus(g)
      llk_y = llk_y + torch.logsumexp(torch.log_softmax(pl[:, :C], 1) + comp, 1).sum(1)
    elif kind.startswith("mixnb"):   # MISA: MixtureSameFamily over C negative binomials per label dimension
      C = int(kind[-1])
      pl = ry.reshape(B, 3 * C, Pj)
      comp = td.NegativeBinomial(total_count=torch.exp(pl[:, C:2 * C].permute(0, 2, 1)), logits=pl[:, 2 * C:].permute(0, 2, 1),
                                 validate_args=False)
      llk_y = llk_y + td.MixtureSameFamily(td.Categorical(logits=pl[:, :C].permute(0, 2, 1)), comp, validate_args=False).log_prob(yj).sum(1)
    else:
      llk_y = llk_y + td.OneHotCategorical(logits=ry).log_prob(yj)
    if observed:
      llk_o, llk_y = llk_o + llk_y, llk_prev
    else:
      llk_y = llk_prev + llk_y
  loss = -(llk_x + llk_o + spec.alpha * m * llk_y - spec.beta * (kl + kl_l)).mean() + extra_vae
  (loss if j_disc is None else loss + j_disc).backward()
  return P, float(loss.detach()), {k: v.grad.numpy() for k, v in P.items()}, new_bn


def keras_adam(spec, P, state, t):
  """torch.optim.Adam with Keras's epsilon: torch divides by sqrt(v / (1 - b2^t)) + eps, Keras by sqrt(v) + eps under
  lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); they coincide for eps_torch = eps / sqrt(1 - b2^t)."""
  if "opt" not in state:
    state["opt"] = torch.optim.Adam(list(P.values()), lr=spec.lr, betas=(spec.adam_beta1, spec.adam_beta2), eps=spec.adam_eps)
    state["names"] = list(P)
  opt = state["opt"]
  for grp in opt.param_groups:
    grp["eps"] = spec.adam_eps / np.sqrt(1.0 - spec.adam_beta2 ** t)
  for p in P.values():   # per-tensor clipnorm (configs/base.yaml:50)
    n = float(p.grad.norm())
    if spec.clipnorm > 0 and n > spec.clipnorm:
      p.grad.mul_(spec.clipnorm / n)
  opt.step()


CASES = {
    "vae_zinb": dict(model="vae", n_genes=60, likelihood="zinb", enc_units=(24, 20), dec_units=(20,), latent_dim=6, input_dropout=0.2),
    "vae_nb_nobn": dict(model="vae", n_genes=40, likelihood="nb", enc_units=(16,), dec_units=(16,), latent_dim=5, batchnorm=False),
    "vae_zinbd": dict(model="vae", n_genes=50, likelihood="zinbd", enc_units=(16,), dec_units=(16, 12), latent_dim=4),
    "vae_nbd_clip": dict(model="vae", n_genes=33, likelihood="nbd", enc_units=(12,), dec_units=(12,), latent_dim=3, clipnorm=0.05),
    "dca_zinb": dict(model="dca", n_genes=45, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=4),
    "dca_linear": dict(model="dca", n_genes=30, likelihood="nb", enc_units=(8,), dec_units=(8,), latent_dim=4, latent_activation="linear"),
    "sisua": dict(model="sisua", n_genes=48, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=5,
                  labels=((7, "nb"), (4, "onehot")), alpha=10.0),
    "misa": dict(model="sisua", n_genes=40, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=5,
                 labels=((6, "mixnb2"), (3, "mixnb3")), alpha=10.0),
    "misa_gauss": dict(model="sisua", n_genes=36, likelihood="nb", enc_units=(16,), dec_units=(16,), latent_dim=4,
                       labels=((5, "mixgauss3"), (4, "mixnb2")), alpha=10.0),
    "misa_zi": dict(model="sisua", n_genes=36, likelihood="nb", enc_units=(16,), dec_units=(16,), latent_dim=4,
                    labels=((6, "mixzinb2"),), alpha=10.0),
    "misa_tril": dict(model="sisua", n_genes=36, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=4,
                      labels=((5, "mixtril3"), (3, "onehot")), alpha=10.0),
    "scale": dict(model="scale", n_genes=44, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=5, n_components=6),
    "scale_tril": dict(model="scale", n_genes=40, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=5, n_components=4, covariance="tril"),
    "scale_post": dict(model="scale", n_genes=40, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=5, n_components=3, latent_mixture=True),
    "scalar": dict(model="scale", n_genes=42, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=5, n_components=4,
                   labels=((6, "nb"), (3, "onehot")), alpha=10.0),   # SCALE + label heads (scale.py:52-59)
    "fvae": dict(model="fvae", n_genes=44, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=5, disc_units=24, disc_layers=3),
    "semifvae": dict(model="fvae", n_genes=40, likelihood="nb", enc_units=(16,), dec_units=(16,), latent_dim=4, disc_units=20, disc_layers=2,
                     labels=((5, "onehot"),), gamma=3.0, alpha=4.0),
    # round 6: SemiFVAE with several label variables (one logit per class of every variable; TC logit over all, cross-entropy per variable)
    "semifvae_three_labels": dict(model="fvae", n_genes=40, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=4, disc_units=20, disc_layers=2,
                                  labels=((5, "onehot"), (2, "onehot"), (7, "onehot")), gamma=3.0, alpha=4.0),
    "semifvae_two_labels_two_outputs": dict(model="fvae", n_genes=36, likelihood="nb", enc_units=(16,), dec_units=(16,), latent_dim=4, disc_units=20, disc_layers=2,
                                            extra_outputs=((5, "nbd"),), labels=((3, "onehot"), (4, "onehot")), alpha=6.0),
    "scvi_zinbd": dict(model="scvi", n_genes=52, likelihood="zinbd", enc_units=(16,), dec_units=(16,), latent_dim=4, encl_units=(8,)),
    "scvi_nbd": dict(model="scvi", n_genes=36, likelihood="nbd", enc_units=(12,), dec_units=(12,), latent_dim=3, encl_units=(6,),
                     batchnorm=False),
    # outputs[1:] (tests/test_singlecell_models.py:129-141: VAE(outputs=[zinb genes, nbd proteins])): fully observed heads, weight 1
    "vae_two_outputs": dict(model="vae", n_genes=48, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=5, extra_outputs=((7, "nbd"),)),
    "vae_three_outputs": dict(model="vae", n_genes=40, likelihood="nb", enc_units=(16,), dec_units=(16,), latent_dim=4,
                              extra_outputs=((6, "zinb"), (5, "zinbd"))),
    "sisua_extra_output": dict(model="sisua", n_genes=44, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=5,
                               extra_outputs=((6, "nb"),), labels=((5, "nbd"), (4, "onehot")), alpha=10.0),
    # round 5: outputs[1:] on FactorVAE / SemiFVAE (its label variable behind the observed outputs) and on the mixture-density posterior
    "fvae_two_outputs": dict(model="fvae", n_genes=44, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=5, disc_units=24, disc_layers=3,
                             extra_outputs=((6, "nbd"),)),
    "semifvae_two_outputs": dict(model="fvae", n_genes=40, likelihood="nb", enc_units=(16,), dec_units=(16,), latent_dim=4, disc_units=20, disc_layers=2,
                                 extra_outputs=((5, "zinb"),), labels=((4, "onehot"),), alpha=5.0),
    "scale_post_two_outputs": dict(model="scale", n_genes=40, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=5, n_components=3, latent_mixture=True,
                                   extra_outputs=((6, "nb"),)),
    "dca_two_outputs": dict(model="dca", n_genes=36, likelihood="zinb", enc_units=(12,), dec_units=(12,), latent_dim=4, extra_outputs=((5, "onehot"),)),
    # scvi.py:168-169 (posteriors[1:] on the decoder output) and scvi.py:55-56,66-86 (dispersion / inflation without a head)
    "scvi_two_outputs": dict(model="scvi", n_genes=40, likelihood="zinbd", enc_units=(16,), dec_units=(16,), latent_dim=4, encl_units=(8,),
                             extra_outputs=((6, "nbd"),)),
    "scvi_gene_dispersion": dict(model="scvi", n_genes=44, likelihood="zinbd", enc_units=(16,), dec_units=(16,), latent_dim=4, encl_units=(8,),
                                 dispersion="share"),
    "scvi_gene_both": dict(model="scvi", n_genes=36, likelihood="zinbd", enc_units=(12,), dec_units=(12,), latent_dim=3, encl_units=(6,),
                           dispersion="share", inflation="share"),
    "scvi_single": dict(model="scvi", n_genes=40, likelihood="zinbd", enc_units=(12,), dec_units=(12,), latent_dim=3, encl_units=(6,),
                        dispersion="single", inflation="single"),
    "scvi_nbd_gene_dispersion": dict(model="scvi", n_genes=36, likelihood="nbd", enc_units=(12,), dec_units=(12,), latent_dim=3, encl_units=(6,),
                                     dispersion="share", batchnorm=False),
}


@pytest.mark.parametrize("name", list(CASES))
def test_torch_autograd_step_equals_the_oracle(name):
  spec = so.Spec(**CASES[name])
  n, B = 90, 40
  x = synth_counts(n, spec.n_genes, sparsity=0.7, seed=2, max_count=300)
  ys = synth_labels(n, spec.extra_outputs + spec.labels)
  _, lm, lv = so.library_size(x)
  lib = np.tile(np.array([[lm, lv]]), (n, 1))
  mask = so.label_mask(n, 0.5, n_omics=1 + len(spec.labels), seed=1)
  params = perturbed_params(spec)
  t_params = {k: v.copy() for k, v in params.items()}
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  t_bn, state = {k: v.copy() for k, v in bn.items()}, {}
  P = None
  for step in range(2):
    rows = np.random.default_rng(step).permutation(n)[:B]
    kw = dict(y=[a[rows] for a in ys], library=lib[rows], mask=mask[rows])
    res = so.train_step(spec, params, bn, opt, x[rows], so.PhiloxNoise(spec.seed, step, rows), **kw)
    if P is None:
      P, loss, grads, new_bn = torch_step(spec, t_params, t_bn, x[rows], so.PhiloxNoise(spec.seed, step, rows), **kw)
    else:   # the same leaf tensors keep their optimiser state across steps
      for p in P.values():
        p.grad = None
      cur = {k: v.detach().numpy().copy() for k, v in P.items()}
      P2, loss, grads, new_bn = torch_step(spec, cur, t_bn, x[rows], so.PhiloxNoise(spec.seed, step, rows), **kw)
      for k in P:
        P[k].grad = P2[k].grad
    assert np.isclose(loss, res["loss"], rtol=1e-9, atol=0), (step, loss, res["loss"])   # (scvi: 1e-10 from exp / softmax in a different order)
    top = max(np.linalg.norm(g) for g in res["grads"].values())
    for k, g in res["grads"].items():
      assert np.linalg.norm(grads[k] - g) <= 1e-8 * max(np.linalg.norm(g), 1e-6 * top), (step, k)
    keras_adam(spec, P, state, step + 1)
    t_bn.update(new_bn)
    for k in params:
      assert np.allclose(P[k].detach().numpy(), params[k], rtol=1e-9, atol=1e-11), (step, k)
    for k in bn:
      assert np.allclose(t_bn[k], bn[k], rtol=1e-12, atol=1e-14), (step, k)
  if "clip" in name:   # the clipped branch really ran
    assert max(np.linalg.norm(g) for g in res["grads"].values()) > spec.clipnorm
