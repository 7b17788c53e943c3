"""The oracle against independent closed forms (scipy, torch.distributions) and
finite differences.  These are the cross-checks that stand in for the golden
vectors the reference does not have."""
import numpy as np
import pytest
import scipy.stats as st
import torch
import torch.distributions as td

from oracle import sisua_oracle as so


def _grid():
  rng = np.random.default_rng(0)
  x = np.concatenate([np.zeros(40), rng.integers(1, 12, 60), [181, 1000, 10738]])
  a = rng.uniform(-4, 6, x.size)
  l = rng.uniform(-8, 8, x.size)
  g = rng.uniform(-6, 6, x.size)
  return x[None, :], a[None, :], l[None, :], g[None, :]


def test_nb_matches_scipy_and_torch():
  x, a, l, g = _grid()
  llk, _ = so.count_llk(x, [a, l], "nb")
  r, p = np.exp(a), so.expit(l)
  ref = st.nbinom.logpmf(x, r, 1 - p)
  assert np.allclose(llk, ref, rtol=1e-10, atol=1e-9)
  tref = td.NegativeBinomial(total_count=torch.tensor(r), logits=torch.tensor(l)).log_prob(torch.tensor(x))
  assert np.allclose(llk, tref.numpy(), rtol=1e-9, atol=1e-8)


def test_zinb_matches_naive_mixture():
  x, a, l, g = _grid()
  llk, _ = so.count_llk(x, [a, l, g], "zinb")
  nb = np.exp(st.nbinom.logpmf(x, np.exp(a), 1 - so.expit(l)))
  pi = so.expit(g)
  naive = np.log(np.where(x == 0, pi + (1 - pi) * nb, (1 - pi) * nb))
  ok = np.isfinite(naive)
  assert ok.sum() > 90
  assert np.allclose(llk[ok], naive[ok], rtol=1e-9, atol=1e-9)


def test_nbd_is_nb_with_mean_dispersion():
  x, a, b, g = _grid()
  llk, _ = so.count_llk(x, [a, b], "nbd")
  mu, th = so.softplus(a), so.softplus1(b)
  ref = st.nbinom.logpmf(x, th, th / (th + mu))
  assert np.allclose(llk, ref, rtol=1e-5, atol=1e-5)  # scVI's 1e-8 epsilons
  llk_d, _ = so.count_llk(x, [mu, th], "nbd", direct=True)
  assert np.allclose(llk_d, llk, rtol=1e-12)


@pytest.mark.parametrize("lk", so.LIKELIHOODS)
@pytest.mark.parametrize("direct", [False, True])
def test_count_llk_gradients_fd(lk, direct):
  if direct and lk in ("nb", "zinb"):
    pytest.skip("direct mode is the scvi mean/dispersion form only")
  x, a, l, g = _grid()
  x = np.minimum(x, 300.0)
  if direct:
    a, l = so.softplus(a) + 0.05, so.softplus1(l)
  planes = [a, l, g][: so.n_params_per_gene(lk)]
  _, grads = so.count_llk(x, planes, lk, direct=direct)
  for i in range(len(planes)):
    h = 1e-6
    pp = [p.copy() for p in planes]; pm = [p.copy() for p in planes]
    pp[i] += h; pm[i] -= h
    fd = (so.count_llk(x, pp, lk, direct=direct)[0] - so.count_llk(x, pm, lk, direct=direct)[0]) / (2 * h)
    assert np.allclose(grads[i], fd, rtol=2e-5, atol=2e-6), (lk, i)


def test_kl_terms_match_torch():
  rng = np.random.default_rng(1)
  mu, s = rng.normal(size=(5, 7)), rng.uniform(0.2, 2.0, size=(5, 7))
  kl = 0.5 * (s ** 2 + mu ** 2 - 1 - 2 * np.log(s)).sum(1)
  ref = td.kl_divergence(td.Independent(td.Normal(torch.tensor(mu), torch.tensor(s)), 1),
                         td.Independent(td.Normal(torch.zeros(5, 7, dtype=torch.float64),
                                                  torch.ones(5, 7, dtype=torch.float64)), 1))
  assert np.allclose(kl, ref.numpy(), rtol=1e-12)


def test_label_llk_onehot_and_nb():
  rng = np.random.default_rng(2)
  y = np.eye(7)[rng.integers(0, 7, 9)]
  raw = rng.normal(size=(9, 7))
  ll, d = so.label_llk(y, raw, "onehot")
  ref = td.OneHotCategorical(logits=torch.tensor(raw)).log_prob(torch.tensor(y))
  assert np.allclose(ll, ref.numpy(), rtol=1e-12)
  yy = rng.uniform(0.5, 9.1, size=(9, 4))
  raw = rng.normal(size=(9, 8))
  ll, d = so.label_llk(yy, raw, "nb")
  h = 1e-6
  for idx in [(0, 0), (3, 5), (8, 7)]:
    rp, rm = raw.copy(), raw.copy()
    rp[idx] += h; rm[idx] -= h
    fd = (so.label_llk(yy, rp, "nb")[0].sum() - so.label_llk(yy, rm, "nb")[0].sum()) / (2 * h)
    assert np.isclose(d[idx], fd, rtol=1e-5, atol=1e-7)


def test_label_llk_mixture_nb_matches_torch_mixture_same_family():
  """MISA's label head (vae.py:47-98): per-dimension mixture of C negative binomials == torch
  MixtureSameFamily(Categorical(logits), NegativeBinomial(total_count, logits)); gradients by central differences."""
  rng = np.random.default_rng(4)
  B, P = 7, 5
  for C in (2, 3):
    y = np.floor(rng.uniform(0, 12, size=(B, P)))
    raw = rng.normal(size=(B, 3 * C * P))
    ll, d = so.label_llk(y, raw, f"mixnb{C}")
    pl = torch.tensor(raw).reshape(B, 3 * C, P)
    mix = td.Categorical(logits=pl[:, :C].permute(0, 2, 1))                                   # [B, P] batch, C categories
    comp = td.NegativeBinomial(total_count=torch.exp(pl[:, C:2 * C].permute(0, 2, 1)), logits=pl[:, 2 * C:].permute(0, 2, 1))
    ref = td.MixtureSameFamily(mix, comp).log_prob(torch.tensor(y)).sum(1)
    assert np.allclose(ll, ref.numpy(), rtol=1e-12)
    h = 1e-6
    for idx in [(0, 0), (3, C * P + 2), (6, 3 * C * P - 1), (2, 2 * C * P + 1)]:
      rp, rm = raw.copy(), raw.copy()
      rp[idx] += h; rm[idx] -= h
      fd = (so.label_llk(y, rp, f"mixnb{C}")[0].sum() - so.label_llk(y, rm, f"mixnb{C}")[0].sum()) / (2 * h)
      assert np.isclose(d[idx], fd, rtol=1e-5, atol=1e-7), (C, idx)
    assert so.label_planes(f"mixnb{C}") == 3 * C


def test_label_llk_mixture_normal_matches_torch_mixture_same_family():
  """MISA's head for continuous labels ('mixgaussian', vae.py:86-92): per-dimension mixture of C normals with scale =
  softplus(raw + softplus_inverse(1)) == torch MixtureSameFamily(Categorical(logits), Normal); gradients by central differences."""
  rng = np.random.default_rng(5)
  B, P = 6, 4
  for C in (2, 4):
    y = rng.normal(1.0, 2.0, size=(B, P))
    raw = rng.normal(size=(B, 3 * C * P))
    ll, d = so.label_llk(y, raw, f"mixgauss{C}")
    pl = torch.tensor(raw).reshape(B, 3 * C, P)
    mix = td.Categorical(logits=pl[:, :C].permute(0, 2, 1))
    scale = torch.nn.functional.softplus(pl[:, 2 * C:].permute(0, 2, 1) + so.SOFTPLUS_INV_1)
    ref = td.MixtureSameFamily(mix, td.Normal(pl[:, C:2 * C].permute(0, 2, 1), scale)).log_prob(torch.tensor(y)).sum(1)
    assert np.allclose(ll, ref.numpy(), rtol=1e-12)
    h = 1e-6
    for idx in [(0, 0), (3, C * P + 2), (5, 3 * C * P - 1), (2, 2 * C * P + 1)]:
      rp, rm = raw.copy(), raw.copy()
      rp[idx] += h; rm[idx] -= h
      fd = (so.label_llk(y, rp, f"mixgauss{C}")[0].sum() - so.label_llk(y, rm, f"mixgauss{C}")[0].sum()) / (2 * h)
      assert np.isclose(d[idx], fd, rtol=1e-5, atol=1e-7), (C, idx)
    assert so.label_planes(f"mixgauss{C}") == 3 * C and so.mixture_components(f"mixgauss{C}") == C


def test_label_llk_mixture_zinb_matches_scipy():
  """MISA(zero_inflated=True) (sisua/models/vae.py:76-84): every label dimension a C-component mixture of ZERO-INFLATED negative
  binomials -- values against scipy's nbinom with the gate mixed in by hand, gradients by central differences."""
  import scipy.stats as st
  from scipy.special import expit, logsumexp, log_softmax
  rng = np.random.default_rng(7)
  B, P = 6, 4
  for C in (2, 3):
    kind = f"mixzinb{C}"
    assert so.label_planes(kind) == 4 * C and so.mixture_components(kind) == C
    y = rng.poisson(3.0, size=(B, P)).astype(np.float64) * (rng.uniform(size=(B, P)) > 0.4)
    raw = rng.normal(size=(B, 4 * C * P)) * 0.8
    ll, d = so.label_llk(y, raw, kind)
    pl = raw.reshape(B, 4 * C, P)
    nb = st.nbinom(np.exp(pl[:, C:2 * C]), expit(-pl[:, 2 * C:3 * C])).pmf(y[:, None, :])
    gate = expit(pl[:, 3 * C:])
    f = np.where(y[:, None, :] == 0, gate + (1 - gate) * nb, (1 - gate) * nb)
    ref = logsumexp(log_softmax(pl[:, :C], axis=1) + np.log(f), axis=1).sum(1)
    assert np.allclose(ll, ref, rtol=1e-10)
    h = 1e-6
    for col in rng.choice(4 * C * P, 14, replace=False):
      rp, rm = raw.copy(), raw.copy()
      rp[:, col] += h; rm[:, col] -= h
      fd = (so.label_llk(y, rp, kind)[0] - so.label_llk(y, rm, kind)[0]) / (2 * h)
      assert np.allclose(d[:, col], fd, rtol=2e-5, atol=1e-7), (C, col)


def test_label_llk_mixture_tril_matches_torch_mixture_same_family():
  """MISA's 'mixtril' head (the docstring example of sisua/models/vae.py:58): ONE mixture of C full-covariance Gaussians over the
  whole label vector == torch MixtureSameFamily(Categorical(logits), MultivariateNormal(loc, scale_tril)), diag(L) = softplus(raw)
  + 1e-5; gradients by central differences; the inert entries (logit planes beyond column 0, L above its diagonal) get none."""
  rng = np.random.default_rng(6)
  B = 5
  for C, P in ((2, 5), (3, 4), (4, 3)):
    kind = f"mixtril{C}"
    ky = so.label_planes(kind, P)
    assert ky == C * (2 + P) and so.mixture_components(kind) == C
    y = rng.normal(0.5, 1.5, size=(B, P))
    raw = rng.normal(size=(B, ky * P)) * 0.7
    ll, d = so.label_llk(y, raw, kind)
    pl = torch.tensor(raw).reshape(B, ky, P)
    cols = pl[:, 2 * C:].reshape(B, C, P, P)                     # [b, c, j, p] = L_c[p][j]
    Lr = cols.permute(0, 1, 3, 2)
    L = torch.tril(Lr, -1) + torch.diag_embed(torch.nn.functional.softplus(torch.diagonal(Lr, dim1=-2, dim2=-1)) + so.TRIL_DIAG_SHIFT)
    ref = td.MixtureSameFamily(td.Categorical(logits=pl[:, :C, 0]), td.MultivariateNormal(pl[:, C:2 * C], scale_tril=L)).log_prob(torch.tensor(y))
    assert np.allclose(ll, ref.numpy(), rtol=1e-11, atol=1e-11)
    h = 1e-6
    inert = np.ones(ky * P, bool)
    for c in range(C):
      inert[c * P] = False
      inert[(C + c) * P:(C + c + 1) * P] = False
      for j in range(P):
        k = 2 * C + c * P + j
        inert[k * P + j:(k + 1) * P] = False
    assert np.all(d[:, inert] == 0.0) and np.all(np.abs(d[:, ~inert]).max(0) > 0)
    for col in rng.choice(np.flatnonzero(~inert), 12, replace=False):
      rp, rm = raw.copy(), raw.copy()
      rp[:, col] += h; rm[:, col] -= h
      fd = (so.label_llk(y, rp, kind)[0] - so.label_llk(y, rm, kind)[0]) / (2 * h)
      assert np.allclose(d[:, col], fd, rtol=2e-5, atol=1e-7), (C, P, col)


# ---------------------------------------------------------------------------
# whole-step gradient check by central differences, every model family
# ---------------------------------------------------------------------------
def _toy(model, lk, labels=(), bn=True, **kw):
  G = 13
  if model == "scale_post":
    model, kw = "scale", dict(kw, latent_mixture=True, n_components=3)
  kw.setdefault("n_components", 4)
  spec = so.Spec(model=model, n_genes=G, likelihood=lk, enc_units=(6, 5), dec_units=(7,), latent_dim=3,
                 encl_units=(4,), labels=labels, batchnorm=bn, dropout_enc=0.25, dropout_dec=0.25,
                 input_dropout=0.2, seed=3, disc_units=5, disc_layers=2, **kw)
  rng = np.random.default_rng(5)
  B = 6
  x = rng.poisson(1.5, size=(B, G)).astype(np.float64) * (rng.uniform(size=(B, G)) < 0.6)
  y = []
  for P, kind in labels:
    y.append(np.eye(P)[rng.integers(0, P, B)] if kind == "onehot" else rng.uniform(0.5, 9.0, size=(B, P)))
  lc, lm, lv = so.library_size(x + 1.0)
  lib = np.tile(np.array([[float(lm), float(lv) + 0.05]]), (B, 1))
  mask = np.array([1, 0, 1, 1, 0, 1], dtype=bool)
  params = so.init_params(spec)
  for name in params:  # move off the symmetric init (gamma=1, beta=0, b=0)
    params[name] = params[name] + 0.1 * rng.normal(size=params[name].shape)
  return spec, params, so.init_bn_state(spec), x, y, lib, mask


def test_scale_tril_prior_gradients_fd():
  """SCALE with covariance='tril' (scale.py:28,35): every entry of the prior's tensors by central differences -- the lower triangles
  and diagonals of the scale factors move the loss, the entries above the diagonals do not (and get a zero gradient)."""
  spec, params, bn_state, x, y, lib, mask = _toy("scale", "zinb", (), True, covariance="tril")
  D, C = spec.latent_dim, spec.n_components
  assert dict(so.manifest(spec))["prior/scale"] == (C * D, D)
  noise = so.PhiloxNoise(spec.seed, 7, np.arange(x.shape[0]) + 100)
  res = so.forward_backward(spec, params, bn_state, x, noise)
  g = res["grads"]["prior/scale"].reshape(C, D, D)
  assert np.all(np.triu(g, 1) == 0.0) and np.all(np.abs(np.tril(g)).reshape(C, -1).max(1) > 0)
  h = 1e-5
  for name in ("prior/scale", "prior/loc", "prior/logits"):
    flat = params[name].reshape(-1)
    for idx in range(flat.size):
      pp = {k: v.copy() for k, v in params.items()}
      pm = {k: v.copy() for k, v in params.items()}
      pp[name].reshape(-1)[idx] += h
      pm[name].reshape(-1)[idx] -= h
      fd = (so.forward_backward(spec, pp, bn_state, x, noise, backward=False)["loss"] - so.forward_backward(spec, pm, bn_state, x, noise, backward=False)["loss"]) / (2 * h)
      assert np.isclose(res["grads"][name].reshape(-1)[idx], fd, rtol=2e-4, atol=2e-8), (name, idx)
  # the initial prior is the unit Gaussian mixture: L = I
  p0 = so.init_params(spec)["prior/scale"].reshape(C, D, D)
  assert np.allclose(np.tril(p0, -1), 0) and np.allclose(so.softplus(np.einsum("cpp->cp", p0)), 1.0)
  with pytest.raises(AssertionError):
    so.Spec(model="scale", n_genes=5, covariance="tril", tie_loc=True)


CASES = [("vae", "zinb", (), True), ("vae", "nb", (), False), ("vae", "zinbd", (), True),
         ("vae", "nbd", (), True), ("dca", "zinb", (), True), ("scvi", "zinbd", (), True),
         ("scvi", "nbd", (), False), ("sisua", "zinb", ((4, "nb"), (3, "onehot")), True),
         ("sisua", "zinb", ((4, "mixnb2"), (3, "mixnb3")), True),   # MISA
         ("sisua", "nb", ((3, "mixgauss2"), (4, "nb")), True),      # MISA with a continuous label variable
         ("sisua", "zinb", ((4, "mixtril2"),), True),               # MISA, full-covariance mixture over the label vector (vae.py:58)
         ("sisua", "nb", ((4, "mixzinb2"),), True),                 # MISA(zero_inflated=True)
         ("scale", "zinb", (), True), ("scale", "nb", (), False),    # SCALE: mixture prior, Monte-Carlo KL
         ("scale_post", "zinb", (), True), ("scale_post", "nb", (), False),   # SCALE read literally: mixture-density posterior
         ("fvae", "zinb", (), True), ("fvae", "nb", ((3, "onehot"),), False),   # FVAE / SemiFVAE: two objectives
         ("fvae", "zinb", ((3, "onehot"), (4, "onehot"), (2, "onehot")), True)]     # SemiFVAE, several label variables (round 6)


# round 5 (VERDICT r04 Missing 5): outputs[1:] on FactorVAE (fvae.py:9-18 passes `outputs` through unchanged), on SemiFVAE (the label
# variable then sits BEHIND the observed outputs in the target order) and on the mixture-density posterior
FVAE_EXTRA_CASES = [("fvae", "zinb", (), True, ((4, "nbd"),)), ("fvae", "nb", ((3, "onehot"),), False, ((3, "nb"), (2, "zinb"))),
                    ("scale_post", "zinb", (), True, ((4, "nbd"),))]


@pytest.mark.parametrize("model,lk,labels,bn,extras", [c + ((),) for c in CASES] + FVAE_EXTRA_CASES)
def test_full_step_gradients_fd(model, lk, labels, bn, extras):
  spec, params, bn_state, x, y, lib, mask = _toy(model, lk, labels, bn, **(dict(extra_outputs=extras) if extras else {}))
  if extras:   # targets in head order: the extra outputs, then the labels
    r9 = np.random.default_rng(9)
    y = [np.eye(P)[r9.integers(0, P, x.shape[0])] if kind == "onehot" else r9.uniform(0.5, 9.0, size=(x.shape[0], P)) for P, kind in extras] + y
  noise = so.PhiloxNoise(spec.seed, 7, np.arange(x.shape[0]) + 100)

  def loss_of(p, key="loss"):
    return so.forward_backward(spec, p, bn_state, x, noise, y=y, library=lib, mask=mask, backward=False)[key]

  res = so.forward_backward(spec, params, bn_state, x, noise, y=y, library=lib, mask=mask)
  assert np.isfinite(res["loss"])
  rng = np.random.default_rng(11)
  assert set(res["grads"]) == {n for n, _ in so.manifest(spec)}
  for name, g in res["grads"].items():
    assert g.shape == params[name].shape
    flat = params[name].reshape(-1)
    for idx in rng.choice(flat.size, size=min(4, flat.size), replace=False):
      h = 1e-5
      pp = {k: v.copy() for k, v in params.items()}
      pm = {k: v.copy() for k, v in params.items()}
      pp[name].reshape(-1)[idx] += h
      pm[name].reshape(-1)[idx] -= h
      # fvae: the discriminator's tensors follow ITS objective (z a constant of it, which holds under this
      # perturbation: z does not depend on them); every other tensor the VAE objective with the discriminator fixed
      key = "dtc_loss" if name.startswith("disc") else "loss"
      fd = (loss_of(pp, key) - loss_of(pm, key)) / (2 * h)
      assert np.isclose(g.reshape(-1)[idx], fd, rtol=2e-4, atol=1e-7), (name, idx, g.reshape(-1)[idx], fd)


EXTRA_CASES = {
    # outputs[1:] of the reference's constructors (tests/test_singlecell_models.py:129-141; scvi.py:168-169): fully observed heads
    "vae_two_outputs": ("vae", "zinb", dict(extra_outputs=((4, "nbd"),))),
    "vae_three_outputs": ("vae", "nb", dict(extra_outputs=((3, "zinb"), (4, "zinbd")))),
    "sisua_extra_output_and_labels": ("sisua", "zinb", dict(extra_outputs=((3, "nb"),), labels=((4, "nbd"), (3, "onehot")))),
    "scvi_two_outputs": ("scvi", "zinbd", dict(extra_outputs=((4, "nbd"),))),
    # scvi.py:55-56,66-86: per-gene dispersion / inflation vectors instead of Dense heads
    "scvi_share_dispersion": ("scvi", "zinbd", dict(dispersion="share")),
    "scvi_share_both": ("scvi", "zinbd", dict(dispersion="share", inflation="share")),
    "scvi_nbd_share_dispersion": ("scvi", "nbd", dict(dispersion="share")),
    "scvi_single_dispersion_share_inflation": ("scvi", "zinbd", dict(dispersion="single", inflation="share")),
}


@pytest.mark.parametrize("name", list(EXTRA_CASES))
def test_extra_outputs_and_scvi_options_gradients_fd(name):
  """Whole-step central differences for the heads of outputs[1:] (observed: weight 1, no mask -- masked-out cells DO contribute) and
  for scvi's per-gene dispersion / inflation vectors (tensors out1/b, out2/b without a kernel)."""
  model, lk, kw = EXTRA_CASES[name]
  labels = kw.pop("labels", ()) if "labels" in kw else ()
  kw = dict(kw)
  spec, params, bn_state, x, y_lab, lib, mask = _toy(model, lk, labels, True, **kw)
  rng = np.random.default_rng(9)
  B = x.shape[0]
  y = [np.eye(P)[rng.integers(0, P, B)] if kind == "onehot" else rng.uniform(0.5, 9.0, size=(B, P)) for P, kind in spec.extra_outputs] + y_lab
  noise = so.PhiloxNoise(spec.seed, 7, np.arange(B) + 100)
  names = {n for n, _ in so.manifest(spec)}
  if spec.model == "scvi":
    assert ("out1/W" in names) == (spec.dispersion == "full") and "out1/b" in names
    assert ("out2/W" in names) == (spec.inflation == "full" and spec.k == 3)
  res = so.forward_backward(spec, params, bn_state, x, noise, y=y, library=lib, mask=mask)
  assert set(res["grads"]) == names
  if spec.extra_outputs:   # the observed heads' term is there for EVERY cell, whatever the label mask
    r0 = so.forward_backward(spec, params, bn_state, x, noise, y=y, library=lib, mask=np.zeros(B, bool), backward=False)
    assert np.allclose(r0["llk_o"], res["llk_o"]) and np.all(r0["llk_o"] != 0)
    assert np.isclose(res["loss"], -(res["llk_x"] + res["llk_o"] + spec.alpha * mask * res["llk_y"] - spec.beta * (res["kl"] + res["kl_l"])).mean())

  def loss_of(p):
    return so.forward_backward(spec, p, bn_state, x, noise, y=y, library=lib, mask=mask, backward=False)["loss"]

  for nm, g in res["grads"].items():
    flat = params[nm].reshape(-1)
    for idx in rng.choice(flat.size, size=min(4, flat.size), replace=False):
      h = 1e-5
      pp = {k: v.copy() for k, v in params.items()}
      pm = {k: v.copy() for k, v in params.items()}
      pp[nm].reshape(-1)[idx] += h
      pm[nm].reshape(-1)[idx] -= h
      fd = (loss_of(pp) - loss_of(pm)) / (2 * h)
      assert np.isclose(g.reshape(-1)[idx], fd, rtol=2e-4, atol=1e-7), (nm, idx, g.reshape(-1)[idx], fd)


def test_scale_tied_mixture_gradients_fd():
  """SCALE's tied mixture parameters (scale.py:29-33): one location and / or one scale vector shared by every component, the
  weights fixed uniform.  A tied tensor keeps identical rows and every row receives the derivative with respect to the SHARED
  value: central differences that move all rows of a column together; the logits of fixed weights get no gradient."""
  for ties in (dict(tie_loc=True), dict(tie_scale=True, tie_mixtures=True), dict(tie_loc=True, tie_scale=True, tie_mixtures=True)):
    spec, params, bn_state, x, y, lib, mask = _toy("scale", "zinb", (), True, **ties)
    rng = np.random.default_rng(2)
    for name, on in (("prior/loc", spec.tie_loc), ("prior/scale", spec.tie_scale)):
      if on:   # a tied tensor has identical rows (here: off the init by one shared perturbation)
        params[name] = np.broadcast_to(params[name][:1], params[name].shape).copy()
    if spec.tie_mixtures:
      params["prior/logits"] = np.zeros_like(params["prior/logits"])
    noise = so.PhiloxNoise(spec.seed, 7, np.arange(x.shape[0]) + 100)
    loss_of = lambda p: so.forward_backward(spec, p, bn_state, x, noise, y=y, library=lib, mask=mask, backward=False)["loss"]
    res = so.forward_backward(spec, params, bn_state, x, noise, y=y, library=lib, mask=mask)
    g = res["grads"]
    if spec.tie_mixtures:
      assert not g["prior/logits"].any()
    for name, on in (("prior/loc", spec.tie_loc), ("prior/scale", spec.tie_scale)):
      if on:
        assert np.array_equal(g[name], np.broadcast_to(g[name][:1], g[name].shape))       # every row the same
      for d in rng.choice(params[name].shape[1], size=2, replace=False):
        h = 1e-5
        pp = {k: v.copy() for k, v in params.items()}
        pm = {k: v.copy() for k, v in params.items()}
        rows = slice(None) if on else slice(1, 2)
        pp[name][rows, d] += h
        pm[name][rows, d] -= h
        fd = (loss_of(pp) - loss_of(pm)) / (2 * h)
        assert np.isclose(g[name][1, d], fd, rtol=2e-4, atol=1e-7), (ties, name, d, g[name][1, d], fd)
    # init: a tied location starts at zero in every row, the other tensors keep their streams
    p0, p1 = so.init_params(spec), so.init_params(so.Spec(**{**spec.__dict__, "tie_loc": False, "tie_scale": False, "tie_mixtures": False}))
    for k in p0:
      if k == "prior/loc" and spec.tie_loc:
        assert not p0[k].any()
      else:
        assert np.array_equal(p0[k], p1[k]), k


def test_adam_clipnorm_step():
  spec = so.Spec(model="vae", n_genes=4, enc_units=(3,), dec_units=(3,), latent_dim=2, clipnorm=1.0)
  params = so.init_params(spec)
  p0 = {k: v.copy() for k, v in params.items()}
  grads = {k: np.full_like(v, 3.0) for k, v in params.items()}
  opt = so.init_opt_state(params)
  norms = so.adam_update(spec, params, grads, opt)
  for k, v in params.items():
    assert np.isclose(norms[k], 3.0 * np.sqrt(v.size))
    # first Adam step moves every weight by ~lr regardless of gradient scale
    assert np.allclose(p0[k] - v, spec.lr, rtol=1e-4)
  # second step with tiny gradient: clip inactive
  grads = {k: np.full_like(v, 1e-3) for k, v in params.items()}
  so.adam_update(spec, params, grads, opt)
  assert opt["t"] == 2


def test_data_side_semantics():
  tr, te = so.split_indices(100, 0.8, seed=1)
  assert len(tr) == 80 and len(te) == 20 and len(set(tr) | set(te)) == 100
  order = so.epoch_order(2500, 0, shuffle=1000, seed=1)
  assert sorted(order.tolist()) == list(range(2500))
  assert order[:10].max() < 1010  # streaming buffer: early picks come from the head
  assert not np.array_equal(order, so.epoch_order(2500, 1, shuffle=1000, seed=1))
  bs = so.batches(order, 128)
  assert len(bs) == 19 and all(len(b) == 128 for b in bs)
  assert not so.label_mask(50, 0.1, n_omics=1).any()
  m = so.label_mask(5000, 0.1, n_omics=2)
  assert abs(m.mean() - 0.1) < 0.02


def test_posterior_llk_against_scipy_nbinom():
  """posterior_llk (Posterior.cal_llk, posterior.py:919-938): the 'imputed' score must equal the plain NB
  log-pmf of the target under the decoded parameters (scipy, independent code), log-mean-exp over draws."""
  from scipy import stats
  from scipy.special import logsumexp
  spec = so.Spec(model="vae", n_genes=40, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=4, seed=5)
  params = so.init_params(spec)
  bn = so.init_bn_state(spec)
  rng = np.random.default_rng(0)
  x = rng.poisson(1.0, (9, 40)).astype(np.float64)
  tgt = rng.poisson(1.5, (9, 40)).astype(np.float64)
  ids = np.arange(9) + 100
  S = 5
  got = so.posterior_llk(spec, params, bn, x, ids, [tgt, None], S)
  per_draw = []
  for s_ in range(S):
    r = so.forward_backward(spec, params, bn, x, so.PhiloxNoise(spec.seed, 0, ids, sample=s_), training=False,
                            backward=False)
    a, l = r["x_params"][0], r["x_params"][1]   # total_count = exp(a), logits l: p_success = sigmoid(l)
    per_draw.append(stats.nbinom.logpmf(tgt, np.exp(a), 1.0 - 1.0 / (1.0 + np.exp(-l))).sum(1))
  ref = logsumexp(np.stack(per_draw), axis=0) - np.log(S)
  assert np.allclose(got[0, 1], ref, rtol=1e-9, atol=1e-9)
  assert got.shape == (2, 2, 9) and (got[:, 0] != got[:, 1]).any()


def test_mse_output_is_minus_the_mean_squared_error_with_its_gradient():
  """'mse' output of the oracle (RVmeta(dim, 'mse'), tests/test_singlecell_models.py:82-91 of the reference): the per-cell
  log-likelihood is minus the mean over the genes of the squared error; gradient by central differences; a VAE / DCA step with
  it runs and its loss is nllk_x (+ KL)."""
  rng = np.random.default_rng(1)
  x = rng.poisson(3.0, size=(5, 17)).astype(np.float64)
  mu = rng.normal(2.0, 1.0, size=(5, 17))
  ell, (d0,) = so.count_llk(x, [mu], "mse")
  assert np.allclose(ell.sum(1), -np.mean((x - mu) ** 2, axis=1), rtol=1e-14)
  e = 1e-6
  up, _ = so.count_llk(x, [mu + e], "mse")
  dn, _ = so.count_llk(x, [mu - e], "mse")
  assert np.allclose((up - dn) / (2 * e), d0, rtol=1e-7, atol=1e-9)
  for model in ("dca", "vae"):
    spec = so.Spec(model=model, n_genes=17, likelihood="mse", enc_units=(8,), dec_units=(8,), latent_dim=3)
    assert spec.k == 1 and dict(so.manifest(spec))["out/W"] == (8, 17)
    p = so.init_params(spec)
    r = so.forward_backward(spec, p, so.init_bn_state(spec), x[:4], so.PhiloxNoise(spec.seed, 0, np.arange(4)))
    assert np.isclose(r["loss"], r["metrics"]["nllk_x"] + r["metrics"]["kl"]) and r["metrics"]["nllk_x"] > 0
    assert set(r["grads"]) == set(p)
