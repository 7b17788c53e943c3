"""Result objects (sisua_amd/distributions.py) against scipy / torch closed forms and the
structural assertions the reference's tests make on model outputs
(tests/test_singlecell_models.py:41-91, 105-114)."""
import numpy as np
import scipy.stats as st
import torch
import torch.distributions as td

from sisua_amd import distributions as D


def test_nb_and_zinb_surface():
  rng = np.random.default_rng(0)
  a, l, g = rng.normal(size=(5, 7)), rng.normal(size=(5, 7)), rng.normal(size=(5, 7))
  x = rng.poisson(2.0, size=(5, 7)).astype(float)
  nb = D.NegativeBinomial(np.exp(a), l)
  assert np.allclose(nb.log_prob(x), st.nbinom.logpmf(x, np.exp(a), 1 - 1 / (1 + np.exp(-l))))
  tnb = td.NegativeBinomial(torch.tensor(np.exp(a)), logits=torch.tensor(l))
  assert np.allclose(nb.mean(), tnb.mean.numpy()) and np.allclose(nb.variance(), tnb.variance.numpy())
  zi = D.ZeroInflated(nb, g)
  pi = 1 / (1 + np.exp(-g))
  naive = np.log(np.where(x == 0, pi + (1 - pi) * np.exp(nb.log_prob(x)), (1 - pi) * np.exp(nb.log_prob(x))))
  assert np.allclose(zi.log_prob(x), naive)
  ind = D.Independent(zi, 1, name="transcriptomic")
  assert ind.batch_shape == (5,) and ind.event_shape == (7,) and ind.name == "transcriptomic"
  assert ind.distribution is zi and zi.count_distribution is nb and ind.reinterpreted_batch_ndims == 1
  assert np.allclose(ind.log_prob(x), zi.log_prob(x).sum(-1))
  s = ind.sample(11, seed=1)
  assert s.shape == (11, 5, 7) and (s >= 0).all()
  big = D.ZeroInflated(D.NegativeBinomial(np.full(20000, 3.0), np.full(20000, 0.5)), np.full(20000, -1.0))
  smp = big.sample(seed=3)
  assert abs(smp.mean() - big.mean()[0]) < 0.15 and abs(smp.var() - big.variance()[0]) < 1.0


def test_nbd_matches_mean_dispersion_nb():
  rng = np.random.default_rng(1)
  mu, th = rng.lognormal(size=(4, 6)), rng.lognormal(size=(4, 6))
  x = rng.poisson(2.0, size=(4, 6)).astype(float)
  d = D.NegativeBinomialDisp(mu, th)
  assert np.allclose(d.log_prob(x), st.nbinom.logpmf(x, th, th / (th + mu)), atol=1e-5)
  assert np.allclose(d.variance(), mu + mu ** 2 / th)


def test_gaussians_onehot_and_concat():
  rng = np.random.default_rng(2)
  mu, s = rng.normal(size=(6, 3)), rng.uniform(0.5, 2, size=(6, 3))
  q = D.MultivariateNormalDiag(mu, s, name="Latents")
  z = rng.normal(size=(6, 3))
  ref = td.Independent(td.Normal(torch.tensor(mu), torch.tensor(s)), 1).log_prob(torch.tensor(z)).numpy()
  assert np.allclose(q.log_prob(z), ref) and q.batch_shape == (6,) and q.event_shape == (3,)
  assert np.allclose(q.mean(), mu) and np.allclose(q.variance(), s ** 2) and q.sample(4).shape == (4, 6, 3)
  oh = D.OneHotCategorical(rng.normal(size=(6, 5)))
  y = np.eye(5)[rng.integers(0, 5, 6)]
  assert np.allclose(oh.log_prob(y), td.OneHotCategorical(logits=torch.tensor(oh.logits)).log_prob(torch.tensor(y)).numpy())
  assert oh.sample(seed=0).sum(-1).tolist() == [1.0] * 6
  parts = [D.Independent(D.ZeroInflated(D.NegativeBinomial(np.ones((2, 4)) * i, np.zeros((2, 4))), np.zeros((2, 4))), 1,
                         name="transcriptomic") for i in (1, 2, 3)]
  cat = D.concat_distributions(parts, axis=0, name="transcriptomic")
  assert cat.batch_shape == (6,) and cat.event_shape == (4,) and isinstance(cat.distribution, D.ZeroInflated)
  assert np.allclose(cat.distribution.count_distribution.total_count[:, 0], [1, 1, 2, 2, 3, 3])
  # leading Monte-Carlo axis: merge along axis 1 (single_cell_model.py:184-187)
  mc = [D.Independent(D.NegativeBinomial(np.ones((3, 2, 4)), np.zeros((3, 2, 4))), 1) for _ in range(2)]
  assert D.concat_distributions(mc, axis=1).batch_shape == (3, 4)
  det = D.concat_distributions([D.Deterministic(np.ones((2, 3))), D.Deterministic(np.zeros((1, 3)))])
  assert det.mean().shape == (3, 3)


def test_count_distribution_factory():
  p = [np.zeros((2, 3)), np.ones((2, 3)), np.zeros((2, 3))]
  d = D.count_distribution("zinb", p, "transcriptomic", activated=False)
  assert isinstance(d.distribution, D.ZeroInflated) and np.allclose(d.distribution.count_distribution.total_count, 1.0)
  d2 = D.count_distribution("nbd", p[:2], "transcriptomic", activated=False)
  assert isinstance(d2.distribution, D.NegativeBinomialDisp)
  assert np.allclose(d2.distribution.loc, np.log(2.0)) and np.allclose(d2.mean(), np.log(2.0))
  d3 = D.count_distribution("zinbd", [np.full((2, 3), 5.0), np.full((2, 3), 2.0), p[2]], "x", activated=True)
  assert np.allclose(d3.distribution.count_distribution.loc, 5.0)


def test_mixture_negative_binomial_surface():
  """MISA label heads: log_prob against torch MixtureSameFamily, moments against a large sample."""
  import torch
  import torch.distributions as td
  from sisua_amd import distributions as D
  rng = np.random.default_rng(0)
  B, C, P = 4, 3, 5
  mix, r, l = rng.normal(size=(B, C, P)), np.exp(rng.normal(size=(B, C, P))), rng.normal(size=(B, C, P))
  d = D.MixtureNegativeBinomial(mix, r, l)
  assert d.batch_shape == (B, P)
  y = np.floor(rng.uniform(0, 9, size=(B, P)))
  ref = td.MixtureSameFamily(td.Categorical(logits=torch.tensor(mix).permute(0, 2, 1)),
                             td.NegativeBinomial(total_count=torch.tensor(r).permute(0, 2, 1), logits=torch.tensor(l).permute(0, 2, 1)))
  assert np.allclose(d.log_prob(y), ref.log_prob(torch.tensor(y)).numpy(), rtol=1e-10)
  assert np.allclose(d.mean(), ref.mean.numpy(), rtol=1e-10) and np.allclose(d.variance(), ref.variance.numpy(), rtol=1e-10)
  s = d.sample(20000, seed=1)
  assert s.shape == (20000, B, P)
  assert np.allclose(s.mean(0), d.mean(), rtol=0.15, atol=0.05)
  ind = D.Independent(d, 1, name="proteomic")
  assert ind.event_shape == (P,) and ind.batch_shape == (B,) and np.allclose(ind.log_prob(y), d.log_prob(y).sum(-1))


def test_mse_posterior_identity():
  """The one exact numeric identity the reference's tests hold for this path (tests/test_singlecell_models.py:82-91):
  for RVmeta(dim, 'mse') the output is a VectorDeterministic and `-dist.log_prob(z) == tf.losses.mse(z, y.mean())` EXACTLY
  (`np.all(d1 == d2)`), where tf.losses.mse is the mean over the last axis of the squared difference in the inputs' float32."""
  from sisua_amd import distributions as D
  from sisua_amd.config import RVmeta
  rv = RVmeta(12, "mse")
  assert rv.is_deterministic and not rv.is_zero_inflated
  rng = np.random.default_rng(0)
  mean = rng.normal(size=(8, 12)).astype(np.float32)               # the reference's x has shape (8, ...), dim 12
  y = D.count_distribution("mse", [mean], "x", activated=False)
  assert isinstance(y, D.VectorDeterministic) and y.event_shape == (12,) and y.batch_shape == (8,)
  assert np.array_equal(y.mean(), mean) and np.all(y.variance() == 0) and np.array_equal(y.sample(3)[2], mean)
  z = rng.uniform(size=(8, 12)).astype(np.float32)
  d1 = -y.log_prob(z).ravel()
  d2 = np.mean(np.square(z - y.mean()), axis=-1).ravel()           # tf.losses.mse(z, mean): K.mean(squared_difference, axis=-1)
  assert d1.dtype == np.float32 and np.all(d1 == d2)
  import torch
  d3 = torch.nn.functional.mse_loss(torch.from_numpy(mean), torch.from_numpy(z), reduction="none").mean(-1).numpy()
  assert np.allclose(d1, d3, rtol=3e-7, atol=0)                    # another library's summation order: to the last bits
  # leading sample axes broadcast as the other result distributions do
  ys = D.VectorDeterministic(np.stack([mean, mean + 1]), name="x")
  assert ys.batch_shape == (2, 8) and ys.log_prob(z).shape == (2, 8)


def test_mixture_multivariate_normal_tril_surface():
  """MISA's 'mixtril' head as a distribution: log_prob / mean / covariance against torch's MixtureSameFamily over
  MultivariateNormal(scale_tril), samples against its moments, concatenation along the batch axis."""
  rng = np.random.default_rng(2)
  N, C, P = 7, 3, 4
  logits, loc = rng.normal(size=(N, C)), rng.normal(size=(N, C, P))
  L = np.tril(rng.normal(size=(N, C, P, P)) * 0.4, -1) + np.eye(P) * rng.uniform(0.5, 1.5, size=(N, C, P, 1))
  d = D.MixtureMultivariateNormalTriL(logits, loc, L, name="proteomic")
  ref = td.MixtureSameFamily(td.Categorical(logits=torch.tensor(logits)), td.MultivariateNormal(torch.tensor(loc), scale_tril=torch.tensor(np.tril(L))))
  x = rng.normal(size=(N, P))
  assert d.batch_shape == (N,) and d.event_shape == (P,)
  assert np.allclose(d.log_prob(x), ref.log_prob(torch.tensor(x)).numpy(), rtol=1e-11, atol=1e-11)
  assert np.allclose(d.mean(), ref.mean.numpy()) and np.allclose(d.variance(), ref.variance.numpy())
  s = d.sample(20000, seed=1)
  assert s.shape == (20000, N, P)
  assert np.abs(s.mean(0) - d.mean()).max() < 0.06 and np.abs(s.var(0) - d.variance()).max() < 0.15
  cov = np.einsum("snp,snq->npq", s - s.mean(0), s - s.mean(0)) / s.shape[0]
  assert np.abs(cov - d.covariance()).max() < 0.15
  both = D.concat_distributions([D.MixtureMultivariateNormalTriL(logits[:3], loc[:3], L[:3]), D.MixtureMultivariateNormalTriL(logits[3:], loc[3:], L[3:])])
  assert np.array_equal(both.log_prob(x), d.log_prob(x))
