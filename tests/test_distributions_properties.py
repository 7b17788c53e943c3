"""Property tests of the result distributions (sisua_amd/distributions.py) against scipy.stats: what
sisua.analysis reads from predict() -- log_prob, mean, variance -- for both NB parametrisations and the
zero-inflated wrapper."""
import numpy as np
from hypothesis import given, settings, strategies as st
from scipy import stats
from scipy.special import expit

from sisua_amd import distributions as D

SET = settings(max_examples=60, deadline=None, derandomize=True)


@SET
@given(r=st.floats(0.05, 500.0), l=st.floats(-8.0, 6.0), g=st.floats(-6.0, 6.0))
def test_total_count_logits_form_matches_scipy(r, l, g):
  x = np.arange(0, 40, dtype=np.float64)
  nb = D.NegativeBinomial(np.full(x.shape, r), np.full(x.shape, l))
  p_fail = 1.0 - expit(l)                      # scipy's p is the probability of the 'stopping' event
  ref = stats.nbinom(r, p_fail)
  assert np.allclose(nb.log_prob(x), ref.logpmf(x), rtol=1e-9, atol=1e-9)
  assert np.isclose(nb.mean()[0], ref.mean(), rtol=1e-9) and np.isclose(nb.variance()[0], ref.var(), rtol=1e-9)
  zi = D.ZeroInflated(nb, np.full(x.shape, g))
  pi = expit(g)
  ref_zi = np.where(x == 0, np.log(pi + (1 - pi) * ref.pmf(0)), np.log1p(-pi) + ref.logpmf(x))
  assert np.allclose(zi.log_prob(x), ref_zi, rtol=1e-9, atol=1e-9)
  assert np.isclose(zi.mean()[0], (1 - pi) * ref.mean(), rtol=1e-9)
  assert np.isclose(zi.variance()[0], (1 - pi) * (ref.var() + pi * ref.mean() ** 2), rtol=1e-9)


@SET
@given(mu=st.floats(0.1, 300.0), th=st.floats(0.05, 1e4))
def test_mean_dispersion_form_matches_scipy(mu, th):
  x = np.arange(0, 60, dtype=np.float64)
  nbd = D.NegativeBinomialDisp(np.full(x.shape, mu), np.full(x.shape, th))
  ref = stats.nbinom(th, th / (th + mu))
  # the scVI form carries eps = 1e-8 inside its logarithms: a deviation of about x * eps / mu from the exact pmf
  assert np.allclose(nbd.log_prob(x), ref.logpmf(x), rtol=1e-6, atol=2e-5)
  assert np.isclose(nbd.mean()[0], mu) and np.isclose(nbd.variance()[0], mu + mu * mu / th, rtol=1e-12)


@SET
@given(r=st.floats(0.2, 30.0), l=st.floats(-4.0, 1.0), g=st.floats(-4.0, 4.0))
def test_probabilities_sum_to_one(r, l, g):
  x = np.arange(0, 4000, dtype=np.float64)
  zi = D.ZeroInflated(D.NegativeBinomial(np.full(x.shape, r), np.full(x.shape, l)), np.full(x.shape, g))
  assert abs(np.exp(zi.log_prob(x)).sum() - 1.0) < 1e-6
