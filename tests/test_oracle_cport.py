"""The C / OpenMP fp32 port (the timed CPU baseline) against the float64 NumPy oracle it restates."""
import numpy as np
import pytest

from oracle import sisua_oracle as so
from oracle.cport import CStep
from tests.util import perturbed_params, rel_l2, synth_counts


@pytest.mark.parametrize("lik,bn,units", [("zinb", True, (24, 16)), ("nb", False, (16,)), ("zinbd", True, (16,)), ("nbd", True, (16,))])
def test_cport_matches_oracle(lik, bn, units):
  spec = so.Spec(model="vae", n_genes=50, likelihood=lik, enc_units=units, dec_units=units[::-1], latent_dim=5, batchnorm=bn,
                 input_dropout=0.2, dropout_enc=0.15, dropout_dec=0.15)
  x = synth_counts(64, 50, sparsity=0.7, seed=2, max_count=300)
  params = perturbed_params(spec)
  cs = CStep(spec, params)
  bnst, opt = so.init_bn_state(spec), so.init_opt_state(params)
  for step in range(3):
    rows = np.arange(step * 20, step * 20 + 20)
    ref = so.train_step(spec, params, bnst, opt, x[rows], so.PhiloxNoise(spec.seed, step, rows + 7))
    loss = cs.train_step(x[rows], rows + 7, step)
    assert np.isclose(loss, ref["loss"], rtol=2e-5), (step, loss, ref["loss"])
    if step == 0:
      g = cs.grads()
      top = max(np.linalg.norm(v) for v in ref["grads"].values())
      for n in g:
        assert rel_l2(g[n], ref["grads"][n], floor=1e-3 * top) < 1e-4, n
  p = cs.params()
  for n in p:
    assert np.allclose(p[n], params[n], rtol=1e-4, atol=5e-4), n
