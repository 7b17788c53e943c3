"""Pin the oracle's data-side functions on fixtures produced by executing the
reference's own code (tests/golden/make_reference_fixtures.py).  Bit-exact."""
import os

import numpy as np
import pytest

from oracle import sisua_oracle as so

FX = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_data_fixtures.npz"))


def test_corruption_bit_exact():
  x = FX["x"]
  assert np.array_equal(so.corrupt_binomial(x, 0.2, 0.2, seed=8), FX["corrupt_seed8"])
  assert np.array_equal(so.corrupt_binomial(x, 0.2, 0.2, seed=1), FX["corrupt_seed1"])
  assert np.array_equal(so.corrupt_binomial(x, 0.35, 0.5, seed=8), FX["corrupt_d35_r50_seed8"])
  # corruption only ever lowers counts of non-zero entries (tests/test_datasets.py:80-98)
  c = FX["corrupt_seed8"]
  assert (c <= x).all() and (c[x == 0] == 0).all() and (c != x).sum() > 0


def test_library_size_bit_exact():
  lc, lm, lv = so.library_size(FX["lib_x"])
  assert np.array_equal(lc.reshape(-1, 1).astype(np.float64), FX["lib_log_counts"])
  assert np.all(FX["lib_local_mean"] == lm) and np.all(FX["lib_local_var"] == lv)


def test_split_bit_exact():
  for n, pct, seed in ((100, 0.8, 1), (3005, 0.8, 1), (2404, 0.9, 1), (4697, 0.8, 1), (3757, 0.9, 1), (57, 0.5, 8)):
    tr, te = so.split_indices(n, pct, seed)
    assert np.array_equal(tr, FX[f"split_{n}_{int(pct * 100)}_{seed}_train"])
    assert np.array_equal(te, FX[f"split_{n}_{int(pct * 100)}_{seed}_test"])
  # the C1/C2 cell counts of SURVEY.md section 8 (train.py:66,120)
  assert len(FX["split_2404_90_1_train"]) == 2163 and len(FX["split_3757_90_1_train"]) == 3381


# ---- committed known-answer vectors of the network numerics (tests/golden/make_oracle_fixtures.py) ----
def _step_fixture():
  import importlib.util
  spec_mod = importlib.util.spec_from_file_location("mkfx", os.path.join(os.path.dirname(__file__), "golden", "make_oracle_fixtures.py"))
  mk = importlib.util.module_from_spec(spec_mod)
  spec_mod.loader.exec_module(mk)
  return mk, np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_step_fixture.npz"))


def test_oracle_reproduces_committed_step():
  mk, fx = _step_fixture()
  spec = so.Spec(**mk.STEP_KW)
  names = [n for n, _ in so.manifest(spec)]
  params = {n: fx[f"p0/{n}"].copy() for n in names}
  drop = {int(k.split("/")[1]): fx[k] for k in fx.files if k.startswith("drop/")}
  eps = {so.STREAM_EPS_Z: fx[f"eps/{so.STREAM_EPS_Z}"]}
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  res = so.train_step(spec, params, bn, opt, fx["x"], so.InjectedNoise(drop, eps))
  assert np.isclose(res["loss"], float(fx["loss"]), rtol=1e-12)
  for n in names:
    assert np.allclose(res["grads"][n], fx[f"g/{n}"], rtol=1e-10, atol=1e-14)
    assert np.allclose(params[n], fx[f"p1/{n}"], rtol=1e-12, atol=1e-15)


def test_oracle_reproduces_committed_trajectory():
  mk, _ = _step_fixture()
  fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_trajectory_fixture.npz"))
  spec = so.Spec(**mk.TRAJ_KW)
  params = so.init_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  x, order = fx["x"], fx["order"]
  for s in range(10):
    rows = order[s * 64:(s + 1) * 64]
    r = so.train_step(spec, params, bn, opt, x[rows], so.PhiloxNoise(spec.seed, s, rows))
    assert np.isclose(r["loss"], fx["loss"][s], rtol=1e-10)
  assert fx["loss"][-5:].mean() < fx["loss"][:5].mean()


def test_oracle_reproduces_committed_c2_trajectory():
  """tests/golden/oracle_c2_trajectory.npz (what the GPU test at BASELINE configs[1] is held against) is the oracle's output:
  same matrix, same row order, and the first steps re-derived here agree to float64 rounding."""
  from tests.golden import make_c2_trajectory as mk
  fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_c2_trajectory.npz"))
  cfg, xt, B, order, probe = mk.inputs()
  assert tuple(fx["x_shape"]) == xt.shape and int(fx["x_crc32"]) == mk.checksum(xt)
  assert np.array_equal(order, fx["order"]) and np.array_equal(probe, fx["probe"])
  out = mk.run(n_steps=4)
  for key in ("loss", "nllk_x", "kl"):
    assert np.allclose(out[key], fx[key][:4], rtol=1e-10), key
  assert len(fx["loss"]) == 300 and fx["loss"][-10:].mean() < 0.45 * fx["loss"][:3].mean()
  for at in (100, 300):
    assert fx[f"z_mean_{at}"].shape == (256, cfg.latent_dim) and np.all(fx[f"z_scale_{at}"] > 0)


def test_oracle_reproduces_committed_c5_trajectory():
  """tests/golden/oracle_c5_trajectory.npz (the 20 000-gene trajectory the GPU test is held against) is the oracle's output."""
  from tests.golden import make_c5_trajectory as mk
  fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_c5_trajectory.npz"))
  cfg, xt, B, order, probe = mk.inputs()
  assert tuple(fx["x_shape"]) == xt.shape and int(fx["x_crc32"]) == mk.checksum(xt)
  out = mk.run(n_steps=2)
  for key in ("loss", "nllk_x", "kl"):
    assert np.allclose(out[key], fx[key][:2], rtol=1e-10), key
  assert len(fx["loss"]) == mk.STEPS and fx["loss"][-3:].mean() < 0.8 * fx["loss"][:3].mean()
  assert fx["z_mean"].shape == (128, cfg.latent_dim) and np.all(fx["z_scale"] > 0)


def _variants_fixture():
  import importlib.util
  spec_mod = importlib.util.spec_from_file_location("mkvar", os.path.join(os.path.dirname(__file__), "golden", "make_head_fixtures.py"))
  mk = importlib.util.module_from_spec(spec_mod)
  spec_mod.loader.exec_module(mk)
  return mk, np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_variants_fixture.npz"))


@pytest.mark.parametrize("name", ["misa_tril", "scale_tril"])
def test_oracle_reproduces_committed_variant_steps(name):
  """tests/golden/oracle_variants_fixture.npz (MISA's 'mixtril' + zero-inflated heads, SCALE with full-covariance components): the
  committed inputs are what the generator builds, and the oracle gives the committed loss terms and gradients again."""
  mk, fx = _variants_fixture()
  spec, x, ys, mask, rows = mk.inputs(name)
  assert np.array_equal(x, fx[f"{name}/x"]) and np.array_equal(mask, fx[f"{name}/mask"])
  names = [n for n, _ in so.manifest(spec)]
  params = {n: fx[f"{name}/p0/{n}"].copy() for n in names}
  res = so.forward_backward(spec, params, so.init_bn_state(spec), x[rows], so.PhiloxNoise(spec.seed, mk.STEP, rows + mk.CELL_BASE),
                            y=[fx[f"{name}/y{j}"][rows] for j in range(len(spec.labels))], mask=mask[rows])
  for k in ("loss", "nllk_x", "nllk_y", "kl"):
    assert np.isclose(res["metrics"][k], float(fx[f"{name}/{k}"]), rtol=1e-12, atol=1e-14), k
  for n in names:
    assert np.allclose(res["grads"][n], fx[f"{name}/g/{n}"], rtol=1e-10, atol=1e-14), n
