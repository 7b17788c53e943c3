"""Pin the oracle's data-side functions on fixtures produced by executing the
reference's own code (tests/golden/make_reference_fixtures.py).  Bit-exact."""
import os

import numpy as np

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
