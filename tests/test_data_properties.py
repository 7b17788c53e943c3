"""Property tests (hypothesis) of the host-side data functions that mirror sisua/data: the index sets of
`split`, the minibatch order, the label mask and both corruption forms -- the reference's own invariants
(tests/test_datasets.py:61-98) over arbitrary small inputs."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import sisua_oracle as so
from sisua_amd import data

SET = settings(max_examples=40, deadline=None, derandomize=True)


@SET
@given(n=st.integers(2, 400), frac=st.floats(0.05, 0.95), seed=st.integers(0, 2**31 - 1))
def test_split_is_a_partition(n, frac, seed):
  a, b = data.split_indices(n, frac, seed=seed)
  assert len(a) + len(b) == n and len(np.intersect1d(a, b)) == 0
  assert np.array_equal(np.sort(np.concatenate([a, b])), np.arange(n))
  a2, b2 = so.split_indices(n, frac, seed=seed)
  assert np.array_equal(a, a2) and np.array_equal(b, b2)


@SET
@given(n=st.integers(1, 3000), epoch=st.integers(0, 50), shuffle=st.sampled_from([0, 1, 7, 1000]), seed=st.integers(0, 10**6),
       batch=st.integers(1, 300), drop=st.booleans())
def test_epoch_order_is_a_permutation_and_batches_cover_it(n, epoch, shuffle, seed, batch, drop):
  order = data.epoch_order(n, epoch, shuffle, seed)
  assert np.array_equal(np.sort(order), np.arange(n))
  if not shuffle:
    assert np.array_equal(order, np.arange(n))
  bs = data.iter_batches(order, batch, drop)
  flat = np.concatenate(bs) if bs else np.array([], dtype=order.dtype)
  assert all(len(b) == batch for b in bs[:-1]) and (not bs or len(bs[-1]) <= batch)
  assert np.array_equal(flat, order[: len(flat)])
  assert len(flat) == (n // batch * batch if drop else n)


@SET
@given(n=st.integers(1, 500), pct=st.floats(0.0, 1.0), n_omics=st.integers(1, 3), seed=st.integers(0, 10**6))
def test_label_mask_semantics(n, pct, n_omics, seed):
  m = data.label_mask(n, pct, n_omics, seed=seed)
  assert m.shape == (n,) and m.dtype == bool
  if n_omics == 1:
    assert not m.any()                    # a single omic has nothing to be labelled with (_single_cell_base.py:578-579)
  assert np.array_equal(m, data.label_mask(n, pct, n_omics, seed=seed))   # frozen after the first pass (.cache(''))


@SET
@given(seed=st.integers(0, 10**6), rate=st.floats(0.01, 0.99), retain=st.floats(0.0, 1.0), shape=st.tuples(st.integers(1, 40), st.integers(1, 40)))
def test_both_corruption_forms_only_thin_nonzero_counts(seed, rate, retain, shape):
  rng = np.random.default_rng(seed)
  x = rng.poisson(1.5, size=shape).astype(np.float32)
  nnz = np.count_nonzero(x)
  y = data.corrupt(x, rate, retain, seed=seed)
  z, n_sel = so.corrupt_philox(x, rate, retain, seed, np.arange(shape[0]))
  for out in (y, z):
    assert out.shape == x.shape and (out <= x).all() and (out >= 0).all() and (out[x == 0] == 0).all()
    assert np.array_equal(out, np.floor(out))
    assert np.count_nonzero(out != x) <= int(np.floor(rate * nnz))
  assert n_sel in (0, int(np.floor(rate * nnz)))   # 0 only for the reference's early return (both rates degenerate)


@SET
@given(seed=st.integers(0, 10**6), shape=st.tuples(st.integers(2, 60), st.integers(1, 50)))
def test_library_size_matches_its_definition(seed, shape):
  rng = np.random.default_rng(seed)
  x = rng.poisson(2.0, size=shape).astype(np.float32)
  lc, mean, var = data.library_size(x)
  ref = np.log(x.sum(1) + 1e-8)
  assert np.allclose(np.ravel(lc), ref, rtol=1e-6) and np.isclose(mean, ref.mean(), rtol=1e-5, atol=1e-6)
  assert np.isclose(var, ref.var(), rtol=1e-4, atol=1e-6)


def test_generated_scaling_matrix_is_sharding_independent_and_shaped_like_c5():
  """oracle.generate_lognormal_rows (the restatement of smx_dataset_generate_lognormal, BASELINE configs[4]): a row is a
  function of (seed, global cell id) only -- shards of any size concatenate to the same matrix --, counts are non-negative
  integers <= 65535 with ~93 % zeros at the default density, no cell is empty, genes differ in their means."""
  from oracle import sisua_oracle as so
  G = 2000
  a = so.generate_lognormal_rows(8, np.arange(0, 96), G)
  b = np.concatenate([so.generate_lognormal_rows(8, np.arange(r * 32, (r + 1) * 32), G) for r in range(3)])
  assert np.array_equal(a, b) and a.dtype == np.float32 and a.shape == (96, G)
  assert not np.array_equal(a, so.generate_lognormal_rows(9, np.arange(0, 96), G))
  assert np.array_equal(a, np.floor(a)) and a.min() >= 0 and a.max() <= 65535 and (a.sum(1) > 0).all()
  assert 0.91 < (a == 0).mean() < 0.95
  big = so.generate_lognormal_rows(8, np.arange(4000000000, 4000000004, dtype=np.uint64), 64)   # 32-bit cell ids beyond 2^31
  assert big.shape == (4, 64) and np.isfinite(big).all()
  dense = so.generate_lognormal_rows(8, np.arange(512), 40, density=1.0)
  m = np.log1p(dense).mean(0)
  assert m.std() > 0.1                                         # per-gene log-means differ (mu_g = 0.5 n_g)
