"""preprocessing of the HBM-resident matrix on the GPU (library statistics, artificial
corruption) against the oracle; all calls go through the C-ABI."""
import numpy as np
import pytest

from oracle import sisua_oracle as so
from tests.util import grad_errors, make_pair, synth_counts

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine():
  from sisua_amd.engine import Engine as E
  return E


def _engine(Engine, x, base=0, model="vae", likelihood="zinb"):
  kw = dict(model=model, n_genes=x.shape[1], likelihood=likelihood, enc_units=(16,), dec_units=(16,), latent_dim=4)
  if model == "scvi":
    kw["encl_units"] = (8,)
  spec, cfg = make_pair(**kw)
  e = Engine(cfg, max_batch=32)
  lib = np.zeros((len(x), 2), np.float32) + 1.0 if model == "scvi" else None
  e.upload(x, library=lib, cell_id_base=base)
  return spec, e


@pytest.mark.parametrize("shape,sparsity", [((300, 203), 0.85), ((64, 31), 0.3), ((1000, 1998), 0.93)])
def test_library_statistics_match_reference_formula(Engine, shape, sparsity):
  x = synth_counts(shape[0], shape[1], sparsity=sparsity, seed=2)
  x[3] = 0.0     # an empty cell: log(0 + 1e-8)
  _, e = _engine(Engine, x)
  mean, var = e.dataset_library()
  _, rm, rv = so.library_size(x)     # pinned against the reference by tests/golden/reference_data_fixtures.npz
  assert np.isclose(mean, rm, rtol=1e-5) and np.isclose(var, rv, rtol=1e-5), (mean, rm, var, rv)
  X, rc, lib = e.dataset_read(library=True)
  assert np.array_equal(X, x)
  assert np.allclose(lib, np.array([[rm, rv]], np.float32), rtol=1e-5)
  from scipy.special import gammaln
  assert np.allclose(rc, gammaln(x.astype(np.float64) + 1.0).sum(1), rtol=1e-6, atol=1e-4)
  e.close()


@pytest.mark.parametrize("dropout,retain,seed,base", [(0.2, 0.2, 8, 0), (0.5, 0.7, 12345678901, 700), (0.0, 0.5, 3, 0),
                                                    (0.999, 0.0, 1, 5)])
def test_corruption_bit_exact_vs_oracle(Engine, dropout, retain, seed, base):
  x = synth_counts(257, 203, sparsity=0.8, seed=4, max_count=3000)
  x[5, 7] = 4097.0   # a long binomial trial loop
  _, e = _engine(Engine, x, base=base)
  n = e.dataset_corrupt(dropout, retain, seed)
  ref, n_ref = so.corrupt_philox(x, dropout, retain, seed, np.arange(len(x)) + base)
  X, rc = e.dataset_read()
  assert n == n_ref == int(np.floor(dropout * np.count_nonzero(x)))
  assert np.array_equal(X, ref)
  from scipy.special import gammaln
  assert np.allclose(rc, gammaln(ref.astype(np.float64) + 1.0).sum(1), rtol=1e-6, atol=1e-4)
  e.close()


def test_corruption_properties_at_8kly_size(Engine):
  """Size-independent properties (the reference's own checks, tests/test_datasets.py:80-98): counts only
  decrease, zeros stay zero, exactly floor(rate * nnz) entries are touched, thinning keeps ~retain_rate of the
  selected mass, idempotent arguments are no-ops, and training still runs on the corrupted matrix."""
  x = synth_counts(3381, 1998, sparsity=0.93, seed=8)
  spec, e = _engine(Engine, x)
  nnz = np.count_nonzero(x)
  assert e.dataset_corrupt(0.0, 1.0, 8) == 0 and e.dataset_corrupt(0.0, 0.0, 8) == 0
  assert np.array_equal(e.dataset_read()[0], x)
  n = e.dataset_corrupt(0.2, 0.2, 8)
  X, _ = e.dataset_read()
  assert n == int(np.floor(0.2 * nnz))
  assert (X <= x).all() and (X[x == 0] == 0).all() and (X == np.floor(X)).all()
  changed = X != x
  assert changed.sum() <= n and changed.sum() > 0.7 * n          # Binomial(n, .2) == n only for small n
  # E[ sum of selected after ] = 0.2 * sum of selected before; selected mass ~ 0.2 of total (uniform choice)
  lost = x.sum() - X.sum()
  assert abs(lost / (0.2 * 0.8 * x.sum()) - 1.0) < 0.05
  with pytest.raises(RuntimeError):
    e.dataset_corrupt(1.0, 0.2, 8)                                # utils.py:184-185
  mean, var = e.dataset_library()
  _, rm, rv = so.library_size(X)
  assert np.isclose(mean, rm, rtol=1e-5) and np.isclose(var, rv, rtol=1e-4)
  loss = e.train_step(np.arange(32, dtype=np.int32))["loss"]
  assert np.isfinite(loss)
  e.close()


def test_library_feeds_scvi_prior(Engine):
  """SCVI reads the resident library prior (scvi.py:100-105): after smx_dataset_library the KL of the
  library latent must match the oracle evaluated with the recomputed statistics."""
  x = synth_counts(200, 96, sparsity=0.7, seed=6)
  spec, e = _engine(Engine, x, model="scvi", likelihood="zinbd")
  mean, var = e.dataset_library()
  params = e.get_params()
  lib = np.tile(np.array([[mean, var]], np.float32), (len(x), 1))
  rows = np.arange(32, dtype=np.int32)
  got = e.eval_step(rows)
  ref = so.forward_backward(spec, {k: v.astype(np.float64) for k, v in params.items()}, so.init_bn_state(spec), x[rows],
                            so.PhiloxNoise(spec.seed, 0, rows), library=lib[rows].astype(np.float64), training=False,
                            backward=False)
  assert np.isclose(got["kl_l"], ref["kl_l"].mean(), rtol=1e-4, atol=1e-5), (got["kl_l"], ref["kl_l"].mean())
  e.close()


@pytest.mark.parametrize("model,likelihood", [("vae", "zinb"), ("scvi", "zinbd"), ("vae", "nb")])
def test_u16_store_is_bit_identical_to_f32(Engine, model, likelihood):
  """SURVEY 8f-2 compact count format: the uint16 resident store must give exactly the float32 results --
  training trajectory, parameters, scoring, corruption, library statistics."""
  x = synth_counts(400, 203, sparsity=0.8, seed=9, max_count=60000)
  kw = dict(model=model, n_genes=203, likelihood=likelihood, enc_units=(32,), dec_units=(32,), latent_dim=6, input_dropout=0.2)
  if model == "scvi":
    kw["encl_units"] = (16,)
  spec, cfg = make_pair(**kw)
  _, lm, lv = so.library_size(x)
  lib = np.tile(np.array([[lm, lv]], np.float32), (len(x), 1)) if model == "scvi" else None
  outs = []
  for storage in ("f32", "u16"):
    e = Engine(cfg, max_batch=64)
    e.upload(x, library=lib, cell_id_base=11, storage=storage)
    order = (np.arange(64 * 5) * 7 % len(x)).astype(np.int32)
    losses = [e.train_step(order[s * 64:(s + 1) * 64])["loss"] for s in range(5)]
    rows = np.arange(40, dtype=np.int32)
    mllk, _ = e.marginal_llk(row_ids=rows, n_samples=3)
    n_cor = e.dataset_corrupt(0.3, 0.4, 5)
    stats = e.dataset_library()
    X, rc, lb = e.dataset_read(library=True)
    after = e.train_step(order[:64])["loss"]
    outs.append((losses, e.get_params(), mllk, n_cor, stats, X, rc, after))
    e.close()
  a, b = outs
  assert a[0] == b[0] and a[7] == b[7]
  for k in a[1]:
    assert np.array_equal(a[1][k], b[1][k]), k
  assert np.array_equal(a[2], b[2]) and a[3] == b[3] and a[4] == b[4]
  assert np.array_equal(a[5], b[5]) and np.array_equal(a[6], b[6])


def test_u16_store_rejects_what_it_cannot_hold(Engine):
  x = synth_counts(50, 64, sparsity=0.5, seed=1)
  _, e = _engine(Engine, x)
  for bad in (x + 0.5, np.where(x > 0, 70000.0, 0.0).astype(np.float32), -x - 1):
    with pytest.raises(ValueError):
      e.upload(bad, storage="u16")
  with pytest.raises(ValueError):
    e.upload(x, storage="bf16")
  e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("model,likelihood,graph", [("vae", "zinb", False), ("vae", "nb", True), ("scvi", "zinbd", False), ("sisua", "zinb", False)])
def test_csr_store_is_bit_identical_to_f32(Engine, model, likelihood, graph):
  """SURVEY 8f-2 compact count format, sparse: the CSR resident store (non-zeros only; every pass expands its
  minibatch's rows into a dense tile first) must give exactly the float32 results -- training trajectory, parameters,
  evaluation, forward pass, scoring, and the rows read back; the resident-matrix kernels refuse it."""
  x = synth_counts(400, 203, sparsity=0.9, seed=9, max_count=60000)
  x[7] = 0.0                                   # an empty row
  kw = dict(model=model, n_genes=203, likelihood=likelihood, enc_units=(32,), dec_units=(32,), latent_dim=6)
  ys, mask = [], None
  if model == "scvi":
    kw["encl_units"] = (16,)
  if model == "sisua":
    kw["labels"] = ((9, "nb"),)
    ys = [np.random.default_rng(3).poisson(3.0, size=(len(x), 9)).astype(np.float32)]
    mask = (np.arange(len(x)) % 3 == 0)
  spec, cfg = make_pair(**kw)
  _, lm, lv = so.library_size(x)
  lib = np.tile(np.array([[lm, lv]], np.float32), (len(x), 1)) if model == "scvi" else None
  outs = []
  for storage in ("f32", "csr"):
    e = Engine(cfg, max_batch=64)
    e.upload(x, ys, library=lib, label_mask=mask, cell_id_base=11, storage=storage)
    order = (np.arange(64 * 5) * 7 % len(x)).astype(np.int32)
    losses = [e.train_step(order[s * 64:(s + 1) * 64], graph=graph)["loss"] for s in range(5)]
    losses += [e.train_steps(order[:128], 2, 64, graph=graph, metrics=True)["loss"], e.train_step(order[:50])["loss"]]
    rows = np.arange(40, dtype=np.int32)
    ev = e.eval_step(rows)["loss"]
    fw = e.forward(row_ids=rows)
    mllk, _ = e.marginal_llk(row_ids=rows, n_samples=3)
    X, rc = e.dataset_read(library=False)
    outs.append((losses, e.get_params(), ev, fw["x_params"], fw["z_mean"], mllk, X, rc))
    if storage == "csr":
      for fn in (lambda: e.dataset_corrupt(0.3, 0.4, 5), e.dataset_library):
        with pytest.raises(Exception):
          fn()
    e.close()
  a, b = outs
  assert a[0] == b[0] and a[2] == b[2]
  for k in a[1]:
    assert np.array_equal(a[1][k], b[1][k]), k
  for i in (3, 4, 5, 6):
    assert np.array_equal(a[i], b[i]), i
  assert np.allclose(a[7], b[7], rtol=1e-6) and np.array_equal(b[6], x)


@pytest.mark.gpu
@pytest.mark.parametrize("G,B", [(4500, 128), (4200, 77)])
def test_csr_store_at_a_wide_gene_panel(Engine, G, B):
  """... and at a gene panel wide enough for the panel kernels (the encoder's weight gradient as smx_panel.h's role 0, the one-launch output
  head, the wide BatchNorm launches).  Round 6 found the encoder's weight-gradient launch reading row ids the sparse store's dense tile does
  not have (a GPU memory fault at address 0 on the first training step): a null row-id pointer now means `cell c is row c` there as in every
  other kernel.  Same bits as the float32 store: the losses of multi-step and single-step calls, every parameter, evaluation, scoring."""
  x = synth_counts(400, G, sparsity=0.9, seed=G, max_count=900)
  spec, cfg = make_pair(model="vae", n_genes=G, likelihood="zinb", enc_units=(128,), dec_units=(128,), latent_dim=16)
  order = (np.arange(B * 6) * 7 % len(x)).astype(np.int32)
  outs = []
  for storage in ("f32", "csr"):
    e = Engine(cfg, max_batch=128, init=False)
    e.set_params(so.init_params(spec))
    e.upload(x, cell_id_base=11, storage=storage)
    assert e.head_fused_bytes(B) > 0
    e.train_steps(order[: 4 * B], 4, B, graph=False)
    h = {k: np.asarray(v).copy() for k, v in e.metrics_history(4).items()}
    one = e.train_step(order[4 * B: 5 * B])["loss"]
    cap = e.train_step(order[5 * B: 6 * B], graph=True)["loss"]
    rows = np.arange(40, dtype=np.int32)
    ev = e.eval_step(rows)["loss"]
    mllk, _ = e.marginal_llk(row_ids=rows, n_samples=3)
    outs.append((h, (one, cap, ev), mllk, e.get_params()))
    e.close()
  a, b = outs
  for k in a[0]:
    assert np.array_equal(a[0][k], b[0][k]), k
  assert a[1] == b[1] and np.array_equal(a[2], b[2])
  for k in a[3]:
    assert np.array_equal(a[3][k], b[3][k]), k
  assert np.isfinite(a[0]["loss"]).all()


@pytest.mark.gpu
def test_csr_store_takes_sparse_inputs_and_rejects_bad_ones(Engine):
  import scipy.sparse as sp
  x = synth_counts(60, 64, sparsity=0.85, seed=1)
  _, e = _engine(Engine, x)
  m = sp.csr_matrix(x)
  for X in (m, (m.indptr, m.indices, m.data), sp.coo_matrix(x)):
    e.upload(X, storage="csr")
    got, _ = e.dataset_read(library=False)
    assert np.array_equal(got, x)
  bad_cols = m.indices.copy(); bad_cols[0] = 64
  for X in ((m.indptr, bad_cols, m.data), (m.indptr[:-1], m.indices, m.data), (m.indptr[::-1].copy(), m.indices, m.data)):
    with pytest.raises(Exception):
      e.upload(X, storage="csr")
  # input dropout is keyed by the dense store's rows: refused, not silently different
  spec, cfg = make_pair(model="vae", n_genes=64, likelihood="nb", enc_units=(32,), dec_units=(32,), latent_dim=4, input_dropout=0.3)
  e2 = Engine(cfg, max_batch=32)
  e2.upload(x, storage="csr")
  with pytest.raises(Exception):
    e2.train_step(np.arange(32, dtype=np.int32))
  e2.close(); e.close()


def _rows_match_generator(got, cell_ids, seed, density, G):
  """Device rows vs oracle.generate_lognormal_rows: identical except where the un-floored value sits within float32
  evaluation error of an integer (the device's v_log / v_sin / v_cos / v_exp forms: ~1e-6 relative on exp(.)), and there by
  exactly one count.  Returns the number of such entries."""
  ref, real = so.generate_lognormal_rows(seed, cell_ids, G, density, return_real=True)
  diff = got != ref
  if diff.any():
    near = np.abs(real - np.round(real)) <= 2e-5 * np.maximum(real, 1.0)
    assert (near | ~diff).all(), "entries differ away from an integer boundary"
    assert np.abs(got - ref)[diff].max() == 1.0
    assert diff.mean() < 1e-4, diff.mean()
  return int(diff.sum())


@pytest.mark.parametrize("storage,G,rank", [("u16", 203, 2), ("f32", 64, 0), ("u16", 1998, 7)])
def test_generator_matches_oracle(Engine, storage, G, rank):
  """smx_dataset_generate_lognormal (BASELINE configs[4]'s matrix, generated on the device from (seed, rank)) against the
  oracle's restatement: rows, the per-row likelihood constants, padded columns, shard independence."""
  from scipy.special import gammaln
  spec, cfg = make_pair(model="vae", n_genes=G, likelihood="zinb", enc_units=(32,), dec_units=(32,), latent_dim=8)
  e = Engine(cfg, max_batch=64)
  n = 300
  e.generate_lognormal(n, seed=8, rank=rank, storage=storage, density=0.14)
  assert e.n_cells == n
  x, lg = e.dataset_read(0, n)
  ids = np.arange(rank * n, (rank + 1) * n)
  _rows_match_generator(x, ids, 8, 0.14, G)
  assert np.allclose(lg, gammaln(x.astype(np.float64) + 1.0).sum(1), rtol=1e-6, atol=1e-4)
  assert 0.90 < (x == 0).mean() < 0.96 and (x[:, 0] >= 1).all()
  # the same cells as part of a larger shard of another job: rank 0 of a shard three times the size holds them too
  if rank == 2:
    e2 = Engine(cfg, max_batch=64)
    e2.generate_lognormal(3 * n, seed=8, rank=0, storage=storage, density=0.14)
    x2, _ = e2.dataset_read(2 * n, n)
    assert np.array_equal(x, x2)
    e2.close()
  # a training step runs on it and matches the oracle on the device's own rows (noise keyed by the global cell ids)
  params = so.init_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  e.set_params(params)
  rows = np.arange(100, 164, dtype=np.int32)
  res = so.train_step(spec, params, bn, opt, x[rows], so.PhiloxNoise(spec.seed, 0, rows + rank * n))
  m = e.train_step(rows)
  assert np.isclose(m["loss"], res["metrics"]["loss"], rtol=1e-4)
  e.close()


def test_c5_at_full_residency(Engine):
  """BASELINE configs[4] at its real residency on ONE GPU: 1e6 cells x 20 000 genes generated on the device as uint16
  (40 GB resident, 2e10 elements: every row offset beyond 2^31 elements must be 64-bit arithmetic).  Rows are drawn from
  the LAST 1 % of the matrix (element offsets > 1.98e10): read-back against the oracle's generator, the per-row likelihood
  constants, three optimiser steps (ELBO scalars; every gradient at the first) against the oracle on those rows, and the
  library statistics over all 1e6 rows."""
  from scipy.special import gammaln
  N, G, B = 1_000_000, 20000, 128
  spec, cfg = make_pair(model="vae", n_genes=G, likelihood="zinb", enc_units=(128,), dec_units=(128,), latent_dim=32)
  e = Engine(cfg, max_batch=B, init=False)
  params = so.init_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  e.set_params(params)
  e.generate_lognormal(N, seed=8, rank=0, storage="u16", density=0.14)
  assert e.n_cells == N
  rng = np.random.default_rng(3)
  tail0 = N - 10_000
  assert tail0 * 20000 > 2 ** 31 * 9                            # far beyond 32-bit element offsets
  # read-back of a block at the very end and of scattered tail rows
  xe, lge = e.dataset_read(N - 64, 64)
  _rows_match_generator(xe, np.arange(N - 64, N), 8, 0.14, G)
  assert np.allclose(lge, gammaln(xe.astype(np.float64) + 1.0).sum(1), rtol=1e-6, atol=1e-3)
  for s in range(3):
    rows = np.sort(rng.choice(np.arange(tail0, N), size=B, replace=False)).astype(np.int32)
    xr = np.concatenate([e.dataset_read(int(r), 1)[0] for r in rows])   # the device's own rows (boundary ties aside, the oracle's)
    if s == 0:
      _rows_match_generator(xr, rows, 8, 0.14, G)
    res = so.train_step(spec, params, bn, opt, xr, so.PhiloxNoise(spec.seed, s, rows))
    m = e.train_step(rows)
    assert m["nan_flag"] == 0
    for key in ("loss", "nllk_x", "kl"):
      assert np.isclose(m[key], res["metrics"][key], rtol=1e-4, atol=1e-5), (s, key, m[key], res["metrics"][key])
    if s == 0:
      worst = grad_errors(e.get_params(which=1), res["grads"])
      assert max(worst.values()) < 1e-4, sorted(worst.items(), key=lambda kv: -kv[1])[:3]
      assert np.isclose(m["grad_norm_max"], max(np.linalg.norm(g) for g in res["grads"].values()), rtol=1e-4)
  # library statistics over the whole resident matrix (get_library_size, sisua/data/utils.py:231-263): against the oracle's
  # generator on a random sample of cells (standard error of the sample mean ~ sd / sqrt(4000))
  lm, lv = e.dataset_library()
  sample = rng.choice(N, size=4000, replace=False)
  lc = np.log(so.generate_lognormal_rows(8, sample, G).sum(1) + 1e-8)
  assert abs(lm - lc.mean()) < 5 * lc.std() / np.sqrt(4000) + 1e-3 and abs(lv - lc.var()) < 0.25 * lc.var()
  lib_tail = e.dataset_read(N - 4, 4, library=True)[-1]
  assert np.allclose(lib_tail, [[lm, lv]] * 4)
  e.close()
