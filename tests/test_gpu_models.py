"""The reference's model-level test strategy (tests/test_singlecell_models.py:28-32, 93-188;
tests/test_save_load_model.py:198-201) on the HIP path: registry, fit/predict for DCA, VAE,
SISUA, SCVI with 'loss decreases', output distribution classes / shapes, save -> load."""
import os

import numpy as np
import pytest

from tests.util import synth_counts, synth_labels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
  from sisua_amd import build
  build.build(verbose=False)
  import sisua_amd.models as M
  return M


def _sco(n=600, g=120, with_labels=True):
  from sisua_amd.data import SingleCellOMIC
  sco = SingleCellOMIC(synth_counts(n, g, sparsity=0.8, seed=3), name="toy")
  if with_labels:
    sco.add_omic("proteomic", synth_labels(n, ((9, "nb"),))[0])
  return sco


def _decreases(hist):
  """`ModelTest._loss_not_rise` of the reference (tests/test_singlecell_models.py:28-32) on per-epoch means: skip the
  first epoch, the loss must fall in more than 80 % of the remaining epoch-to-epoch moves."""
  loss = list(hist)[1:]
  falls = [i > j for i, j in zip(loss, loss[1:])]
  return np.sum(falls) > 0.8 * (len(loss) - 1)


def test_registry(api):
  ids = [m.id for m in api.get_all_models()]
  assert {"dca", "vae", "sisua", "scvi"} <= set(ids)
  assert api.get_model("vae") is api.VAE and api.get_model("DeepCountAutoencoder") is api.DeepCountAutoencoder
  assert api.get_model(api.SCVI) is api.SCVI
  with pytest.raises(RuntimeError):
    api.get_model("nope")


@pytest.mark.parametrize("name", ["dca", "vae", "sisua", "scvi"])
def test_fit_predict(api, name):
  from sisua_amd import distributions as D
  sco = _sco()
  train, test = sco.split(0.8)
  cls = api.get_model(name)
  rna = sco.get_rv("transcriptomic", "zinbd" if name == "scvi" else "zinb")
  kw = dict(outputs=rna, latents=api.RVmeta(8, "diag" if name != "dca" else "relu", True, "Latents"),
            encoder=api.NetConf([32], batchnorm=True, dropout=0.1), decoder=api.NetConf([32], batchnorm=True, dropout=0.1))
  if name == "sisua":
    kw["labels"] = [sco.get_rv("proteomic")]
  if name == "dca":
    kw.pop("latents")
  model = cls(**kw)
  assert not model.is_fitted and model.is_zero_inflated and model.is_semi_supervised == (name == "sisua")
  with pytest.raises(RuntimeError):   # fit without metadata (single_cell_model.py:227-231)
    cls(**kw).fit(train.create_dataset(batch_size=32))
  omics = ["transcriptomic"] + (["proteomic"] if name == "sisua" else [])
  ds = train.create_dataset(omics, labels_percent=0.5, batch_size=64, drop_remainder=True)
  vs = test.create_dataset(omics, labels_percent=1.0, batch_size=64, drop_remainder=True)
  model.fit(ds, valid=vs, metadata=sco, epochs=12, valid_freq=20, learning_rate=2e-3)
  assert model.is_fitted and model.dataset == "toy" and "transcriptomic" in model.metadata
  assert len(model.train_history["loss"]) == 12            # one value per epoch (mean over its steps)
  if name == "sisua":
    # the toy protein levels are pure noise and only half of the cells are labelled: the alpha = 10 label term moves
    # with WHICH cells a batch holds (8 of 10 moves fall here, the criterion asks for more than 8); the criterion is
    # applied to the transcriptomic term, the total must still fall from the first to the last quarter
    h = np.asarray(model.train_history["loss"])
    assert _decreases(model.train_history["nllk_x"]) and h[-3:].mean() < h[:3].mean(), (h, model.train_history["nllk_x"])
  else:
    assert _decreases(model.train_history["loss"]), model.train_history["loss"]
  assert len(model.valid_history["val_loss"]) >= 1
  X, Z = model.predict(test.create_dataset(omics, batch_size=50, shuffle=0), verbose=False)
  n = test.n_obs
  Xs = X if isinstance(X, tuple) else (X,)
  assert isinstance(Xs[0], D.Independent) and isinstance(Xs[0].distribution, D.ZeroInflated)
  assert Xs[0].batch_shape == (n,) and Xs[0].event_shape == (120,) and Xs[0].name == "transcriptomic"
  assert np.isfinite(Xs[0].mean()).all() and (Xs[0].mean() >= 0).all()
  if name == "sisua":
    assert len(Xs) == 2 and Xs[1].event_shape == (9,) and Xs[1].name == "proteomic"
  Zs = Z if isinstance(Z, tuple) else (Z,)
  dz = 10 if name == "dca" else 8
  assert Zs[0].mean().shape == (n, dz)
  if name == "scvi":
    assert len(Zs) == 2 and Zs[1].mean().shape == (n, 1)
  if name == "dca":
    assert isinstance(Zs[0], D.Deterministic) and (Zs[0].mean() >= 0).all()
  # Monte-Carlo axis (posterior.py:175-182 uses sample_shape=10)
  X3, _ = model.predict(test.numpy()[:20], sample_shape=3, batch_size=8, verbose=False)
  X3 = X3[0] if isinstance(X3, tuple) else X3
  assert X3.batch_shape == (3, 20) and X3.mean().shape == (3, 20, 120)
  # the one-call path (smx_predict: batch loop in the library, results written once into their final arrays) holds the
  # same numbers as minibatch-by-minibatch calls, ragged last batch included
  from sisua_amd.data import library_matrix
  xs = test.numpy()[:50]
  lib = library_matrix(xs)
  for S in ((), 2):
    Xa, Za = model.predict(xs, sample_shape=S, batch_size=16, verbose=False)
    Xa0 = Xa[0] if isinstance(Xa, tuple) else Xa
    Za0 = Za[0] if isinstance(Za, (tuple, list)) else Za
    for i in range(0, 50, 16):
      pX, qZ = model(inputs=xs[i:i + 16], library=lib[i:i + 16], sample_shape=S)
      pX0 = pX[0] if isinstance(pX, tuple) else pX
      qZ0 = qZ[0] if isinstance(qZ, (tuple, list)) else qZ
      if S == ():
        assert np.array_equal(pX0.mean(), Xa0.mean()[..., i:i + 16, :])
      else:   # several draws: the one-call path decodes them as rows of one pass (same draws, other kernels: rounding)
        assert np.allclose(pX0.mean(), Xa0.mean()[..., i:i + 16, :], rtol=2e-5, atol=1e-6)
      assert np.array_equal(qZ0.mean(), Za0.mean()[i:i + 16])
  # ... bitwise with the stacked decode switched off (every draw then runs the same launches in both paths)
  model._engine.set_flag("stacked_scoring", False)
  Xl, _ = model.predict(xs, sample_shape=2, batch_size=16, verbose=False)
  Xl0 = Xl[0] if isinstance(Xl, tuple) else Xl
  pX, _ = model(inputs=xs[:16], library=lib[:16], sample_shape=2)
  assert np.array_equal((pX[0] if isinstance(pX, tuple) else pX).mean(), Xl0.mean()[..., :16, :])
  assert np.allclose(Xl0.mean(), Xa0.mean(), rtol=2e-5, atol=1e-6)
  model._engine.set_flag("stacked_scoring", True)
  # ... and when the result leaves the device in several chunks (staging forced small: 16 cells per chunk)
  from sisua_amd import _hip
  _hip.set_tuning("predict_stage_floats", 20000)
  try:
    Xb, Zb = model.predict(xs, sample_shape=2, batch_size=16, verbose=False)
  finally:
    _hip.clear_tuning("predict_stage_floats")
  Xb0 = Xb[0] if isinstance(Xb, tuple) else Xb
  Zb0 = Zb[0] if isinstance(Zb, (tuple, list)) else Zb
  assert np.array_equal(Xb0.mean(), Xa0.mean()) and np.array_equal(Zb0.mean(), Za0.mean())
  if isinstance(Xb, tuple):
    assert np.array_equal(Xb[1].mean(), Xa[1].mean())
  # encode / decode round trip agrees with __call__
  q = model.encode(test.numpy()[:16])
  q0 = q[0] if isinstance(q, list) else q
  assert np.allclose(q0.mean(), Zs[0].mean()[:16], atol=1e-5)
  lat = [d.mean() for d in q] if isinstance(q, list) else q0.mean()
  pX = model.decode(lat)
  pX = pX[0] if isinstance(pX, tuple) else pX
  assert pX.mean().shape == (16, 120)
  mllk, llk = model.marginal_log_prob(inputs=test.numpy()[:16], sample_shape=8)
  assert mllk.shape == (16,) and np.isfinite(mllk).all() and "transcriptomic" in llk
  assert (mllk >= llk["transcriptomic"] - 50).all()
  sc = model.posterior_llk(test.numpy()[:16], original=test.numpy()[:16] + 1.0, sample_shape=4)
  assert set(sc) == {f"llk_transcriptomic_{a}_{b}" for a in ("imp", "rec") for b in ("org", "cor")}
  assert all(np.isfinite(v) for v in sc.values())


@pytest.mark.parametrize("name,genes", [("vae", 120), ("sisua", 4200), ("scvi", 4100), ("dca", 300)])
def test_fit_on_each_resident_store_is_the_same_fit(api, name, genes):
  """`fit(..., storage=)`: the compact stores of SURVEY 8 f-2 -- uint16 counts, CSR -- hold the same cells as the reference's dense float32
  matrix, so a fit on any of them IS the same fit: per-epoch training history, validation losses, parameters and the predictions of the
  fitted model, bit for bit -- at a narrow panel and at panels wide enough for the one-launch output head and the panel kernels (where round
  6 found the CSR store faulting on its first step), validation cells resident behind the training cells, a ragged last batch."""
  from sisua_amd.data import SingleCellOMIC
  n = 500
  sco = SingleCellOMIC(synth_counts(n, genes, sparsity=0.9, seed=genes, max_count=900), name="toy")
  sco.add_omic("proteomic", synth_labels(n, ((9, "nb"),))[0])
  train, test = sco.split(0.8)
  cls = api.get_model(name)
  omics = ["transcriptomic"] + (["proteomic"] if name == "sisua" else [])
  runs = []
  for storage in ("f32", "u16", "csr"):
    kw = dict(outputs=sco.get_rv("transcriptomic", "zinbd" if name == "scvi" else "zinb"), encoder=api.NetConf([128], batchnorm=True, dropout=0.1),
              decoder=api.NetConf([128], batchnorm=True, dropout=0.1))
    if name != "dca":
      kw["latents"] = api.RVmeta(10, "diag", True, "Latents")
    if name == "sisua":
      kw["labels"] = [sco.get_rv("proteomic")]
    model = cls(**kw)
    ds = train.create_dataset(omics, labels_percent=0.5, batch_size=96, drop_remainder=False)
    vs = test.create_dataset(omics, labels_percent=1.0, batch_size=50, drop_remainder=False)
    model.fit(ds, valid=vs, metadata=sco, epochs=3, valid_freq=4, storage=storage)
    X, Z = model.predict(test.numpy()[:40], batch_size=16, verbose=False)
    X0 = X[0] if isinstance(X, tuple) else X
    Z0 = Z[0] if isinstance(Z, (tuple, list)) else Z
    runs.append(({k: np.asarray(v) for k, v in model.train_history.items()}, np.asarray(model.valid_history["val_loss"]), model._engine.get_params(),
                 X0.mean(), Z0.mean()))
  ref = runs[0]
  assert len(ref[0]["loss"]) == 3 and len(ref[1]) >= 2 and np.isfinite(ref[0]["loss"]).all()
  for r in runs[1:]:
    for k in ref[0]:
      assert np.array_equal(ref[0][k], r[0][k]), k
    assert np.array_equal(ref[1], r[1])
    for k in ref[2]:
      assert np.array_equal(ref[2][k], r[2][k]), k
    assert np.array_equal(ref[3], r[3]) and np.array_equal(ref[4], r[4])


def test_save_load_roundtrip(api, tmp_path):
  sco = _sco(with_labels=False)
  train, test = sco.split(0.8)
  kw = dict(outputs=sco.get_rv("transcriptomic"), latents=api.RVmeta(6, "diag", True, "Latents"),
            encoder=api.NetConf([32, 16], batchnorm=True, dropout=0.1), decoder=api.NetConf([16], batchnorm=True))
  m1 = api.VAE(**kw)
  path = os.path.join(tmp_path, "model")
  m1.fit(train, epochs=4, batch_size=64, checkpoint=lambda: m1.save_weights(path))
  _, z1 = m1.predict(test.numpy(), batch_size=64, verbose=False)
  m1.save_weights(path)
  m2 = api.load_model(path)
  assert type(m2) is api.VAE and m2.dataset == "toy" and m2.step == m1.step
  _, z2 = m2.predict(test.numpy(), batch_size=64, verbose=False)
  assert np.allclose(z1.mean(), z2.mean()) and np.allclose(z1.variance(), z2.variance())   # test_save_load_model.py:198-201
  # resume: both continue identically (optimizer state + BN stats + step restored)
  ds = train.create_dataset(batch_size=64, drop_remainder=True)
  m1.fit(ds, epochs=1)
  m2.fit(ds, epochs=1, metadata=sco)
  assert np.allclose(m1.train_history["loss"][-1], m2.train_history["loss"][-1], rtol=1e-6)
  m3 = api.VAE(**kw).load_weights(os.path.join(tmp_path, "missing"))
  assert not m3.is_fitted
  with pytest.raises(FileNotFoundError):
    api.VAE(**kw).load_weights(os.path.join(tmp_path, "missing"), raise_notfound=True)


def test_experiment_driver_cortex_plumbing(api):
  """BASELINE.json configs[0]: cortex, VAE, batch 32, 5 epochs (335 steps): loss finite and falling."""
  from sisua_amd.train import Experiment
  exp = Experiment(dict(model=dict(name="vae"), dataset=dict(name="cortex", batch_size=32), train=dict(epochs=5)))
  model = exp.run()
  assert model.step == 335 and np.isfinite(model.train_history["loss"]).all()
  assert model.train_history["loss"][-1] < model.train_history["loss"][0]


def test_fit_trains_on_the_ragged_last_batch_and_reports_nan(api):
  """fit(SingleCellOMIC) batches with drop_remainder=False (the reference's default): the last, smaller batch is a
  step of its own; terminate_on_nan (configs/base.yaml:59) raises FloatingPointError, and is honoured
  when switched off."""
  sco = _sco(n=330, with_labels=False)
  kw = dict(outputs=sco.get_rv("transcriptomic"), latents=api.RVmeta(6, "diag", True, "Latents"),
            encoder=api.NetConf([32], batchnorm=True, dropout=0.1), decoder=api.NetConf([32], batchnorm=True))
  m = api.VAE(**kw)
  m.fit(sco, epochs=2, batch_size=64)            # 330 = 5 * 64 + 10 -> 6 steps per epoch
  assert m.step == 12 and len(m.train_history["loss"]) == 2
  with pytest.raises(ValueError):                # fewer cells than one batch under drop_remainder
    api.VAE(**kw).fit(sco[np.arange(20)].create_dataset(batch_size=64, drop_remainder=True), metadata=sco, epochs=1)
  bad = api.VAE(**kw)
  bad._ensure_engine(64)
  p = bad._engine.get_params()
  p["enc0/W"][:] = np.nan
  bad._engine.set_params(p)
  with pytest.raises(FloatingPointError):
    bad.fit(sco, epochs=1, batch_size=64)
  with pytest.warns(UserWarning):
    bad.fit(sco, epochs=1, batch_size=64, terminate_on_nan=False)


def test_experiment_resumes_bit_identically(api, tmp_path):
  """SURVEY 8f-3 (train.py:49-59,107-108): every configuration owns `exp_<5-char hash>/`; an interrupted run of the
  same configuration restores weights / optimiser state / BN statistics / step from there and finishes ITS
  schedule -- ending bit-identical to the run that was never interrupted."""
  from sisua_amd.train import Experiment, config_hash
  cfg = dict(model=dict(name="vae", encoder=dict(units=[32]), decoder=dict(units=[32])),
             dataset=dict(name="cortex", batch_size=64), train=dict(epochs=4, valid_freq=20))
  ref = Experiment(cfg, save_path=str(tmp_path / "a")).run()
  assert ref.step == 4 * 33
  # the interrupted run: same configuration, killed after 50 iterations (mid-epoch; its last checkpoint is older)
  e1 = Experiment(dict(cfg, train=dict(epochs=4, valid_freq=20, max_iter=50)), save_path=str(tmp_path / "b"))
  assert e1.hash == config_hash(dict(e1.cfg)) and len(e1.hash) == 5 and e1.model_dir.endswith("exp_" + e1.hash)
  m1 = e1.run()
  assert m1.step == 50 and os.path.exists(os.path.join(e1.model_dir, "model.npz"))
  e2 = Experiment(cfg, save_path=str(tmp_path / "b"))
  assert e2.model_dir == e1.model_dir            # 'train' keys are not part of the identity
  m2 = e2.run()
  assert 0 < e2.resumed_from <= 50 and e2.resumed_from % 20 == 0   # restored from the checkpoint of a validation pass
  assert m2.step == ref.step
  a, b = ref._engine.get_params(), m2._engine.get_params()
  for k in a:
    assert np.array_equal(a[k], b[k]), k
  for i, st in ref._engine.get_bn().items():
    assert np.array_equal(st["moving_var"], m2._engine.get_bn()[i]["moving_var"])
  # a different model configuration gets a different directory
  assert Experiment(dict(cfg, model=dict(name="dca")), save_path=str(tmp_path / "b")).model_dir != e1.model_dir


def test_fit_data_parallel_two_replicas_on_one_gpu(api):
  """The data-parallel path BEHIND the operator surface (VERDICT r01 missing #4): two SingleCellModel.fit calls as
  threads of one process (LocalControlPlane -> loopback communicator).  Each rank keeps half of the training cells
  resident and draws batch_size / 2 of them per step (global batch preserved); with SyncBatchNorm the replicas
  stay bit-identical to each other, report the same history, and only rank 0 writes checkpoints."""
  import threading
  from sisua_amd.parallel import LocalControlPlane
  sco = _sco(n=520, with_labels=True)
  train, test = sco.split(0.8)
  omics = ["transcriptomic", "proteomic"]
  kw = dict(outputs=sco.get_rv("transcriptomic"), labels=[sco.get_rv("proteomic")], latents=api.RVmeta(6, "diag", True, "Latents"),
            encoder=api.NetConf([32], batchnorm=True, dropout=0.1), decoder=api.NetConf([32], batchnorm=True, dropout=0.1))
  planes = LocalControlPlane.group(2)
  models = [api.SISUA(**kw) for _ in range(2)]
  saved, errs = [[], []], [None, None]

  def run(r):
    try:
      ds = train.create_dataset(omics, labels_percent=0.5, batch_size=64, drop_remainder=True)
      vs = test.create_dataset(omics, labels_percent=1.0, batch_size=52, drop_remainder=True)
      models[r].fit(ds, valid=vs, metadata=sco, epochs=3, valid_freq=5, distributed=planes[r], sync_bn=True,
                    checkpoint=lambda: saved[r].append(models[r].step))
    except BaseException as e:  # noqa: BLE001
      errs[r] = e
      planes[r]._s.barrier.abort()

  ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(2)]
  for t in ts:
    t.start()
  for t in ts:
    t.join(300)
  assert not any(t.is_alive() for t in ts)
  for e in errs:
    if e is not None:
      raise e
  # 416 training cells -> 208 per rank, 32 per rank per step -> 6 steps per epoch
  assert [m.step for m in models] == [18, 18] and models[0]._engine.world == 2
  assert models[0].train_history == models[1].train_history and models[0].valid_history == models[1].valid_history
  assert len(models[0].train_history["loss"]) == 3 and np.isfinite(models[0].train_history["loss"]).all()
  a, b = models[0]._engine.get_params(), models[1]._engine.get_params()
  for k in a:
    assert np.array_equal(a[k], b[k]), k
  assert saved[0] and not saved[1]               # rank 0 is the only writer
  for m in models:
    m._engine.close()


def test_notebook_loss_band_plausibility(api):
  """The only numbers the reference holds for this path: the training log of its tutorial notebook
  (tutorials/notebook/SISUA_basic_tutorial.ipynb:238,364 -- VAE, zinb, hdim 64, zdim 16, batch 64, 64 epochs on
  pbmc8k_ly: loss 1129.1 (running mean of epoch 1) -> 522.8 in epoch 64, nllk_x 1172.8 -> 520.6, KLqp 0.70 -> 2.03).
  The data here is the shape- and sparsity-matched synthetic stand-in and the odin version behind that log is unknown,
  so this is a PLAUSIBILITY anchor, not parity: the same model and schedule start and end in the same band --
  untrained loss within a factor 1.6 of 1129 (here 1.1e3 at the first step), converged loss within a factor 1.6 of 523
  (here ~400), a KL of a few nats, the reconstruction term carrying the loss.  What does NOT match and is not
  asserted: the notebook's run needs ~60 epochs for what this one does in 2, and its KL starts at 0.7 and rises
  (here it starts at ~7 and falls to ~4) -- a warm-up or a different prior scale in that odin version; unpinnable."""
  from sisua_amd.train import Experiment
  cfg = dict(model=dict(name="vae", encoder=dict(units=[64]), decoder=dict(units=[64])),
             variables=dict(latents=dict(event_shape=16), transcriptomic=dict(posterior="zinb")),
             dataset=dict(name="8kly", batch_size=64))
  first = Experiment(dict(cfg, train=dict(epochs=1, max_iter=1))).run().train_history["loss"][0]   # the untrained model
  m = Experiment(dict(cfg, train=dict(epochs=64, valid_freq=500))).run()
  h = m.train_history
  assert len(h["loss"]) == 64 and m.step == 64 * (3381 // 64)
  last = h["loss"][-1]
  assert 1129.1 / 1.6 < first < 1129.1 * 1.6, first
  assert 522.8 / 1.6 < last < 522.8 * 1.6, last
  assert 0.1 < h["kl"][-1] < 20, h["kl"][-1]
  assert abs(h["nllk_x"][-1] + h["kl"][-1] - last) < 1e-2 * last      # beta = 1: loss = nllk_x + KL
  assert h["nllk_x"][-1] > 20 * h["kl"][-1]                            # as in the notebook (520.6 vs 2.03)
  assert _decreases(h["loss"][:12]) and m.valid_history["val_loss"][-1] < m.valid_history["val_loss"][0]


def test_misa_fit_predict(api):
  """MISA (sisua/models/vae.py:47-98): label posteriors become mixtures (with the reference's warning), the model
  trains, and predict returns the mixture distribution for the labels."""
  from sisua_amd import distributions as D
  from sisua_amd.data import SingleCellOMIC
  sco = SingleCellOMIC(synth_counts(600, 120, sparsity=0.8, seed=3), name="toy")
  sco.add_omic("proteomic", synth_labels(600, ((9, "mixnb2"),))[0])
  train, test = sco.split(0.8)
  assert api.get_model("misa") is api.MISA
  with pytest.warns(UserWarning, match="mixture distribution"):
    m = api.MISA(outputs=sco.get_rv("transcriptomic"), labels=[sco.get_rv("proteomic")], n_components=3,
                 latents=api.RVmeta(8, "diag", True, "Latents"),
                 encoder=api.NetConf([32], batchnorm=True, dropout=0.1), decoder=api.NetConf([32], batchnorm=True, dropout=0.1))
  assert m.labels[0].posterior == "mixnb" and m.labels[0].kwargs["n_components"] == 3 and m.is_semi_supervised
  omics = ["transcriptomic", "proteomic"]
  m.fit(train.create_dataset(omics, labels_percent=0.5, batch_size=64, drop_remainder=True),
        valid=test.create_dataset(omics, labels_percent=1.0, batch_size=60, drop_remainder=True), metadata=sco, epochs=15,
        valid_freq=20, learning_rate=2e-3)
  h = np.asarray(m.train_history["nllk_y"])
  assert len(h) == 15 and h[-3:].mean() < h[:3].mean()          # the label likelihood is being learnt
  X, Z = m.predict(test.create_dataset(omics, batch_size=40, shuffle=0), verbose=False)
  assert isinstance(X, tuple) and isinstance(X[1].distribution, D.MixtureNegativeBinomial)
  assert X[1].batch_shape == (test.n_obs,) and X[1].event_shape == (9,) and X[1].name == "proteomic"
  assert np.isfinite(X[1].mean()).all() and np.isfinite(X[1].log_prob(test.numpy("proteomic"))).all()
  with pytest.raises(ValueError):
    api.MISA(outputs=sco.get_rv("transcriptomic"), labels=[sco.get_rv("proteomic")], n_components=7)._make_config()


def test_misa_zero_inflated_fit_predict(api):
  """MISA(..., zero_inflated=True) (sisua/models/vae.py:76-84: the flag reaches every discrete label's kwargs): mixtures of
  ZERO-INFLATED negative binomials per label dimension; on labels with dropouts the trained head's density beats the plain mixture's."""
  from sisua_amd import distributions as D
  from sisua_amd.data import SingleCellOMIC
  sco = SingleCellOMIC(synth_counts(600, 100, sparsity=0.8, seed=4), name="toy")
  sco.add_omic("proteomic", synth_labels(600, ((8, "mixzinb2"),))[0])      # bimodal counts, a third of the entries zeroed
  train, test = sco.split(0.8)
  omics = ["transcriptomic", "proteomic"]
  lps = {}
  for zi in (True, False):
    m = api.MISA(outputs=sco.get_rv("transcriptomic"), labels=api.RVmeta(8, "mixnb", True, "proteomic"), n_components=2, zero_inflated=zi,
                 latents=api.RVmeta(8, "diag", True, "Latents"),
                 encoder=api.NetConf([32], batchnorm=True, dropout=0.1), decoder=api.NetConf([32], batchnorm=True, dropout=0.1))
    assert m.labels[0].kwargs["zero_inflated"] is zi
    assert m._make_config().labels == ((8, "mixzinb2" if zi else "mixnb2"),)
    m.fit(train.create_dataset(omics, labels_percent=0.8, batch_size=64, drop_remainder=True),
          valid=test.create_dataset(omics, labels_percent=1.0, batch_size=60, drop_remainder=True), metadata=sco, epochs=60,
          valid_freq=20, learning_rate=3e-3)
    h = np.asarray(m.train_history["nllk_y"])
    assert len(h) == 60 and np.isfinite(h).all() and h[-3:].mean() < h[:3].mean()
    X, Z = m.predict(test.create_dataset(omics, batch_size=40, shuffle=0), verbose=False)
    y = test.numpy("proteomic")
    assert isinstance(X[1].distribution, D.MixtureNegativeBinomial) and isinstance(X[1].distribution.components, D.ZeroInflated) == zi
    lps[zi] = X[1].log_prob(y)
    assert np.isfinite(lps[zi]).all() and np.isfinite(X[1].mean()).all() and X[1].event_shape == (8,)
    assert X[1].sample(2, seed=0).shape == (2, test.n_obs, 8)
  assert lps[True].mean() > lps[False].mean()


def test_misa_continuous_labels_fit_predict(api):
  """MISA with a CONTINUOUS label variable (vae.py:86-92): a non-mixture continuous posterior becomes 'mixgaussian' (with the
  reference's warning) -- a mixture of normals per label dimension --, the model trains on it, and predict returns that mixture."""
  from sisua_amd import distributions as D
  from sisua_amd.data import SingleCellOMIC
  sco = SingleCellOMIC(synth_counts(600, 100, sparsity=0.8, seed=4), name="toy")
  sco.add_omic("proteomic", synth_labels(600, ((7, "mixgauss2"),))[0])      # log-normalised levels: real-valued, bimodal
  train, test = sco.split(0.8)
  with pytest.warns(UserWarning, match="mixture distribution"):
    m = api.MISA(outputs=sco.get_rv("transcriptomic"), labels=[sco.get_rv("proteomic", "gaussian")], n_components=2,
                 latents=api.RVmeta(8, "diag", True, "Latents"),
                 encoder=api.NetConf([32], batchnorm=True, dropout=0.1), decoder=api.NetConf([32], batchnorm=True, dropout=0.1))
  assert m.labels[0].posterior == "mixgaussian" and m.labels[0].kwargs["n_components"] == 2
  assert m._make_config().labels == ((7, "mixgauss2"),)
  omics = ["transcriptomic", "proteomic"]
  m.fit(train.create_dataset(omics, labels_percent=0.5, batch_size=64, drop_remainder=True),
        valid=test.create_dataset(omics, labels_percent=1.0, batch_size=60, drop_remainder=True), metadata=sco, epochs=15,
        valid_freq=20, learning_rate=2e-3)
  h = np.asarray(m.train_history["nllk_y"])
  assert len(h) == 15 and np.isfinite(h).all() and h[-3:].mean() < h[:3].mean()
  X, Z = m.predict(test.create_dataset(omics, batch_size=40, shuffle=0), verbose=False)
  assert isinstance(X, tuple) and isinstance(X[1].distribution, D.MixtureNormal)
  assert X[1].batch_shape == (test.n_obs,) and X[1].event_shape == (7,)
  y = test.numpy("proteomic")
  lp = X[1].log_prob(y)
  assert np.isfinite(X[1].mean()).all() and np.isfinite(lp).all()
  # the head's own density integrates to one where it matters: its log-probability of the labels beats a far-off constant guess
  assert lp.mean() > D.Independent(D.Normal(np.zeros_like(y) + 10.0, np.ones_like(y)), 1).log_prob(y).mean()
  assert np.allclose(X[1].distribution.mean(), (np.exp(X[1].distribution._log_pi()) * X[1].distribution.components.mean()).sum(-2))


def test_misa_mixtril_labels_fit_predict(api):
  """MISA as the reference's own docstring builds it (sisua/models/vae.py:54-60): `adt = RVmeta(dim, 'mixtril', True, 'proteomic')`,
  `MISA(rna, adt, n_components=2)` -- ONE mixture of full-covariance Gaussians over the label vector.  Trains, and predict returns
  that mixture; on correlated labels its density beats the independent-dimensions head's trained the same way."""
  from sisua_amd import distributions as D
  from sisua_amd.data import SingleCellOMIC
  sco = SingleCellOMIC(synth_counts(600, 100, sparsity=0.8, seed=4), name="toy")
  sco.add_omic("proteomic", synth_labels(600, ((6, "mixtril2"),))[0])      # real-valued, two populations, correlated dimensions
  train, test = sco.split(0.8)
  omics = ["transcriptomic", "proteomic"]
  lps = {}
  for post in ("mixtril", "mixgaussian"):
    adt = api.RVmeta(6, post, True, "proteomic")
    m = api.MISA(outputs=sco.get_rv("transcriptomic"), labels=adt, n_components=2, latents=api.RVmeta(8, "diag", True, "Latents"),
                 encoder=api.NetConf([32], batchnorm=True, dropout=0.1), decoder=api.NetConf([32], batchnorm=True, dropout=0.1))
    assert m._make_config().labels == ((6, "mixtril2" if post == "mixtril" else "mixgauss2"),)
    m.fit(train.create_dataset(omics, labels_percent=0.8, batch_size=64, drop_remainder=True),
          valid=test.create_dataset(omics, labels_percent=1.0, batch_size=60, drop_remainder=True), metadata=sco, epochs=40,
          valid_freq=20, learning_rate=3e-3)
    h = np.asarray(m.train_history["nllk_y"])
    assert len(h) == 40 and np.isfinite(h).all() and h[-3:].mean() < h[:3].mean()
    X, Z = m.predict(test.create_dataset(omics, batch_size=40, shuffle=0), verbose=False)
    y = test.numpy("proteomic")
    lps[post] = X[1].log_prob(y)
    assert np.isfinite(lps[post]).all() and X[1].batch_shape == (test.n_obs,) and X[1].event_shape == (6,)
    if post == "mixtril":
      assert isinstance(X[1], D.MixtureMultivariateNormalTriL) and X[1].name == "proteomic"
      assert np.isfinite(X[1].mean()).all() and (X[1].variance() > 0).all()
      smp = X[1].sample(3, seed=0)
      assert smp.shape == (3, test.n_obs, 6) and np.isfinite(smp).all()
  assert lps["mixtril"].mean() > lps["mixgaussian"].mean()
  with pytest.raises(ValueError):
    api.MISA(outputs=sco.get_rv("transcriptomic"), labels=api.RVmeta(70, "mixtril", True, "proteomic"))._make_config()


def test_scale_fit_predict(api, tmp_path):
  """SCALE (sisua/models/scale.py:13-49): mixture prior over the latents, Monte-Carlo KL; trains, predicts, and its
  prior parameters move and survive a checkpoint."""
  sco = _sco(with_labels=False)
  train, test = sco.split(0.8)
  assert api.get_model("scale") is api.SCALE
  m = api.SCALE(outputs=sco.get_rv("transcriptomic"), latents=api.RVmeta(8, "mixgaus", True, "Latents"), n_components=5,
                encoder=api.NetConf([32], batchnorm=True, dropout=0.1), decoder=api.NetConf([32], batchnorm=True, dropout=0.1))
  m.fit(train, epochs=12, batch_size=64, learning_rate=2e-3)
  h = np.asarray(m.train_history["loss"])   # (the one-sample Monte-Carlo KL makes late epochs noisy: judge the trend)
  assert _decreases(h[:7]) and h[-3:].mean() < h[3:6].mean() and np.isfinite(m.train_history["kl"]).all(), h
  p = m._engine.get_params()
  assert p["prior/loc"].shape == (5, 8) and p["prior/logits"].shape == (5,) and np.abs(p["prior/scale"]).max() > 1e-3
  X, Z = m.predict(test.numpy(), batch_size=64, verbose=False)
  assert Z.mean().shape == (test.n_obs, 8) and X.mean().shape == (test.n_obs, 120)
  mllk, _ = m.marginal_log_prob(inputs=test.numpy()[:16], sample_shape=8)
  assert np.isfinite(mllk).all()
  path = os.path.join(tmp_path, "scale")
  m.save_weights(path)
  m2 = api.load_model(path)
  assert type(m2) is api.SCALE and np.array_equal(m2._engine.get_params()["prior/loc"], p["prior/loc"])
  with pytest.raises(ValueError):
    api.SCALE(outputs=sco.get_rv("transcriptomic"), covariance="spherical")
  with pytest.raises(ValueError):
    api.SCALE(outputs=sco.get_rv("transcriptomic"), covariance="tril", tie_loc=True)


def test_scale_mixture_posterior_fit_predict(api, tmp_path):
  """SCALE(mixture='posterior'): sisua/models/scale.py:26,38-47 read literally -- the latent POSTERIOR is the mixture-density layer
  ((1 + 2 C) D-wide latent head, no prior tensors), standard-normal prior, Monte-Carlo KL.  Trains; predict / encode report the
  mixture's moments; scoring and a checkpoint round trip work."""
  sco = _sco(with_labels=False)
  train, test = sco.split(0.8)
  m = api.SCALE(outputs=sco.get_rv("transcriptomic"), latents=api.RVmeta(6, "mixgaus", True, "Latents"), n_components=3, mixture="posterior",
                encoder=api.NetConf([32], batchnorm=True, dropout=0.1), decoder=api.NetConf([32], batchnorm=True, dropout=0.1))
  cfg = m._make_config()
  assert cfg.latent_mixture and not cfg.scale_tril
  m.fit(train, epochs=12, batch_size=64, learning_rate=2e-3)
  h = np.asarray(m.train_history["loss"])
  assert _decreases(h[:7]) and h[-3:].mean() < h[3:6].mean() and np.isfinite(m.train_history["kl"]).all(), h
  p = m._engine.get_params()
  assert p["lat/W"].shape == (32, (1 + 2 * 3) * 6) and not any(k.startswith("prior/") for k in p)
  X, Z = m.predict(test.numpy(), batch_size=64, verbose=False)
  assert Z.mean().shape == (test.n_obs, 6) and np.isfinite(Z.mean()).all() and (Z.stddev() > 0).all() and np.isfinite(X.mean()).all()
  mllk, _ = m.marginal_log_prob(inputs=test.numpy()[:16], sample_shape=8)
  assert np.isfinite(mllk).all()
  path = os.path.join(tmp_path, "scale_post")
  m.save_weights(path)
  m2 = api.load_model(path)
  assert type(m2) is api.SCALE and m2._make_config().latent_mixture
  assert np.array_equal(m2._engine.get_params()["lat/W"], p["lat/W"])
  for bad in (dict(n_components=9), dict(covariance="tril"), dict(tie_loc=True), dict(mixture="both")):
    with pytest.raises(ValueError):
      api.SCALE(outputs=sco.get_rv("transcriptomic"), latents=api.RVmeta(6, "mixgaus", True, "Latents"), **dict(dict(n_components=3, mixture="posterior"), **bad))


def test_scale_full_covariance_fit_predict(api, tmp_path):
  """SCALE(covariance='tril') (sisua/models/scale.py:28,35): every mixture component carries a lower-triangular scale factor; trains,
  the factors leave the identity (below the diagonal too, never above it), scoring and a checkpoint round trip work."""
  sco = _sco(with_labels=False)
  train, test = sco.split(0.8)
  m = api.SCALE(outputs=sco.get_rv("transcriptomic"), latents=api.RVmeta(6, "mixgaus", True, "Latents"), n_components=4, covariance="tril",
                encoder=api.NetConf([32], batchnorm=True, dropout=0.1), decoder=api.NetConf([32], batchnorm=True, dropout=0.1))
  assert m._make_config().covariance == "tril" and m._make_config().scale_tril
  m.fit(train, epochs=12, batch_size=64, learning_rate=2e-3)
  h = np.asarray(m.train_history["loss"])
  assert _decreases(h[:7]) and h[-3:].mean() < h[3:6].mean() and np.isfinite(m.train_history["kl"]).all(), h
  p = m._engine.get_params()
  L = p["prior/scale"].reshape(4, 6, 6)
  init = np.zeros((6, 6), np.float32); init[np.arange(6), np.arange(6)] = np.log(np.expm1(1.0))
  assert p["prior/scale"].shape == (24, 6) and np.all(np.triu(L, 1) == 0.0)
  assert np.abs(np.tril(L, -1)).max() > 1e-3 and np.abs(np.einsum("cpp->cp", L) - np.log(np.expm1(1.0))).max() > 1e-3
  X, Z = m.predict(test.numpy(), batch_size=64, verbose=False)
  assert Z.mean().shape == (test.n_obs, 6) and np.isfinite(X.mean()).all()
  mllk, _ = m.marginal_log_prob(inputs=test.numpy()[:16], sample_shape=8)
  assert np.isfinite(mllk).all()
  path = os.path.join(tmp_path, "scale_tril")
  m.save_weights(path)
  m2 = api.load_model(path)
  assert type(m2) is api.SCALE and m2._make_config().covariance == "tril"
  assert np.array_equal(m2._engine.get_params()["prior/scale"], p["prior/scale"])


def test_scale_tied_mixture_parameters(api, tmp_path):
  """SCALE's tie options (scale.py:29-33): one scale vector shared by every component and fixed uniform weights stay tied through
  training (identical rows, zero logits) while the untied locations move apart; the options survive save / load."""
  sco = _sco(with_labels=False)
  train, _ = sco.split(0.8)
  m = api.SCALE(outputs=sco.get_rv("transcriptomic"), latents=api.RVmeta(6, "mixgaus", True, "Latents"), n_components=4,
                tie_scale=True, tie_mixtures=True,
                encoder=api.NetConf([32], batchnorm=True, dropout=0.1), decoder=api.NetConf([32], batchnorm=True, dropout=0.1))
  cfg = m._make_config()
  assert cfg.tie_scale and cfg.tie_mixtures and not cfg.tie_loc
  m.fit(train, epochs=6, batch_size=64, learning_rate=2e-3)
  p = m._engine.get_params()
  assert np.isfinite(m.train_history["loss"]).all()
  assert np.abs(p["prior/scale"]).max() > 1e-4 and np.array_equal(p["prior/scale"], np.broadcast_to(p["prior/scale"][:1], p["prior/scale"].shape))
  assert not p["prior/logits"].any()
  assert not np.array_equal(p["prior/loc"][0], p["prior/loc"][1])
  path = os.path.join(tmp_path, "scale_tied")
  m.save_weights(path)
  m2 = api.load_model(path)
  c2 = m2._make_config()
  assert c2.tie_scale and c2.tie_mixtures and not c2.tie_loc and np.array_equal(m2._engine.get_params()["prior/scale"], p["prior/scale"])


def test_unsupervised_mse_fit_predict(api):
  """The reference's test_unsupervised_fit_predict (tests/test_singlecell_models.py:93-114): a DeepCountAutoencoder with
  outputs=RVmeta(dim, posterior='mse'), latent_dim 10: the loss falls, predict(sample_shape=2) returns VectorDeterministic
  outputs AND latents with the draw axis in front, from a dataset and from a raw matrix; and -log_prob is the mean squared
  error of the prediction exactly."""
  from sisua_amd import distributions as D
  sco = _sco(with_labels=False)
  train, test = sco.split(0.8)
  dca = api.DeepCountAutoencoder(outputs=api.RVmeta(sco.n_vars if hasattr(sco, "n_vars") else sco.numpy().shape[1], posterior="mse"),
                                 latents=api.RVmeta(10, "relu", True, name="Latents"))
  assert not dca.is_zero_inflated
  dca.fit(train.create_dataset(batch_size=64, drop_remainder=True), valid=test.create_dataset(batch_size=60, drop_remainder=True),
          metadata=sco, epochs=10, valid_freq=9, learning_rate=2e-3)
  # (the squared error of a few large counts dominates a minibatch's loss: the epoch means fall overall, not in every move)
  h = np.asarray(dca.train_history["loss"])
  assert h[-3:].mean() < 0.8 * h[:3].mean() and np.mean(np.diff(h) < 0) >= 0.6, h
  assert len(dca.valid_history["val_loss"]) >= 2 and dca.valid_history["val_loss"][-1] < dca.valid_history["val_loss"][0]
  pX, qZ = dca.predict(test.create_dataset(batch_size=50, shuffle=0), sample_shape=2, verbose=False)
  assert isinstance(pX, D.VectorDeterministic) and pX.batch_shape[0] == 2 and pX.batch_shape[1] == test.n_obs
  assert isinstance(qZ, D.Deterministic)
  X = sco.numpy()[:128]
  pX, qZ = dca.predict(X, sample_shape=2, verbose=False)
  assert isinstance(pX, D.VectorDeterministic) and pX.batch_shape[0] == 2 and pX.batch_shape[1] == X.shape[0]
  d1 = -pX.log_prob(X)[0]
  assert np.all(d1 == np.mean(np.square(X - pX.mean()[0]), axis=-1))
  with pytest.raises(Exception):
    dca.marginal_log_prob(X[:8], sample_shape=3)          # not a normalised density


def test_scalar_fit_predict(api):
  """SCALAR (sisua/models/scale.py:52-59: `class SCALAR(SCALE, SISUA)`): SCALE's mixture prior with SISUA's semi-supervised
  label heads -- registry, the reference's warning for a non-mixture latent, fit (both terms are learnt), predict."""
  from sisua_amd import distributions as D
  sco = _sco()
  train, test = sco.split(0.8)
  assert api.get_model("scalar") is api.SCALAR and issubclass(api.SCALAR, api.SCALE)
  with pytest.warns(UserWarning, match="mixture distribution"):
    m = api.SCALAR(outputs=sco.get_rv("transcriptomic"), labels=[sco.get_rv("proteomic")], latents=api.RVmeta(8, "diag", True, "Latents"),
                   n_components=4, encoder=api.NetConf([32], batchnorm=True, dropout=0.1), decoder=api.NetConf([32], batchnorm=True, dropout=0.1))
  assert m.is_semi_supervised and m._make_config().model == "scale" and m._make_config().labels == ((9, "nb"),)
  omics = ["transcriptomic", "proteomic"]
  m.fit(train.create_dataset(omics, labels_percent=0.5, batch_size=64, drop_remainder=True),
        valid=test.create_dataset(omics, labels_percent=1.0, batch_size=60, drop_remainder=True), metadata=sco, epochs=12,
        valid_freq=20, learning_rate=2e-3)
  hx, hy = np.asarray(m.train_history["nllk_x"]), np.asarray(m.train_history["nllk_y"])
  assert len(hx) == 12 and hx[-3:].mean() < hx[:3].mean() and hy[-3:].mean() < hy[:3].mean()
  X, Z = m.predict(test.create_dataset(omics, batch_size=40, shuffle=0), verbose=False)
  assert isinstance(X, tuple) and X[1].batch_shape == (test.n_obs,) and X[1].event_shape == (9,)
  assert np.isfinite(X[1].mean()).all() and Z.event_shape == (8,)


def test_fvae_fit_predict(api, tmp_path):
  """FVAE / SemiFVAE (sisua/models/fvae.py:9-18): the VAE trains against -ELBO + gamma TC while the discriminator learns
  to tell z from permute_dims(z) in the same step; both objectives are logged; predict / checkpoint as any model."""
  sco = _sco(with_labels=False)
  train, test = sco.split(0.8)
  assert api.get_model("fvae") is api.FVAE and api.get_model("semifvae") is api.SemiFVAE
  kw = dict(latents=api.RVmeta(8, "diag", True, "Latents"), encoder=api.NetConf([32], batchnorm=True, dropout=0.1),
            decoder=api.NetConf([32], batchnorm=True, dropout=0.1), discriminator=dict(units=64, n_hidden_layers=3))
  m = api.FVAE(outputs=sco.get_rv("transcriptomic"), gamma=4.0, **kw)
  assert m._make_config().gamma == 4.0 and m._make_config().disc_units == 64
  m.fit(train, valid=test, epochs=10, batch_size=64, learning_rate=2e-3, valid_freq=15)
  h = m.train_history
  assert len(m.valid_history["val_loss"]) >= 1 and np.isfinite(m.valid_history["val_loss"]).all()
  assert _decreases(h["nllk_x"]) and np.isfinite(h["tc"]).all() and np.isfinite(h["dtc_loss"]).all()
  # an untrained discriminator scores chance: 1/2 [softplus(0) + softplus(0)] = log 2; training it must not end above that
  assert h["dtc_loss"][-1] <= np.log(2.0) + 1e-3, h["dtc_loss"]
  p = m._engine.get_params()
  assert p["disc0/W"].shape == (8, 64) and p["discout/W"].shape == (64, 1) and np.abs(p["disc2/b"]).max() > 0
  X, Z = m.predict(test.numpy(), batch_size=64, verbose=False)
  assert Z.mean().shape == (test.n_obs, 8) and X.mean().shape == (test.n_obs, 120)
  path = os.path.join(tmp_path, "fvae")
  m.save_weights(path)
  m2 = api.load_model(path)
  assert type(m2) is api.FVAE and np.array_equal(m2._engine.get_params()["disc1/W"], p["disc1/W"])
  with pytest.raises(ValueError):
    api.FVAE(outputs=sco.get_rv("transcriptomic"), discriminator=dict(batchnorm=True))
  # semi-supervised form: one one-hot label variable, classified by the discriminator
  from sisua_amd.data import SingleCellOMIC
  x = synth_counts(600, 120, sparsity=0.8, seed=3)
  cls = (np.log1p(x[:, :40]).sum(1) > np.median(np.log1p(x[:, :40]).sum(1))).astype(int) + 2 * (x[:, 40:80].sum(1) > np.median(x[:, 40:80].sum(1)))
  sco2 = SingleCellOMIC(x, name="toy")
  sco2.add_omic("celltype", np.eye(4, dtype=np.float32)[cls])
  tr2, _ = sco2.split(0.8)
  s = api.SemiFVAE(outputs=sco2.get_rv("transcriptomic"), labels=[sco2.get_rv("celltype", "onehot")], alpha=5.0, **kw)
  assert s.is_semi_supervised and s._make_config().disc_outputs == 4
  s.fit(tr2.create_dataset(["transcriptomic", "celltype"], labels_percent=0.5, batch_size=64, drop_remainder=True), metadata=sco2,
        epochs=10, learning_rate=2e-3)
  hs = s.train_history
  assert hs["nllk_y"][-1] < hs["nllk_y"][0] and np.isfinite(hs["loss"]).all(), hs["nllk_y"]   # the classifier learns the labelled cells
  with pytest.raises(ValueError):
    api.SemiFVAE(outputs=sco2.get_rv("transcriptomic"), labels=[sco.get_rv("transcriptomic")])
  # several label variables (round 6): `labels` is a list in the reference's constructor (fvae.py:15-18) -- one logit per class of every
  # variable; both classifiers learn their labelled cells
  cls2 = (x[:, 80:].sum(1) > np.median(x[:, 80:].sum(1))).astype(int)
  sco2.add_omic("condition", np.eye(2, dtype=np.float32)[cls2])
  tr3, te3 = sco2.split(0.8)
  s2 = api.SemiFVAE(outputs=sco2.get_rv("transcriptomic"), labels=[sco2.get_rv("celltype", "onehot"), sco2.get_rv("condition", "onehot")], alpha=5.0, **kw)
  cfg2 = s2._make_config()
  assert cfg2.disc_outputs == 6 and cfg2.labels == ((4, "onehot"), (2, "onehot")) and cfg2.head_labels == ()
  s2.fit(tr3.create_dataset(["transcriptomic", "celltype", "condition"], labels_percent=0.5, batch_size=64, drop_remainder=True), metadata=sco2,
         epochs=10, learning_rate=2e-3)
  h2 = s2.train_history
  assert h2["nllk_y"][-1] < h2["nllk_y"][0] and np.isfinite(h2["loss"]).all() and h2["nllk_y"][0] > hs["nllk_y"][0], (h2["nllk_y"], hs["nllk_y"])   # (two cross-entropies)
  assert s2._engine.get_params()["discout/W"].shape == (64, 6)
  X2, Z2 = s2.predict(te3.numpy(), batch_size=64, verbose=False)
  assert Z2.mean().shape == (te3.n_obs, 8) and np.isfinite(X2.mean()).all()
  p2 = os.path.join(tmp_path, "semifvae2")
  s2.save_weights(p2)
  s3 = api.load_model(p2)
  assert type(s3) is api.SemiFVAE and np.array_equal(s3._engine.get_params()["discout/W"], s2._engine.get_params()["discout/W"])
  with pytest.raises(ValueError):   # 33 classes in all
    api.SemiFVAE(outputs=sco2.get_rv("transcriptomic"), labels=[api.RVmeta(20, "onehot", name="a"), api.RVmeta(13, "onehot", name="b")])


def test_variational_model_with_two_outputs(api):
  """The reference's `test_variational_model` (tests/test_singlecell_models.py:129-141): `VAE(outputs=[RVmeta(G, 'zinb'),
  RVmeta(P, 'nbd')])`, `fit(sco)`, `(pX, pY), qZ = vae.predict(X, sample_shape=2)` with its structure asserts -- and the second
  output is TRAINED (a fully observed head with weight 1: its likelihood falls, its tensors move, the metric nllk_o is logged)."""
  from sisua_amd import distributions as D
  sco = _sco()
  n_genes, n_prots = sco.n_vars, sco.numpy("proteomic").shape[1]
  kw = dict(latents=api.RVmeta(8, "diag", True, "Latents"), encoder=api.NetConf([32], batchnorm=True, dropout=0.1),
            decoder=api.NetConf([32], batchnorm=True, dropout=0.1))
  vae = api.VAE(outputs=[api.RVmeta(n_genes, "zinb", name="transcriptomic"), api.RVmeta(n_prots, "nbd", name="proteomic")], **kw)
  cfg = vae._make_config()
  assert cfg.extra_outputs == ((n_prots, "nbd"),) and cfg.labels == () and not vae.is_semi_supervised
  assert [l.name for l in vae.posteriors] == ["transcriptomic", "proteomic"] and len(vae.output_layers) == 2
  vae.fit(sco, epochs=12, batch_size=64, learning_rate=2e-3, verbose=False)   # (a SingleCellOMIC: both OMICs are taken)
  assert _decreases(vae.train_history["loss"]), vae.train_history["loss"]
  h = vae.train_history["nllk_o"]
  assert len(h) == 12 and h[-1] < h[0] and all(v > 0 for v in h)
  assert any(k.startswith("lab0/") for k in vae._engine.names)
  W0 = api.VAE(outputs=[api.RVmeta(n_genes, "zinb"), api.RVmeta(n_prots, "nbd")], **kw)._ensure_engine(64).get_params()["lab0/W"]
  assert np.abs(vae._engine.get_params()["lab0/W"] - W0).max() > 1e-3
  X = sco.numpy()[:128]
  (pX, pY), qZ = vae.predict(X, sample_shape=2, verbose=False)
  assert isinstance(pX.distribution, D.ZeroInflated) and isinstance(pX.distribution.count_distribution, D.NegativeBinomial)
  assert isinstance(pY.distribution, D.NegativeBinomialDisp) and pY.name == "proteomic"
  assert pX.batch_shape[0] == 2 and pX.batch_shape[1] == X.shape[0]
  assert pY.batch_shape[0] == 2 and pY.batch_shape[1] == X.shape[0] and pY.event_shape == (n_prots,)
  assert isinstance(qZ, D.MultivariateNormalDiag)
  assert qZ.sample().shape == (X.shape[0], vae.latents[0].event_shape[0])
  # the proteins' predicted mean tracks their scale after training (real-valued levels around 2: mean within a factor 2)
  ratio = pY.mean().mean() / sco.numpy("proteomic")[:128].mean()
  assert 0.5 < ratio < 2.0, ratio
  # a dataset that lacks the second output's targets is refused, and so is the (unbuilt) joint marginal likelihood
  with pytest.raises(ValueError):
    api.VAE(outputs=[api.RVmeta(n_genes, "zinb"), api.RVmeta(n_prots, "nbd")], **kw).fit(sco.create_dataset(batch_size=64), metadata=sco, epochs=1)
  # the JOINT marginal likelihood log p(x, y) against the oracle's (same Philox draws: cell ids = index within the minibatch)
  from oracle import sisua_oracle as so
  Y = sco.numpy("proteomic")[:128]
  mllk, llk = vae.marginal_log_prob(inputs=[X[:40], Y[:40]], sample_shape=6, batch_size=64)
  assert set(llk) == {"transcriptomic", "proteomic"} and mllk.shape == (40,)
  spec = so.Spec(**vae._make_config().to_dict())
  e = vae._engine
  params = {k: v.astype(np.float64) for k, v in e.get_params().items()}
  names = [p for p, _ in so.bn_manifest(spec)]
  bn = {}
  for i, st in e.get_bn().items():
    bn[f"{names[i]}/moving_mean"], bn[f"{names[i]}/moving_var"] = st["moving_mean"].astype(np.float64), st["moving_var"].astype(np.float64)
  ref_m, ref_l = so.marginal_log_prob(spec, params, bn, X[:40], np.arange(40), 6, y=[Y[:40]])
  assert np.allclose(mllk, ref_m, rtol=1e-4, atol=1e-2), np.abs(mllk - ref_m).max()
  assert np.allclose(llk["transcriptomic"], ref_l, rtol=1e-4, atol=1e-2)
  with pytest.raises(ValueError):   # the second output's targets are part of the joint
    vae.marginal_log_prob(inputs=X[:8], sample_shape=4)
  with pytest.raises(ValueError):   # a posterior that has no head form is refused at construction, never ignored
    api.VAE(outputs=[api.RVmeta(n_genes, "zinb"), api.RVmeta(n_prots, "poisson")], **kw)
  # round 5: outputs[1:] on FactorVAE (fvae.py:9-18 passes `outputs` through unchanged) -- trained and predicted like the VAE's
  fv = api.FVAE(outputs=[api.RVmeta(n_genes, "zinb", name="transcriptomic"), api.RVmeta(n_prots, "nbd", name="proteomic")], **kw)
  fv.fit(sco, epochs=4, batch_size=64, learning_rate=2e-3, verbose=False)
  assert np.isfinite(fv.train_history["loss"]).all() and fv.train_history["nllk_o"][-1] < fv.train_history["nllk_o"][0]
  (fX, fY), fZ = fv.predict(X, sample_shape=2, verbose=False)
  assert isinstance(fY.distribution, D.NegativeBinomialDisp) and fY.batch_shape[1] == X.shape[0] and fY.event_shape == (n_prots,)


@pytest.mark.parametrize("covariance", ["none", "tril"])
def test_scale_joint_marginal_of_two_outputs(api, covariance):
  """VERDICT r04 Missing 5: the JOINT marginal likelihood log p(x, y) of a model with several outputs under SCALE's trainable mixture
  prior (scale.py:13-49; posterior.py:964-967) -- every draw's weight is log p(x|z) + log p(y|z) + log p_mix(z) - log q(z|x), with the
  mixture density from the model's current parameters (diagonal and full-covariance components) -- against the oracle's, same Philox
  draws.  It used to refuse."""
  from oracle import sisua_oracle as so
  sco = _sco()
  n_genes, n_prots = sco.n_vars, sco.numpy("proteomic").shape[1]
  m = api.SCALE(outputs=[api.RVmeta(n_genes, "zinb", name="transcriptomic"), api.RVmeta(n_prots, "nbd", name="proteomic")],
                latents=api.RVmeta(6, "mixgaus", True, "Latents"), n_components=4, covariance=covariance,
                encoder=api.NetConf([32], batchnorm=True, dropout=0.1), decoder=api.NetConf([32], batchnorm=True, dropout=0.1))
  m.fit(sco, epochs=3, batch_size=64, learning_rate=2e-3, verbose=False)   # (parameters off their initial values: the components differ)
  X, Y = sco.numpy()[:40], sco.numpy("proteomic")[:40]
  mllk, llk = m.marginal_log_prob(inputs=[X, Y], sample_shape=6, batch_size=64)
  assert set(llk) == {"transcriptomic", "proteomic"} and mllk.shape == (40,) and np.isfinite(mllk).all()
  spec = so.Spec(**m._make_config().to_dict())
  e = m._engine
  params = {k: v.astype(np.float64) for k, v in e.get_params().items()}
  names = [p for p, _ in so.bn_manifest(spec)]
  bn = {}
  for i, st in e.get_bn().items():
    bn[f"{names[i]}/moving_mean"], bn[f"{names[i]}/moving_var"] = st["moving_mean"].astype(np.float64), st["moving_var"].astype(np.float64)
  ref_m, ref_l = so.marginal_log_prob(spec, params, bn, X, np.arange(40), 6, y=[Y])
  assert np.allclose(mllk, ref_m, rtol=1e-4, atol=1e-2), np.abs(mllk - ref_m).max()
  assert np.allclose(llk["transcriptomic"], ref_l, rtol=1e-4, atol=1e-2)
  # ... and the mixture term matters: with a standard-normal prior in its place the estimate moves
  z = np.random.default_rng(0).normal(size=(3, 5, 6))
  assert np.abs(m._mixture_prior_log_prob(z) - (-0.5 * z ** 2 - 0.5 * np.log(2 * np.pi)).sum(-1)).max() > 1e-3


def test_joint_marginal_under_the_mixture_density_posterior_refuses(api):
  """SCALE(mixture='posterior') with several outputs has no per-draw log q_mix(z | x) on the host side of marginal_log_prob; the
  single-Gaussian density of the mixture's mean / scale is not it.  The call refuses instead of returning wrongly weighted estimates; the
  one-output call (the device's per-draw term) stays available."""
  sco = _sco()
  n_genes, n_prots = sco.n_vars, sco.numpy("proteomic").shape[1]
  kw = dict(latents=api.RVmeta(6, "mixgaus", True, "Latents"), n_components=3, mixture="posterior",
            encoder=api.NetConf([32], batchnorm=True, dropout=0.1), decoder=api.NetConf([32], batchnorm=True, dropout=0.1))
  m = api.SCALE(outputs=[api.RVmeta(n_genes, "zinb", name="transcriptomic"), api.RVmeta(n_prots, "nbd", name="proteomic")], **kw)
  m.fit(sco, epochs=1, batch_size=64, verbose=False)
  X, Y = sco.numpy()[:16], sco.numpy("proteomic")[:16]
  with pytest.raises(NotImplementedError, match="mixture-density posterior"):
    m.marginal_log_prob(inputs=[X, Y], sample_shape=4, batch_size=64)
  one = api.SCALE(outputs=api.RVmeta(n_genes, "zinb", name="transcriptomic"), **kw)
  one.fit(sco, epochs=1, batch_size=64, verbose=False)
  mllk, _ = one.marginal_log_prob(inputs=X, sample_shape=4, batch_size=64)
  assert mllk.shape == (16,) and np.isfinite(mllk).all()


def test_scvi_extra_outputs_and_gene_dispersion(api, tmp_path):
  """scvi.py:168-169 (`pY = [p(d) for p in self.posteriors[1:]]`) and scvi.py:55-56,66-86 (`dispersion` / `inflation` kept by the
  distribution layer instead of a Dense head): trained, predicted, saved and restored."""
  from sisua_amd import distributions as D
  sco = _sco()
  n_genes, n_prots = sco.n_vars, sco.numpy("proteomic").shape[1]
  kw = dict(latents=api.RVmeta(8, "diag", True, "Latents"), encoder=api.NetConf([32], batchnorm=True, dropout=0.1),
            encoder_l=api.NetConf([16], batchnorm=True, dropout=0.1), decoder=api.NetConf([32], batchnorm=True, dropout=0.1))
  m = api.SCVI([api.RVmeta(n_genes, "zinbd", name="rna", kwargs=dict(dispersion="gene", inflation="share")),
                api.RVmeta(n_prots, "nbd", name="adt")], **kw)
  assert (m.dispersion, m.inflation) == ("share", "share")
  names = [nm for nm, _ in __import__("sisua_amd.config", fromlist=["manifest"]).manifest(m._make_config())]
  assert "out0/W" in names and "out1/W" not in names and "out2/W" not in names and {"out1/b", "out2/b", "lab0/W"} <= set(names)
  m.fit(sco, epochs=12, batch_size=64, learning_rate=2e-3, verbose=False)
  assert _decreases(m.train_history["loss"]), m.train_history["loss"]
  p = m._engine.get_params()
  assert np.abs(p["out1/b"]).max() > 1e-3 and np.abs(p["out2/b"]).max() > 1e-3 and p["out1/b"].shape == (n_genes,)   # the shared vectors moved
  X = sco.numpy()[:100]
  (pX, pY), (qZ, qL) = m.predict(X, verbose=False)
  assert isinstance(pX.distribution, D.ZeroInflated) and isinstance(pX.distribution.count_distribution, D.NegativeBinomialDisp)
  assert isinstance(pY.distribution, D.NegativeBinomialDisp) and pY.batch_shape == (100,) and pY.event_shape == (n_prots,)
  # one dispersion / gate vector for every cell: theta = exp(out1/b), gate logits = out2/b
  disp = pX.distribution.count_distribution.disp
  assert np.allclose(disp, np.exp(p["out1/b"])[None, :], rtol=1e-5) and np.allclose(pX.distribution.logits, p["out2/b"][None, :], rtol=1e-5, atol=1e-7)
  assert qL.sample(1).shape == (1, 100, 1)
  path = str(tmp_path / "scvi_opts")
  m.save_weights(path)
  m2 = api.load_model(path)
  import dataclasses
  assert (m2.dispersion, m2.inflation) == ("share", "share")
  assert dataclasses.replace(m2._make_config(), lr=m._make_config().lr) == m._make_config()   # (fit set the learning rate on `m`)
  (pX2, pY2), _ = m2.predict(X, verbose=False)
  assert np.array_equal(pX2.mean(), pX.mean()) and np.array_equal(pY2.mean(), pY.mean())
  sc = m.posterior_llk(X[:16], sample_shape=4)
  assert all(np.isfinite(v) for v in sc.values())
  with pytest.raises(ValueError):
    api.SCVI(api.RVmeta(n_genes, "zinbd", kwargs=dict(dispersion="nonsense")), **kw)
  # 'single': ONE trainable scalar for every cell and gene
  ms = api.SCVI(api.RVmeta(n_genes, "zinbd", name="rna", kwargs=dict(dispersion="single", inflation="single")), **kw)
  ms.fit(sco, epochs=6, batch_size=64, learning_rate=2e-3, verbose=False)
  ps = ms._engine.get_params()
  assert ps["out1/b"].shape == (1,) and ps["out2/b"].shape == (1,) and abs(float(ps["out1/b"][0])) > 1e-3 and "out1/W" not in ps
  pXs, _ = ms.predict(X[:40], verbose=False)
  assert np.allclose(pXs.distribution.count_distribution.disp, np.exp(ps["out1/b"][0]), rtol=1e-5) and np.allclose(pXs.distribution.logits, ps["out2/b"][0], rtol=1e-5, atol=1e-7)
  full = api.SCVI(api.RVmeta(n_genes, "nbd", kwargs=dict(dispersion="gene")), **kw)   # nbd: no gate plane, inflation is moot
  assert (full.dispersion, full.inflation) == ("share", "full")


def test_sisua_nbd_labels_are_nbd_heads(api):
  """vae.py:30 `RVmeta(adt_dim, 'onehot'/'nbd'/'nb', True, 'ADT')`: an 'nbd' label variable is a NegativeBinomialDisp head (mean /
  dispersion planes), not the 'nb' head (tests/test_singlecell_models.py:158-165 asserts that class for SISUA's second output)."""
  from sisua_amd import distributions as D
  sco = _sco()
  m = api.SISUA(outputs=sco.get_rv("transcriptomic", "zinbd"), labels=[sco.get_rv("proteomic", "nbd")],
                latents=api.RVmeta(8, "diag", True, "Latents"), encoder=api.NetConf([32]), decoder=api.NetConf([32]))
  assert m._make_config().labels == ((9, "nbd"),) and m.is_semi_supervised
  ds = sco.create_dataset(["transcriptomic", "proteomic"], labels_percent=0.5, batch_size=64, drop_remainder=True)
  m.fit(ds, metadata=sco, epochs=6, learning_rate=2e-3)
  (pX, pY), qZ = m.predict(sco.numpy()[:64], sample_shape=2, verbose=False)
  assert isinstance(pX.distribution.count_distribution, D.NegativeBinomialDisp) and isinstance(pY.distribution, D.NegativeBinomialDisp)
  assert pY.batch_shape == (2, 64) and qZ.sample(1).shape == (1, 64, 8)


@pytest.mark.parametrize("name,lk", [("vae", "zinb"), ("vae", "nb"), ("vae", "zinbd"), ("scvi", "zinbd"), ("sisua", "zinb")])
def test_lazy_predict_keeps_the_planes_on_the_device(api, name, lk):
  """predict(lazy=True) (the default for SingleCellOMIC inputs): the gene output is a device-side handle whose `.mean()` / `.variance()` /
  `.log_prob(x)` / `.mean_over_samples()` run as kernels (smx_predict_stat) and return only the statistic -- equal to what the eager,
  NumPy-backed distributions compute from the planes (float32 on the device against float64 on the host: 2e-5), the latents and the head
  outputs bit-identical, `materialize()` the eager result itself (what posterior.py:187-255 asks of predict's result)."""
  from sisua_amd import distributions as D
  sco = _sco(n=300)
  kw = dict(outputs=sco.get_rv("transcriptomic", lk), latents=api.RVmeta(8, "diag", True, "Latents"),
            encoder=api.NetConf([32], batchnorm=True, dropout=0.1), decoder=api.NetConf([32], batchnorm=True, dropout=0.1))
  if name == "sisua":
    kw["labels"] = [sco.get_rv("proteomic")]
  m = api.get_model(name)(**kw)
  omics = ["transcriptomic"] + (["proteomic"] if name == "sisua" else [])
  m.fit(sco.create_dataset(omics, labels_percent=0.5, batch_size=64, drop_remainder=True), metadata=sco, epochs=3, learning_rate=2e-3)
  X = sco.numpy()
  for S in ((), 3):
    eX, eZ = m.predict(X, sample_shape=S, batch_size=50, verbose=False)              # arrays: eager by default
    lX, lZ = m.predict(sco, sample_shape=S, batch_size=50, verbose=False)            # SingleCellOMIC: lazy by default
    e0, l0 = (eX[0], lX[0]) if isinstance(eX, tuple) else (eX, lX)
    assert isinstance(l0, D.LazyCountOutput) and not isinstance(e0, D.LazyCountOutput)
    assert l0.batch_shape == e0.batch_shape and l0.event_shape == e0.event_shape and l0.name == e0.name
    tol = dict(rtol=2e-5, atol=1e-6)
    assert np.allclose(l0.mean(), e0.mean(), **tol) and np.allclose(l0.variance(), e0.variance(), rtol=1e-4, atol=1e-5)
    assert l0.mean().dtype == np.float32 and l0.mean().shape == e0.mean().shape
    em = e0.mean() if S == () else e0.mean().mean(0)
    assert np.allclose(l0.mean_over_samples(), em, **tol) and l0.mean_over_samples().shape == (300, 120)
    # log_prob of the input counts and of another target (here: the counts + 1), summed over the genes
    assert np.allclose(l0.log_prob(), e0.log_prob(X), rtol=2e-5, atol=1e-3)
    assert np.allclose(l0.log_prob(X + 1.0), e0.log_prob(X + 1.0), rtol=2e-5, atol=1e-3)
    if l0.is_zero_inflated:   # 'imputed' of posterior.py:218-225: the count distribution without the zero-inflation wrapper
      lc, ec = l0.distribution.count_distribution, e0.distribution.count_distribution
      assert np.allclose(lc.mean(), ec.mean(), **tol) and (lc.mean() >= l0.mean() - 1e-6).all()
      assert np.allclose(lc.log_prob(X), ec.log_prob(X).sum(-1), rtol=2e-5, atol=1e-3)
    else:
      with pytest.raises(AttributeError):
        l0.count_distribution
    # a reused result array (no first-touch page faults), and the eager twin on demand: bit-identical planes
    buf = np.empty(l0.mean().shape, np.float32)
    assert l0.mean(out=buf) is buf and np.array_equal(buf, l0.mean())
    assert np.array_equal(l0.materialize().mean(), e0.mean())
    lz = lZ[0] if isinstance(lZ, (tuple, list)) else lZ
    ez = eZ[0] if isinstance(eZ, (tuple, list)) else eZ
    assert np.array_equal(lz.mean(), ez.mean())
    if isinstance(eX, tuple):
      assert len(lX) == len(eX) and np.array_equal(lX[1].mean(), eX[1].mean())
  # raw parameters a caller of the reference may touch on predict's result come from the eager twin
  inner = e0.distribution.count_distribution if l0.is_zero_inflated else e0.distribution
  attr = "total_count" if hasattr(inner, "total_count") else "mean_param" if hasattr(inner, "mean_param") else None
  if attr and hasattr(e0.distribution, attr):
    assert np.array_equal(getattr(l0, attr), getattr(e0.distribution, attr))
  with pytest.raises(AttributeError):
    l0.no_such_attribute
  # a restore of weights at the SAME optimiser step is a change of parameters too: the handle refuses
  import tempfile
  with tempfile.TemporaryDirectory() as tmp:
    m.save_weights(os.path.join(tmp, "w"))
    l0.mean()
    m.load_weights(os.path.join(tmp, "w"))
    with pytest.raises(RuntimeError):
      l0.mean()
  lX, _ = m.predict(sco, batch_size=50, verbose=False)
  l0 = lX[0] if isinstance(lX, tuple) else lX
  m.fit(sco.create_dataset(omics, batch_size=64, drop_remainder=True), metadata=sco, epochs=1)
  with pytest.raises(RuntimeError):   # the handle names the parameters it was made with
    l0.mean()
