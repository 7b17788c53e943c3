"""Model classes with the reference's operator surface.

`SingleCellModel`, `VAE`, `SISUA`, `SCVI`, `DeepCountAutoencoder`, `get_model`,
`get_all_models`, `load_model` keep the names, constructor arguments, method
signatures and error behaviour of sisua/models/{single_cell_model,vae,scvi,dca,__init__}.py
so that `sisua.train` / `sisua.analysis` style callers can switch packages.  Everything
numeric is delegated to libsisua_hip.so through `Engine`; this module holds no arithmetic
of the training step.
"""
from __future__ import annotations

import inspect
import os
import pickle
import time
import warnings
from typing import Dict, List, Optional, Sequence, Tuple, Union

from concurrent.futures import ThreadPoolExecutor

import numpy as np

from sisua_amd import distributions as D
from sisua_amd.config import ModelConfig, NetConf, RVmeta, init_params
from sisua_amd.data import BatchDataset, SingleCellOMIC, library_matrix
from sisua_amd.engine import Engine

__all__ = ["SingleCellModel", "VAE", "SISUA", "MISA", "SCALE", "SCVI", "DeepCountAutoencoder", "NetConf", "RVmeta", "get_model",
           "get_all_models", "load_model", "SCALAR"]

_OMIC_ORDER = ["transcriptomic", "proteomic", "celltype", "disease", "progenitor", "chromatin"]


class classproperty:

  def __init__(self, fn):
    self.fn = fn

  def __get__(self, obj, owner):
    return self.fn(owner)


def _flatten(x):
  if x is None:
    return []
  if isinstance(x, (list, tuple)):
    out = []
    for i in x:
      out += _flatten(i)
    return out
  return [x]


def _to_data(x, batch_size=64) -> BatchDataset:
  """single_cell_model.py:44-61: SingleCellOMIC -> its dataset; prepared dataset -> itself;
  raw ndarray(s) -> wrapped, further arrays assigned the next OMIC names in order."""
  if isinstance(x, SingleCellOMIC):
    return x.create_dataset(batch_size=batch_size)
  if isinstance(x, BatchDataset):
    return x
  arrs = _flatten(x)
  sco = SingleCellOMIC(np.asarray(arrs[0]))
  for arr, om in zip(arrs[1:], _OMIC_ORDER[1:]):
    sco.add_omic(om, np.asarray(arr))
  return sco.create_dataset(sco.omics, batch_size=batch_size, drop_remainder=True)


def _head_kind(rv: RVmeta, what: str):
  """(dim, kind) of a head on the decoder output for a label variable (`labels=`) or a further output variable (`outputs[1:]`):
  the count posteriors of RVmeta ('nb' / 'nbd' / 'zinb' / 'zinbd', vae.py:30), 'onehot', and MISA's mixtures."""
  if rv.posterior in ("nb", "nbd", "zinb", "zinbd"):
    return (rv.event_shape, rv.posterior)
  if rv.posterior in ("onehot", "categorical"):
    return (rv.event_shape, "onehot")
  if rv.posterior in ("mixnb", "mixnbd", "mixzinb", "mixzinbd"):     # MISA (vae.py:47-98): mixture of negative binomials per label dimension
    C = int(rv.kwargs.get("n_components", 2))
    if not 2 <= C <= 4:
      raise ValueError(f"mixture label heads are built for 2..4 components, given: {C}")
    return (rv.event_shape, f"mixzinb{C}" if (rv.kwargs.get("zero_inflated", False) or rv.posterior[:7] == "mixzinb") else f"mixnb{C}")   # (vae.py:76-84)
  if rv.posterior in ("mixgaussian", "mixgaus", "mixgauss", "mixnormal", "mdn", "mixtril", "mixfull", "mdntril"):
    # MISA's continuous labels (vae.py:86-92); 'mixtril' (the class's docstring example, vae.py:58) = covariance 'tril'
    C = int(rv.kwargs.get("n_components", 2))
    if not 2 <= C <= 4:
      raise ValueError(f"mixture label heads are built for 2..4 components, given: {C}")
    cov = rv.kwargs.get("covariance", "tril" if rv.posterior in ("mixtril", "mixfull", "mdntril") else "none")
    if cov in ("tril", "full"):
      if rv.event_shape > 64:
        raise ValueError("'mixtril' label heads are built for at most 64 label dimensions")
      return (rv.event_shape, f"mixtril{C}")
    if cov in ("none", "diag"):
      return (rv.event_shape, f"mixgauss{C}")
    raise ValueError(f"mixture-of-Gaussians label heads are built with covariance 'none' / 'diag' (independent dimensions) or 'tril' / 'full', given: {cov}")
  raise ValueError(f"{what} posterior '{rv.posterior}' is not built (supported: 'nb', 'nbd', 'zinb', 'zinbd', 'onehot', 'mixnb', 'mixgaussian', 'mixtril')")


class _Layer:
  """Named handle for `output_layers[i].name` / `posteriors[i]` (train.py:110; posterior.py:176)."""

  def __init__(self, rv: RVmeta):
    self.rv = rv
    self.name = rv.name
    self.event_shape = (rv.event_shape,)
    self.is_zero_inflated = rv.is_zero_inflated
    self.posterior = rv.posterior


class SingleCellModel:
  r"""Note: seed the model (`seed=`) for reproducible results; it keys the Philox
  streams of dropout and the reparameterisation noise."""

  _kind = "vae"

  def __init__(self,
               outputs: RVmeta,
               latents: RVmeta = RVmeta(10, "diag", True, "Latents"),
               encoder: NetConf = NetConf([64, 64], batchnorm=True, input_dropout=0.3),
               decoder: NetConf = NetConf([64, 64], batchnorm=True),
               log_norm=True,
               beta=1.0,
               name=None,
               **kwargs):
    frame_args = dict(outputs=outputs, latents=latents, encoder=encoder, decoder=decoder, log_norm=log_norm, beta=beta,
                      name=name, **kwargs)
    self.init_args = frame_args
    outs = [o.copy() for o in _flatten(outputs)]
    if not outs or outs[0].posterior not in ("nb", "zinb", "nbd", "zinbd", "mse"):
      raise ValueError("the first output must be a count distribution ('nb', 'zinb', 'nbd', 'zinbd') or the deterministic 'mse', "
                       f"given: {outs[0].posterior if outs else None}")
    self._outputs = outs
    self._labels = [l.copy() for l in _flatten(kwargs.pop("labels", None))]
    self._latents = [l.copy() for l in _flatten(latents)]
    self._encoder = _flatten(encoder)
    self._decoder = _flatten(decoder)[0]
    self._log_norm = bool(log_norm)
    self.beta = float(beta)
    self.alpha = float(kwargs.pop("alpha", 10.0))
    self.seed = int(kwargs.pop("seed", 8))
    self.clip_library = float(kwargs.pop("clip_library", 1e3))
    self.device = int(kwargs.pop("device", 0))
    for k in ("reduce_latent", "input_shape", "step", "path", "analytic", "gamma", "lamda"):
      kwargs.pop(k, None)
    self.name = name or type(self).__name__
    self.dataset = None
    self.metadata = dict()
    self._n_inputs = 1
    self.train_history: Dict[str, list] = {}
    self.valid_history: Dict[str, list] = {}
    self._engine: Optional[Engine] = None
    self._opt = dict(lr=1e-3, clipnorm=100.0)
    self._cfg = self._make_config()

  # ---- configuration --------------------------------------------------------------
  def _latent_activation(self):
    return "relu"

  def _make_config(self) -> ModelConfig:
    enc = self._encoder[0]
    labels = [_head_kind(rv, "label") for rv in self._labels]
    extras = [_head_kind(rv, "output") for rv in self._outputs[1:]]   # outputs[1:]: fully observed heads (weight 1, no label mask)
    if len(extras) + len(labels) > 4:
      raise ValueError("at most 4 heads (outputs[1:] + label variables) are built")
    encl = self._encoder[1].units if len(self._encoder) > 1 else (64,)
    return ModelConfig(model=self._kind, n_genes=self._outputs[0].event_shape, likelihood=self._outputs[0].posterior,
                       enc_units=tuple(enc.units), dec_units=tuple(self._decoder.units),
                       latent_dim=self._latents[0].event_shape, encl_units=tuple(encl), labels=tuple(labels),
                       batchnorm=bool(enc.batchnorm), dropout_enc=float(enc.dropout), dropout_dec=float(self._decoder.dropout),
                       input_dropout=float(enc.input_dropout), log_norm=self._log_norm, beta=self.beta, alpha=self.alpha,
                       latent_activation=self._latent_activation(), clip_library=self.clip_library,
                       lr=float(self._opt["lr"]), clipnorm=float(self._opt["clipnorm"]), seed=self.seed,
                       extra_outputs=tuple(extras), dispersion=str(getattr(self, "_dispersion", "full")), inflation=str(getattr(self, "_inflation", "full")),
                       n_components=int(getattr(self, "_n_components", 10)), covariance=str(getattr(self, "_covariance", "none")), latent_mixture=bool(getattr(self, "_latent_mixture", False)),
                       **getattr(self, "_ties", {}), **getattr(self, "_disc_cfg", {}))

  def _ensure_engine(self, max_batch: int) -> Engine:
    cfg = self._make_config()
    e = self._engine
    if e is not None and e.max_batch >= max_batch and e.cfg == cfg:
      return e
    state = None
    if e is not None:  # carry weights / optimiser state into the re-created engine
      state = self._get_state()
      e.close()
    self._cfg = cfg
    self._engine = Engine(cfg, max_batch=max(int(max_batch), 64), device=self.device, init=state is None)
    if state is not None:
      self._set_state(state)
    return self._engine

  def _get_state(self):
    e = self._engine
    return dict(params=e.get_params(0), m=e.get_params(2), v=e.get_params(3), bn=e.get_bn(), step=e.step)

  def _set_state(self, st):
    e = self._engine
    self._param_version = getattr(self, "_param_version", 0) + 1   # (lazy prediction handles are keyed by it: distributions.LazyCountOutput)
    e.set_params(st["params"], 0)
    e.set_params(st["m"], 2)
    e.set_params(st["v"], 3)
    e.set_bn(st["bn"])
    e.step = int(st["step"])

  # ---- reference attribute surface ---------------------------------------------------
  def set_metadata(self, sco: SingleCellOMIC):
    """Remember which dataset the model was fitted on and the variable names of each of its OMICs (single_cell_model.py:103-109)."""
    if not isinstance(sco, SingleCellOMIC):
      raise AssertionError(f"sco must be instance of SingleCellOMIC but given: {type(sco)}")
    self.metadata.update({omic: sco.get_var_names(omic) for omic in sco.omics})
    self.dataset = sco.name
    return self

  @property
  def log_norm(self):
    return self._log_norm

  @property
  def posteriors(self):
    return [_Layer(rv) for rv in self._outputs + self._labels]

  @property
  def output_layers(self):
    return [_Layer(rv) for rv in self._outputs]

  @property
  def labels(self):
    return list(self._labels)

  @property
  def latents(self):
    return [_Layer(rv) for rv in self._latents]

  @property
  def encoder(self):
    return self._encoder[0] if len(self._encoder) == 1 else list(self._encoder)

  @property
  def decoder(self):
    return self._decoder

  @property
  def is_zero_inflated(self):
    return self._outputs[0].is_zero_inflated

  @property
  def is_semi_supervised(self):
    return len(self._labels) > 0

  @property
  def is_fitted(self):
    return self.step > 0

  @property
  def step(self):
    return 0 if self._engine is None else self._engine.step

  @classproperty
  def id(cls):
    return "".join(c for c in cls.__name__ if c.isupper()).lower()

  # ---- fit -------------------------------------------------------------------------------
  def fit(self,
          train: Union[SingleCellOMIC, BatchDataset],
          valid: Union[SingleCellOMIC, BatchDataset] = None,
          metadata: SingleCellOMIC = None,
          **kwargs):
    r"""Train on the GPU.  Keyword arguments are the `train:` block of configs/base.yaml:45-62
    (optimizer, learning_rate, clipnorm, valid_freq, epochs, max_iter, earlystop_*,
    terminate_on_nan, checkpoint, log_tag, ...) plus `batch_size`."""
    # the first container among (train, valid, metadata) describes the dataset; prepared batches carry no names (single_cell_model.py:213-231)
    described_by = next((c for c in (train, valid, metadata) if isinstance(c, SingleCellOMIC)), None)
    if described_by is not None:
      self.set_metadata(described_by)
    if not self.metadata or self.dataset is None:
      raise RuntimeError("First time call `fit`, set the 'metadata' argument to a "
                         "SingleCellOMIC dataset to keep the dataset name and OMICs' "
                         "variables description.")
    batch_size = kwargs.pop("batch_size", 64)
    need = len(self._outputs) - 1 + len(self._labels)   # target omics beside the counts: outputs[1:], then the label variables

    def to_data(x):
      # a SingleCellOMIC handed to a model with several variables: its first 1 + need OMICs (the reference's create_dataset default is
      # the current OMIC alone, _single_cell_base.py:557-558, which such a model cannot train on); label variables stay unlabelled
      # (labels_percent = 0, the reference's default) unless the caller prepares the dataset with its own labels_percent
      if isinstance(x, SingleCellOMIC) and need > 0 and len(x.omics) > need:
        # a variable that names one of the container's OMICs takes that OMIC, the others the remaining OMICs in order
        rvs = self._outputs[1:] + self._labels
        taken = [x.omics[0]] + [rv.name for rv in rvs if rv.name in x.omics[1:]]
        rest = [o for o in x.omics[1:] if o not in taken]
        omics = [x.omics[0]] + [rv.name if rv.name in x.omics[1:] else rest.pop(0) for rv in rvs]
        for rv, om in zip(rvs, omics[1:]):
          if x.get_dim(om) != rv.event_shape:
            raise ValueError(f"variable '{rv.name}' has {rv.event_shape} dimensions but OMIC '{om}' has {x.get_dim(om)}")
        return x.create_dataset(omics, batch_size=batch_size)
      return _to_data(x, batch_size=batch_size)

    train = to_data(train)
    if valid is not None:
      valid = to_data(valid)
    return self._fit(train, valid, **kwargs)

  def _fit(self, train: BatchDataset, valid: Optional[BatchDataset], optimizer="adam", learning_rate=1e-3, clipnorm=100.0,
           valid_freq=500, valid_interval=0, epochs=500, max_iter=-1, sample_shape=(), logging_interval=2,
           earlystop_threshold=0.001, earlystop_progress_length=0, earlystop_patience=20, earlystop_min_epoch=-1,
           terminate_on_nan=True, checkpoint=None, allow_rollback=False, allow_none_gradients=False,
           track_gradient_norms=False, log_tag=None, verbose=False, distributed="auto",
           dp_batch="global", sync_bn=False, storage="f32", epochs_are_total=False, **ignored):
    r"""The training loop (odin Trainer under single_cell_model.py:213-236).

    The minibatch schedule is a pure function of the optimiser step count: iteration `it` visits batch
    `it % steps_per_epoch` of the shuffle order of epoch `it // steps_per_epoch`, and the Philox streams are keyed by
    `it` -- so a model restored with `load_weights` continues exactly where the interrupted run was (train.py:107-108).
    `epochs` counts the epochs of THIS call (`epochs_are_total=True`: of the whole schedule, used when an experiment resumes).  `train_history[k]` holds one value per epoch (the mean over the epoch's
    steps, kept on the device: `smx_metrics_history`), `valid_history['val_loss']` one per validation pass.

    Data parallel (one process per GPU; `distributed='auto'` reads RANK / LOCAL_RANK / WORLD_SIZE as set by
    `python -m torch.distributed.run`): rank r keeps the r-th contiguous 1/world of the training cells resident and
    draws `batch_size / world` of them per step (`dp_batch='global'`: the reference's global batch is preserved,
    SURVEY.md 8e) or `batch_size` (`dp_batch='per_rank'`); ONE RCCL all-reduce of the gradients per step;
    `sync_bn=True` adds SyncBatchNorm (statistics over the global batch, as the single process computes them);
    every rank evaluates the validation cells; only rank 0 runs `checkpoint`."""
    if str(optimizer).lower() != "adam":
      raise ValueError("only the 'adam' optimizer of configs/base.yaml is built")
    from sisua_amd import data as _data
    from sisua_amd.parallel import ControlPlane, attach_engine, env_rank_world
    self._opt = dict(lr=float(learning_rate), clipnorm=float(clipnorm or 0.0))
    n_lab = len(self._outputs) - 1 + len(self._labels)   # target arrays beside the counts: outputs[1:], then the label variables
    if len(train.arrays) < 1 + n_lab:
      raise ValueError(f"{type(self).__name__} needs {1 + n_lab} omics per batch, the dataset has {len(train.arrays)}")
    cp = None
    if distributed in ("auto", True):
      rank, local_rank, world = env_rank_world()
    elif hasattr(distributed, "rank") and hasattr(distributed, "world"):   # a ready control plane (tests: LocalControlPlane)
      cp, rank, local_rank, world = distributed, distributed.rank, self.device, distributed.world
    else:
      rank, local_rank, world = 0, 0, 1
    B, drop_rem, lo, hi = train.batch_size, train.drop_remainder, 0, train.n_obs
    if world > 1:
      if dp_batch == "global":
        if B % world:
          raise ValueError(f"batch_size {B} is not divisible by the {world} ranks (dp_batch='global')")
        B = B // world
      elif dp_batch != "per_rank":
        raise ValueError("dp_batch must be 'global' or 'per_rank'")
      lo, hi = _data.shard_range(train.n_obs, rank, world)
      drop_rem = True   # every rank runs the same number of equal-size steps
      self.device = local_rank
      cp = cp or ControlPlane(rank, world)
    e = self._ensure_engine(max(B, valid.batch_size if valid is not None else 1))
    if world > 1 and e.world != world:
      self._dp_mode = attach_engine(e, cp)
      self._dp_calibrated = False
    if world > 1:
      e.set_sync_bn(bool(sync_bn))
    n_tr = hi - lo
    if n_tr < 1 or (drop_rem and n_tr < B):
      raise ValueError(f"{n_tr} training cells cannot fill one batch of {B} (drop_remainder)")
    # train (this rank's shard) and validation cells live in ONE resident matrix; validation rows are offset
    X = train.arrays[0][lo:hi]
    labs = [train.arrays[1 + j][lo:hi] for j in range(n_lab)]
    lib, mask = train.library[lo:hi], train.mask[lo:hi]
    if valid is not None:
      X = np.concatenate([X, valid.arrays[0]], 0)
      labs = [np.concatenate([a, valid.arrays[1 + j]], 0) for j, a in enumerate(labs)]
      lib = np.concatenate([lib, valid.library], 0)
      mask = np.concatenate([mask, valid.mask], 0)
    e.upload(X, labs, lib, mask, cell_id_base=lo, storage=storage)   # Philox cell ids are GLOBAL: sharding-independent noise
    if cp is not None:
      cp.barrier()   # every rank's shard is resident before the first collective (the exchange's waits are bounded: no upload skew inside them)
    seed_r = train.seed + 7919 * rank

    def _prepare(ep):   # pure function of (seed, epoch): prepared on a host thread while the device runs the epoch before
      return _data.iter_batches(_data.epoch_order(n_tr, ep, train.shuffle, seed_r), B, drop_rem)

    if world > 1 and getattr(self, "_dp_mode", "loopback") != "loopback" and not getattr(self, "_dp_calibrated", True):
      # the exchange form of this job's steps, measured once per communicator on the job's own step and set identically on every rank
      # (parallel.calibrate_forms: trial steps from the current state, which is restored; SMX_DP_CALIBRATE=0 / SMX_DP_FORM switch it off)
      from sisua_amd.parallel import calibrate_forms
      ids = np.concatenate([np.asarray(b, np.int32) for b in _prepare(0) if len(b) == B] or [np.zeros(0, np.int32)])
      if ids.size >= B:
        self.dp_report = calibrate_forms(e, cp, self._dp_mode, np.resize(ids, 35 * B), B)
      self._dp_calibrated = True

    spe = n_tr // B if drop_rem else -(-n_tr // B)
    hist_t, hist_v = self.train_history, self.valid_history
    it = int(e.step)
    ep0 = it // spe
    it_end = (int(epochs) if epochs_are_total else ep0 + int(epochs)) * spe
    if max_iter and max_iter > 0:
      # a resumed run (`epochs_are_total`: the schedule counts from step 0 of the experiment, train.py) keeps the ORIGINAL
      # cap: max_iter is then an absolute step count as well, not `max_iter` more steps from the restored one
      it_end = min(it_end, int(max_iter) if epochs_are_total else it + int(max_iter))
    best, bad, t_log, stop = np.inf, 0, time.time(), False
    pool = ThreadPoolExecutor(max_workers=1)
    # One library call covers every step up to the next event on the host's side -- a validation pass, the end of the
    # schedule, a ragged batch, a progress line -- across epoch boundaries (at 26 steps per epoch a call per epoch left
    # the device idle for ~8 % of the time between calls).  The per-epoch history comes from the per-step scalars the
    # device keeps (smx_metrics_history), cut at the epoch boundaries afterwards.
    max_chunk = 4096 if not verbose else max(spe, 512)
    cache, pending = {}, {}

    def batches_of(ep):
      if ep not in cache:
        cache[ep] = pending.pop(ep).result() if ep in pending else _prepare(ep)
      return cache[ep]

    def prefetch(ep_from, ep_to):   # on the host thread, while the device runs the chunk just launched
      for ep in range(ep_from, ep_to):
        if ep not in cache and ep not in pending and ep * spe < it_end:
          pending[ep] = pool.submit(_prepare, ep)

    acc = {}
    first_call = True
    try:
      prefetch(ep0, ep0 + 2)
      while it < it_end and not stop:
        until_valid = valid_freq - (it % valid_freq) if valid_freq and valid_freq > 0 else it_end - it
        want = max(1, min(it_end - it, until_valid, max_chunk))
        if first_call and want > 2 * spe:
          # the call's FIRST chunk is two epochs: the device starts after two epochs' schedule instead of after the ~20 a valid_freq of 500 spans
          # (4-6 ms of a 50-epoch fit), and the host thread prepares the rest while it runs; later chunks are as long as before
          want = 2 * spe - (it % spe)
        first_call = False
        # gather `want` steps of equal batch size, walking over epoch boundaries
        parts, n, bs, j = [], 0, None, it
        while n < want:
          ep, pos = divmod(j, spe)
          if n > 0 and ep not in cache and ep in pending and not pending[ep].done():
            break   # (the host thread is still preparing that epoch: launch what is gathered instead of waiting for it)
          b = batches_of(ep)
          take = 0
          if drop_rem:   # every batch has B cells: the run's length is arithmetic
            bs = B
            take = min(spe - pos, want - n)
          while not drop_rem and pos + take < spe and n + take < want and (bs is None or len(b[pos + take]) == bs):
            if bs is None:
              bs = len(b[pos])   # a ragged last batch (drop_remainder=False, the reference's default for fit(SingleCellOMIC)) is a step of its own
            take += 1
          if take == 0:
            break
          parts += b[pos:pos + take]
          n += take
          j += take
          if pos + take < spe:   # stopped inside the epoch (ragged batch ahead or enough steps)
            break
        order = np.concatenate(parts).astype(np.int32)
        first_ep, last_ep = it // spe, (it + n - 1) // spe
        for ep in [k for k in cache if k < first_ep]:
          del cache[ep]
        m = e.train_steps(order, n, bs, metrics=True)   # (eager launches: a captured hipGraph of the step replays 11 % slower)
        nxt = min(it_end - (it + n), valid_freq if valid_freq and valid_freq > 0 else max_chunk, max_chunk)
        prefetch(last_ep, last_ep + 2 + max(nxt, 0) // max(spe, 1))   # what the next call will walk
        h = e.metrics_history(n)
        # per-epoch means: cut the per-step scalars at the epoch boundaries (every key at once: one reduction per epoch, not one per key and epoch)
        keys = list(h)
        H = np.stack([np.asarray(h[k], np.float32) for k in keys])   # [key][step]
        s0 = 0
        while s0 < n:
          ep_here = (it + s0) // spe
          s1 = min(n, (ep_here + 1) * spe - it)
          done = it + s1 == (ep_here + 1) * spe   # the epoch is complete: one value per epoch, the mean over its steps
          if done and not acc:
            for k, mean in zip(keys, H[:, s0:s1].mean(axis=1)):
              hist_t.setdefault(k, []).append(float(mean))
          else:
            for i_k, k in enumerate(keys):
              acc.setdefault(k, []).append(H[i_k, s0:s1])
            if done:
              for k, seg in acc.items():
                hist_t.setdefault(k, []).append(float(np.mean(np.concatenate(seg))))
              acc = {}
          s0 = s1
        it += n
        epoch = (it - 1) // spe
        if m["nan_flag"] or not np.isfinite(h["loss"]).all():
          if terminate_on_nan:
            raise FloatingPointError(f"non-finite loss or gradient norm at iteration {it}")
          warnings.warn(f"non-finite loss or gradient norm at iteration {it}")
        if verbose and rank == 0 and time.time() - t_log > logging_interval:
          print(f"[{log_tag or self.name}] it {it} epoch {epoch} loss {m['loss']:.4f} nllk_x {m['nllk_x']:.4f} kl {m['kl']:.4f}")
          t_log = time.time()
        if valid is not None and valid_freq and it % valid_freq == 0:
          vl = self._validate(e, valid, n_tr, cp)
          hist_v.setdefault("val_loss", []).append(vl)
          improved = vl < best * (1.0 - float(earlystop_threshold)) if np.isfinite(best) else True
          if vl < best:
            best = vl
            self._checkpoint(checkpoint, rank, cp, e)
          bad = 0 if improved else bad + 1
          if earlystop_patience and bad >= int(earlystop_patience) and epoch >= int(earlystop_min_epoch):
            stop = True
      if acc:   # the schedule ended (max_iter, early stop) inside an epoch: its mean over the steps that ran
        for k, seg in acc.items():
          hist_t.setdefault(k, []).append(float(np.mean(np.concatenate(seg))))
    finally:
      for f in pending.values():
        f.cancel()
      pool.shutdown(wait=True)
    if valid is not None and not hist_v.get("val_loss"):
      hist_v.setdefault("val_loss", []).append(self._validate(e, valid, n_tr, cp))
    if valid is None:
      self._checkpoint(checkpoint, rank, cp, e)
    if cp is not None:
      if cp.world > 1:
        #  flag opt_shard: the heads' Adam moments are whole again on every rank when fit returns -- a later save_weights /
        # _get_state reads them without being a collective (every rank is here; a no-op when nothing is stale)
        e.opt_gather()
      cp.barrier()
    return self

  @staticmethod
  def _checkpoint(checkpoint, rank, cp, engine=None):
    if checkpoint is not None and engine is not None and cp is not None and cp.world > 1:
      engine.opt_gather()   # (flag opt_shard: the heads' Adam moments live sliced over the ranks between checkpoints; a collective, every rank is here)
    if checkpoint is not None and rank == 0:   # replicas are identical: one writer
      checkpoint()
    if cp is not None:
      cp.barrier()

  @staticmethod
  def _validate(e: Engine, valid: BatchDataset, offset: int, cp=None) -> float:
    """Eval-mode ELBO of the validation cells.  Data parallel: every rank scores all of them (its own draw of the
    latent noise) and the ranks' values are averaged over the control plane, so that every rank takes the same
    early-stopping / checkpoint decisions."""
    tot, n = 0.0, 0
    for ids in valid.epoch_batches(0):
      if len(ids) == 0:
        continue
      m = e.eval_step((ids + offset).astype(np.int32))
      tot += m["loss"] * len(ids)
      n += len(ids)
    vl = tot / max(n, 1)
    if cp is not None and cp.world > 1:
      vl = float(cp.sum_array(np.array([vl]))[0]) / cp.world
    return vl

  # ---- inference ------------------------------------------------------------------------------
  def _latent_dists(self, out, sl=slice(None)):
    cfg = self._cfg
    if cfg.stochastic:
      qz = D.MultivariateNormalDiag(out["z_mean"][sl], out["z_scale"][sl], name=self._latents[0].name or "Latents")
    else:
      qz = D.Deterministic(out["z_sample"][sl], name=self._latents[0].name or "Latents")
    if cfg.model == "scvi":
      ql = D.Independent(D.Normal(out["l_mean"][sl][:, None], out["l_scale"][sl][:, None], name="Library"), 1)
      return [qz, ql]
    return qz

  def _output_dists(self, xp_list, yp_list, stacked=False, heads_only=False):
    """xp_list: per-MC-sample x_params [k,B,G]; yp_list: per-sample list of label raw outputs.
    stacked=True: xp_list is one array [S,k,B,G] and yp_list one array [S,B,w] per label head (views, no copies).
    heads_only: the distributions of the heads alone, as a list (the gene output is handled by the caller)."""
    cfg = self._cfg
    if stacked:
      planes = None if heads_only else [xp_list[:, c] for c in range(cfg.k)]
      stack = None
    else:
      stack = (lambda a: a[0]) if len(yp_list if heads_only else xp_list) == 1 else (lambda a: np.stack(a, 0))
      planes = None if heads_only else [stack([xp[c] for xp in xp_list]) for c in range(cfg.k)]
    outs = [] if heads_only else [D.count_distribution(cfg.likelihood, planes, self._outputs[0].name or "transcriptomic", activated=cfg.model == "scvi")]
    n_extra = len(cfg.extra_outputs)
    for j, (P, kind) in enumerate(cfg.head_labels):
      raw = yp_list[j] if stacked else stack([yp[j] for yp in yp_list])
      nm = (self._outputs[1 + j].name or f"output{1 + j}") if j < n_extra else (self._labels[j - n_extra].name or f"label{j - n_extra}")
      if kind in ("nb", "nbd", "zinb", "zinbd"):   # planes of width P, as for the gene output
        outs.append(D.count_distribution(kind, [raw[..., c * P:(c + 1) * P] for c in range(3 if kind[0] == "z" else 2)], nm, activated=False))
      elif kind.startswith("mixnb"):
        C = int(kind[-1])
        pl = np.reshape(raw, raw.shape[:-1] + (3 * C, P))     # planes: C mixture logits | C log total_counts | C logits
        outs.append(D.Independent(D.MixtureNegativeBinomial(pl[..., :C, :], np.exp(pl[..., C:2 * C, :]), pl[..., 2 * C:, :]), 1, name=nm))
      elif kind.startswith("mixzinb"):
        C = int(kind[-1])
        pl = np.reshape(raw, raw.shape[:-1] + (4 * C, P))     # planes: C mixture logits | C log total_counts | C logits | C gate logits
        outs.append(D.Independent(D.MixtureNegativeBinomial(pl[..., :C, :], np.exp(pl[..., C:2 * C, :]), pl[..., 2 * C:3 * C, :],
                                                            gate_logits=pl[..., 3 * C:, :]), 1, name=nm))
      elif kind.startswith("mixgauss"):
        C = int(kind[-1])
        pl = np.reshape(raw, raw.shape[:-1] + (3 * C, P))     # planes: C mixture logits | C locations | C raw scales
        scale = np.logaddexp(0.0, pl[..., 2 * C:, :].astype(np.float64) + np.log(np.expm1(1.0)))   # softplus1
        outs.append(D.Independent(D.MixtureNormal(pl[..., :C, :], pl[..., C:2 * C, :], scale), 1, name=nm))
      elif kind.startswith("mixtril"):
        C = int(kind[-1])
        pl = np.reshape(raw, raw.shape[:-1] + (C * (2 + P), P)).astype(np.float64)   # planes: C logits (column 0) | C locations | C x P columns of L
        cols = np.reshape(pl[..., 2 * C:, :], raw.shape[:-1] + (C, P, P))              # [..., c, j, p] = L_c[p][j]
        L = np.tril(np.swapaxes(cols, -1, -2), -1)
        dg = np.logaddexp(0.0, np.einsum("...pp->...p", np.swapaxes(cols, -1, -2))) + 1e-5   # softplus + TFP's diag_shift
        L = L + dg[..., :, None] * np.eye(P)
        outs.append(D.MixtureMultivariateNormalTriL(pl[..., :C, 0], pl[..., C:2 * C, :], L, name=nm))
      else:
        outs.append(D.OneHotCategorical(raw, name=nm))
    if heads_only:
      return outs
    return outs[0] if len(outs) == 1 else tuple(outs)

  def __call__(self, inputs=None, library=None, mask=None, training=None, sample_shape=(), **kwargs):
    arrs = _flatten(inputs)
    x = np.ascontiguousarray(arrs[0], dtype=np.float32)
    if self._cfg.model == "scvi" and library is None:
      library = library_matrix(x)
    n = int(np.prod(sample_shape)) if np.size(sample_shape) else 0
    e = self._ensure_engine(x.shape[0])
    if n > 1:   # every draw in one call: the encoders run once, the draws re-sample the latents and decode
      o = e.forward_samples(n, x=x, library=library)
      pX = self._output_dists(o["x_params"], o["y_params"], stacked=True)
      first = dict(o, z_sample=o["z_sample"][0])
      if "l_sample" in o:
        first["l_sample"] = o["l_sample"][0]
      return pX, self._latent_dists(first)
    outs = [e.forward(x=x, library=library, sample_index=s) for s in range(max(n, 1))]
    pX = self._output_dists([o["x_params"] for o in outs], [o["y_params"] for o in outs])
    return pX, self._latent_dists(outs[0])

  def encode(self, inputs, library=None, training=None, mask=None, sample_shape=(), **kwargs):
    r"""log1p + encoder network + latent posterior (single_cell_model.py:119-139); SCVI
    returns [q(z|x), q(l|x)] (scvi.py:88-106)."""
    arrs = _flatten(inputs)[:self._n_inputs]
    x = np.ascontiguousarray(arrs[0], dtype=np.float32)
    if self._cfg.model == "scvi" and library is None:
      library = library_matrix(x)
    e = self._ensure_engine(x.shape[0])
    out = e.forward(x=x, library=library, want_x_params=False)
    return self._latent_dists(out)

  def decode(self, latents, training=None, mask=None, sample_shape=(), **kwargs):
    r"""Decoder network + output distributions (single_cell_model.py:141-151; scvi.py:108-171).
    `latents`: array [B,D] or latent distribution(s) (a distribution is sampled, as TFP's
    tensor coercion does)."""
    lat = _flatten(latents)

    def val(v):
      return v.sample(seed=self.seed) if isinstance(v, D.Distribution) else np.asarray(v)

    z = val(lat[0]).astype(np.float32)
    l = val(lat[1]).astype(np.float32).reshape(-1) if len(lat) > 1 else None
    e = self._ensure_engine(z.shape[0])
    out = e.decode(z, l)
    return self._output_dists([out["x_params"]], [out["y_params"]])

  def predict(self, inputs, sample_shape=(), batch_size=32, verbose=True, device="GPU", lazy=None):
    r"""Predict on minibatches then return a single distribution by concatenation.

    lazy : the gene output as a device-side handle (`distributions.LazyCountOutput`): its parameter planes -- 24 KB per cell and
      draw at 1998 genes -- stay off the host, `.mean()` / `.variance()` / `.log_prob(x)` run as kernels and return only what
      is asked.  Default: on for `SingleCellOMIC` inputs with cells in their own order (what `Posterior` hands in,
      sisua/analysis/posterior.py:172-182), off for arrays and prepared datasets (eager NumPy-backed distributions).

    Return:
      X : `Distribution` or tuple of `Distribution` (multiple outputs)
      Z : `Distribution` or tuple of `Distribution` (multiple latents)
    """
    assert device in ("CPU", "GPU"), f"Only support device CPU or GPU, but given: {device}"
    if device == "CPU":
      raise RuntimeError("sisua_amd has no CPU path; predict runs on the MI355X")
    if lazy is None:
      lazy = isinstance(inputs, SingleCellOMIC)
    if lazy and isinstance(inputs, SingleCellOMIC):
      inputs = inputs.numpy()   # (cells in their own order: the one-call path)
    ds = _to_data(inputs, batch_size=batch_size) if not isinstance(inputs, BatchDataset) else inputs
    if not isinstance(inputs, (BatchDataset, SingleCellOMIC)):
      ds.drop_remainder = False
    ds.shuffle = 0 if isinstance(inputs, (np.ndarray, list, tuple)) else ds.shuffle
    if ds.shuffle == 0 and ds.n_obs > 0:
      # cells in their own order: the minibatch loop runs inside the library (smx_predict) and every result is written
      # once, straight into the arrays the returned distributions hold -- no per-batch arrays, no concatenation
      n_use = ds.n_obs if not ds.drop_remainder else (ds.n_obs // ds.batch_size) * ds.batch_size
      if n_use > 0:
        return self._predict_all(ds.arrays[0][:n_use], ds.library[:n_use], sample_shape, ds.batch_size, lazy=bool(lazy))
    if lazy:
      raise ValueError("lazy=True needs the cells in their own order (an array, a SingleCellOMIC, or a dataset with shuffle=0)")
    X, Z = [], []
    for data in ds:
      pX, qZ = self(**data, training=False, sample_shape=sample_shape)
      X.append(pX)
      Z.append(qZ)
    multi_x = isinstance(X[0], (tuple, list))
    first = X[0][0] if multi_x else X[0]
    axis = 0 if len(first.batch_shape) == 1 else 1
    if multi_x:
      Xc = tuple(D.concat_distributions([x[i] for x in X], axis=axis, name=self.posteriors[i].name) for i in range(len(X[0])))
    else:
      Xc = D.concat_distributions(X, axis=axis, name=self.posteriors[0].name)
    if isinstance(Z[0], (tuple, list)):
      Zc = tuple(D.concat_distributions([z[i] for z in Z], axis=0) for i in range(len(Z[0])))
    else:
      Zc = D.concat_distributions(Z, axis=0)
    return Xc, Zc

  def _predict_all(self, x, library, sample_shape, batch_size, lazy=False):
    n = int(np.prod(sample_shape)) if np.size(sample_shape) else 0
    # (room for super-batches: the library decodes several minibatches per pass when its max_batch allows, same numbers)
    e = self._ensure_engine(max(min(int(batch_size), x.shape[0]), 512 if x.shape[0] >= 1024 else 1))
    lib = library if self._cfg.model == "scvi" else None
    if lib is not None and lib.shape[1] != 2:
      lib = library_matrix(x)
    lazy = lazy and self._cfg.likelihood != "mse"
    B = min(int(batch_size), e.max_batch)
    o = e.predict(x, library=lib, n_samples=max(n, 1), batch=B, want_x_params=not lazy)
    if lazy:
      # the gene output stays a handle (its planes never leave the device); the small head outputs and the latents are eager as always
      heads = self._output_dists(None, o["y_params"], stacked=True, heads_only=True) if n > 1 else \
          self._output_dists(None, [[y[0] for y in o["y_params"]]], heads_only=True)
      px = D.LazyCountOutput(self, np.ascontiguousarray(x, dtype=np.float32), lib, n, B, self._outputs[0].name or "transcriptomic")
      pX = (px,) + tuple(heads) if heads else px
    elif n > 1:
      pX = self._output_dists(o["x_params"], o["y_params"], stacked=True)
    else:
      pX = self._output_dists([o["x_params"][0]], [[y[0] for y in o["y_params"]]])
    first = dict(o, z_sample=o["z_sample"][0])
    if "l_sample" in o:
      first["l_sample"] = o["l_sample"][0]
    qZ = self._latent_dists(first)
    return pX, (tuple(qZ) if isinstance(qZ, list) else qZ)

  def marginal_log_prob(self, inputs=None, library=None, mask=None, sample_shape=100, batch_size=128, **kwargs):
    r"""Importance-weighted estimate of log p(x) with `sample_shape` posterior draws
    (Posterior.cal_marginal_llk, analysis/posterior.py:941-976), computed on the GPU
    (`smx_marginal_llk`: encoder once; the draws of a batch then go through the decoder and the output head --
    fused with the forward-only likelihood -- stacked as rows of one pass, up to 16 384 rows at a time, with a running
    log-sum-exp across passes; scvi materialises its raw planes per pass and normalises row-locally).
    Returns (mllk[B], {output name: mean_s log p(x|z_s) [B]})."""
    arrs = _flatten(inputs)
    if len(self._outputs) > 1:
      return self._joint_marginal_log_prob(arrs, library, sample_shape, batch_size)
    x = np.ascontiguousarray(arrs[0], dtype=np.float32)
    S = int(np.prod(sample_shape)) if np.size(sample_shape) else 1
    if self._cfg.model == "scvi" and library is None:
      library = library_matrix(x)
    e = self._ensure_engine(min(int(batch_size), x.shape[0]))
    B = min(e.max_batch, x.shape[0])
    mllk, llk = [], []
    for s0 in range(0, x.shape[0], B):
      sl = slice(s0, s0 + B)
      a, b = e.marginal_llk(x=x[sl], library=None if library is None else library[sl], n_samples=S)
      mllk.append(a)
      llk.append(b)
    return np.concatenate(mllk), {self._outputs[0].name or "transcriptomic": np.concatenate(llk)}

  def _joint_marginal_log_prob(self, arrs, library, sample_shape, batch_size):
    r"""Several OUTPUT variables: the importance-weighted estimate of the joint log p(x, y_1, ...), every draw's weight
    log w_s = sum_o log p(x_o | z_s) + log p(z_s) - log q(z_s | x).  The gene output's log-likelihood of every draw comes from the device
    (smx_predict_stat: [S, N], the planes never leave it), the heads' raw outputs and the latent draws from the same passes (smx_predict,
    same Philox draws); the few scalars per draw are combined here.  Returns (mllk [N], {output name: mean_s log p(x_o | z_s) [N]})."""
    cfg = self._cfg
    n_out = len(self._outputs)
    if cfg.latent_mixture:
      #  a draw's weight needs log q_mix(z_s | x) of the mixture-density posterior: predict() reports the mixture's mean and
      # scale, from which a single Gaussian's density is NOT that -- refused rather than estimated with the wrong weights.  The
      # one-output call computes the per-draw term on the device (smx_marginal_llk).
      raise NotImplementedError("marginal_log_prob of several outputs under SCALE's mixture-density posterior (mixture='posterior') is not "
                                "built: the joint estimate needs the per-draw log q_mix(z | x) from the device")
    if len(arrs) < n_out:
      raise ValueError(f"marginal_log_prob of this model needs the {n_out} output variables' arrays as inputs=[x, y, ...]")
    x = np.ascontiguousarray(arrs[0], dtype=np.float32)
    ys = [np.ascontiguousarray(a, dtype=np.float32) for a in arrs[1:n_out]]
    S = int(np.prod(sample_shape)) if np.size(sample_shape) else 1
    if cfg.model == "scvi" and library is None:
      library = library_matrix(x)
    e = self._ensure_engine(min(int(batch_size), x.shape[0]))
    B = min(e.max_batch, int(batch_size))
    o = e.predict(x, library=library if cfg.model == "scvi" else None, n_samples=S, batch=B, want_x_params=False)
    llk = {self._outputs[0].name or "transcriptomic": e.predict_stat(x, "log_prob", library=library if cfg.model == "scvi" else None,
                                                                    n_samples=S, batch=B).astype(np.float64)}
    heads = self._output_dists(None, o["y_params"], stacked=True, heads_only=True)
    for j, yj in enumerate(ys):   # (the observed heads come first)
      llk[self._outputs[1 + j].name or f"output{1 + j}"] = np.asarray(heads[j].log_prob(yj), np.float64)
    logw = sum(llk.values())
    if cfg.stochastic:
      z, mu, sg = o["z_sample"].astype(np.float64), o["z_mean"].astype(np.float64), o["z_scale"].astype(np.float64)
      eps = (z - mu) / sg
      if cfg.model == "scale" and not cfg.latent_mixture:
        # SCALE (scale.py:13-49): the prior is the trainable mixture -- log p_mix(z_s) - log q(z_s | x), minus the Monte-Carlo KL of the draw
        log_q = (-0.5 * eps ** 2 - np.log(sg) - 0.5 * np.log(2 * np.pi)).sum(-1)
        logw = logw + self._mixture_prior_log_prob(z) - log_q
      else:
        logw = logw + (-0.5 * z ** 2 + 0.5 * eps ** 2 + np.log(sg)).sum(-1)
    if cfg.model == "scvi":
      l, ml, sl = o["l_sample"].astype(np.float64), o["l_mean"].astype(np.float64), o["l_scale"].astype(np.float64)
      lib = np.asarray(library, np.float64)
      mp, sp = lib[:, 0], np.sqrt(lib[:, 1])
      logw = logw + (-0.5 * ((l - mp) / sp) ** 2 - np.log(sp) + 0.5 * ((l - ml) / sl) ** 2 + np.log(sl))
    mx = logw.max(0)
    mllk = mx + np.log(np.exp(logw - mx).sum(0)) - np.log(S)
    return mllk.astype(np.float32), {k: v.mean(0).astype(np.float32) for k, v in llk.items()}

  def _mixture_prior_log_prob(self, z):
    r"""log p(z) under SCALE's trainable mixture prior for z [..., D] (float64): p(z) = sum_c softmax(a)_c N(z; m_c, S_c), S_c diagonal with
    s = softplus1(raw) or L_c L_c^T with a softplus diagonal + 1e-5 (covariance = 'tril': tfp's FillScaleTriL) -- the prior kernel's
    arithmetic (smx_kernels.hip: scale_prior_*), from the model's current parameters."""
    cfg = self._cfg
    pr = {k: np.asarray(v, np.float64) for k, v in self._engine.get_params().items() if k.startswith("prior/")}
    a, m_c = pr["prior/logits"], pr["prior/loc"]
    C, D = m_c.shape
    log_pi = a - (a.max() + np.log(np.exp(a - a.max()).sum()))
    zz = np.asarray(z, np.float64)[..., None, :]                                            # [..., 1, D]
    if cfg.scale_tril:
      Lraw = pr["prior/scale"].reshape(C, D, D)
      dg = np.logaddexp(0.0, np.einsum("cpp->cp", Lraw)) + 1e-5
      L = np.tril(Lraw, -1)
      L[:, np.arange(D), np.arange(D)] = dg
      diff = zz - m_c                                                                        # [..., C, D]
      u = np.zeros_like(diff)
      for p_ in range(D):   # forward substitution, component by component
        u[..., p_] = (diff[..., p_] - (L[:, p_, :p_] * u[..., :p_]).sum(-1)) / dg[:, p_]
      comp = log_pi - 0.5 * (u * u).sum(-1) - np.log(dg).sum(1) - 0.5 * D * np.log(2 * np.pi)
    else:
      s_c = np.logaddexp(0.0, pr["prior/scale"] + np.log(np.expm1(1.0)))                     # softplus1
      comp = log_pi + (-0.5 * ((zz - m_c) / s_c) ** 2 - np.log(s_c) - 0.5 * np.log(2 * np.pi)).sum(-1)
    cm = comp.max(-1, keepdims=True)
    return (cm + np.log(np.exp(comp - cm).sum(-1, keepdims=True)))[..., 0]

  def posterior_llk(self, corrupted, original=None, library=None, sample_shape=10, batch_size=128):
    r"""The four scores of `Posterior.cal_llk` (analysis/posterior.py:919-938) on the GPU
    (`smx_score_llk`): cells are encoded from `corrupted`, `sample_shape` posterior draws are decoded, and
    the 'reconstructed' (output distribution) and 'imputed' (count distribution without zero inflation)
    likelihoods of the original and of the corrupted counts are reduced as
    mean_cells(logsumexp_draws - log n_draws).  Returns the dict with the reference's keys."""
    x_cor = np.ascontiguousarray(_flatten(corrupted)[0], dtype=np.float32)
    x_org = x_cor if original is None else np.ascontiguousarray(_flatten(original)[0], dtype=np.float32)
    S = int(np.prod(sample_shape)) if np.size(sample_shape) else 1
    if self._cfg.model == "scvi" and library is None:
      library = library_matrix(x_cor)
    e = self._ensure_engine(min(int(batch_size), x_cor.shape[0]))
    B = min(e.max_batch, x_cor.shape[0])
    parts = []
    for s0 in range(0, x_cor.shape[0], B):
      sl = slice(s0, s0 + B)
      parts.append(e.score_llk([x_org[sl], None], x=x_cor[sl], library=None if library is None else library[sl],
                               n_samples=S))
    sc = np.concatenate(parts, axis=2).mean(axis=2)   # [target, dist]
    name = self._outputs[0].name or "transcriptomic"
    return {f"llk_{name}_imp_org": float(sc[0, 1]), f"llk_{name}_imp_cor": float(sc[1, 1]),
            f"llk_{name}_rec_cor": float(sc[1, 0]), f"llk_{name}_rec_org": float(sc[0, 0])}

  # ---- evaluation hand-off -----------------------------------------------------------------------
  def create_posterior(self, test_sco: SingleCellOMIC = None, **kwargs):
    r"""The reference builds `sisua.analysis.Posterior(scm=self, sco=test, ...)`
    (single_cell_model.py:247-281).  The analysis package is outside this build
    (SURVEY.md section 2 row 13); when the reference's `sisua` is importable its
    Posterior is used with this model, otherwise a clear error is raised."""
    if not self.is_fitted:
      raise RuntimeError("fit() must be called before creating Posterior.")
    try:
      from sisua.analysis.posterior import Posterior  # type: ignore
    except Exception as err:
      raise NotImplementedError("sisua.analysis.Posterior is not part of sisua_amd; install the reference "
                                "package to score this model (predict / marginal_log_prob are provided)") from err
    return Posterior(scm=self, sco=test_sco, **kwargs)

  # ---- checkpoints --------------------------------------------------------------------------------
  def save_weights(self, filepath, overwrite=True):
    r"""Weights + optimiser state + BN statistics + step in `<filepath>.npz`, and the reference's sidecar
    `<filepath>.metamodel` = pickle([class_name, dataset, metadata, init_kwargs]) (single_cell_model.py:295-306).
    The sidecar holds PLAIN records only (dicts / lists / numpy arrays: `RVmeta` / `NetConf` arguments are stored as
    `{"__record__": "RVmeta", ...}` and rebuilt on load), so reading it needs neither this package's classes nor
    odin's.  Both files are written to a temporary name and renamed (the weights last): an interrupted save leaves
    the previous checkpoint intact (the TF checkpoint writer of the reference is atomic in the same way)."""
    if self._engine is None:
      self._ensure_engine(64)
    if not overwrite and os.path.exists(f"{filepath}.npz"):
      raise FileExistsError(f"{filepath}.npz")
    st = self._get_state()
    flat = {f"p/{k}": v for k, v in st["params"].items()}
    flat.update({f"m/{k}": v for k, v in st["m"].items()})
    flat.update({f"v/{k}": v for k, v in st["v"].items()})
    for i, b in st["bn"].items():
      flat[f"bn/{i}/moving_mean"], flat[f"bn/{i}/moving_var"] = b["moving_mean"], b["moving_var"]
    flat["step"] = np.array(st["step"], dtype=np.int64)
    d = os.path.dirname(os.path.abspath(filepath))
    os.makedirs(d, exist_ok=True)
    tmp = f"{filepath}.tmp{os.getpid()}"
    with open(f"{tmp}.metamodel", "wb") as f:
      pickle.dump([self.__class__.__name__, self.dataset, _to_plain(self.metadata), _to_plain(dict(self.init_args))], f)
    os.replace(f"{tmp}.metamodel", f"{filepath}.metamodel")
    with open(f"{tmp}.npz", "wb") as f:
      np.savez(f, **flat)
    os.replace(f"{tmp}.npz", f"{filepath}.npz")
    return self

  def load_weights(self, filepath, raise_notfound=False, verbose=False):
    r"""Load all the saved weights at given path (single_cell_model.py:283-293)."""
    if not os.path.exists(f"{filepath}.npz"):
      if raise_notfound:
        raise FileNotFoundError(f"Cannot find saved weights at path: {filepath}")
      return self
    z = np.load(f"{filepath}.npz")
    if self._engine is None:
      self._ensure_engine(64)
    bn = {}
    for k in z.files:
      if k.startswith("bn/"):
        _, i, nm = k.split("/")
        bn.setdefault(int(i), {})[nm] = z[k]
    self._set_state(dict(params={k[2:]: z[k] for k in z.files if k.startswith("p/")},
                         m={k[2:]: z[k] for k in z.files if k.startswith("m/")},
                         v={k[2:]: z[k] for k in z.files if k.startswith("v/")}, bn=bn, step=int(z["step"])))
    metamodel_path = f"{filepath}.metamodel"
    if os.path.exists(metamodel_path):
      class_name, dataset, metadata, kwargs = read_metamodel(metamodel_path)
      assert class_name == self.__class__.__name__
      self.dataset = dataset
      self.metadata = metadata
    if verbose:
      print(f"Loaded weights from {filepath} (step {self.step})")
    return self

  def plot_learning_curves(self, path=None, **kwargs):
    return dict(train=self.train_history, valid=self.valid_history)

  def __repr__(self):
    c = self._cfg
    return (f"<{type(self).__name__} id={self.id} G={c.n_genes} {c.likelihood} enc={list(c.enc_units)} D={c.latent_dim} "
            f"dec={list(c.dec_units)} labels={list(c.labels)} step={self.step}>")


class VAE(SingleCellModel):
  r"""Variational Auto Encoder (sisua/models/vae.py:15-16)."""
  _kind = "vae"


class SISUA(SingleCellModel):
  r"""Multi-task SemI-SUpervised Autoencoder (sisua/models/vae.py:19-44):
  transcriptomic zero-inflated negative binomial, proteomic negative binomial or
  one-hot categorical label heads weighted by `alpha`, diagonal Gaussian latent.

      RVmeta(rna_dim, 'zinbd'/'zinb', projection=True, name='RNA')
      RVmeta(adt_dim, 'onehot'/'nbd'/'nb', True, 'ADT')
  """
  _kind = "sisua"

  def __init__(self, outputs, labels, **kwargs):
    super().__init__(outputs=outputs, labels=labels, **kwargs)
    self._n_inputs = 1


class MISA(SISUA):
  r"""MIxture of labels for Semi-supervised Autoencoder (sisua/models/vae.py:47-98): SISUA whose label heads are
  mixture distributions.  Discrete labels (ADT counts) become `n_components`-component mixtures of negative
  binomials per label dimension ('mixnb'); other label posteriors are converted with the reference's warning.
  Built: 'mixnb' with 2..4 components, zero-inflated or not (`zero_inflated`, vae.py:76-84); 'mixgaussian' with 2..4 components and independent label
  dimensions; 'mixtril' (the docstring example of vae.py:58): ONE mixture of 2..4 full-covariance Gaussians over the whole label
  vector (at most 64 label dimensions)."""
  _kind = "sisua"
  _DISCRETE = frozenset(("nb", "nbd", "zinb", "zinbd", "onehot", "categorical", "bernoulli", "poisson"))

  @classmethod
  def _as_mixture(cls, rv, n_components, zero_inflated):
    """One label variable as MISA trains it (vae.py:76-95): a discrete posterior becomes 'mixnb' (zero-inflated on request), any other
    non-mixture posterior 'mixgaussian' -- each conversion with the reference's warning (its wording, typo included: callers match on it) --
    and every variable learns how many components it has."""
    kind = rv.posterior
    is_mixture = kind.startswith(("mix", "mdn"))
    counts_like = kind in cls._DISCRETE or kind.startswith(("mixnb", "mixzinb"))
    if not is_mixture:
      warnings.warn(f"MISA only support labels is a mixture distribution , given: {kind}")
      rv.posterior = "mixnb" if counts_like else "mixgaussian"
    if counts_like:
      rv.kwargs.setdefault("zero_inflated", zero_inflated)
    rv.kwargs.setdefault("n_components", n_components)
    return rv

  def __init__(self, outputs, labels, n_components=2, zero_inflated=False, **kwargs):
    n_components, zero_inflated = int(n_components), bool(zero_inflated)
    labs = [self._as_mixture(l.copy(), n_components, zero_inflated) for l in _flatten(labels)]
    super().__init__(outputs=outputs, labels=labs, **kwargs)
    self.init_args = dict(outputs=outputs, labels=labels, n_components=n_components, zero_inflated=zero_inflated, **kwargs)


class SCALE(SingleCellModel):
  r"""SCALE - "Single-Cell ATAC-seq analysis via Latent feature Extraction" (sisua/models/scale.py:13-49; Xiong et al.
  2019, Nature Communications): a VAE whose prior over z is a TRAINABLE mixture of `n_components` diagonal Gaussians; the
  KL term is the one-sample Monte-Carlo estimate log q(z|x) - log p(z) (`analytic=False`, scale.py:49).  Built:
  covariance='none' / 'diag' (diagonal components) with or without tied mixture weights / locations / scales (scale.py:29-33), and
  covariance='tril' / 'full' (a lower-triangular scale factor per component, at most 32 latent dimensions, untied).
  `mixture='posterior'` selects the LITERAL reading of scale.py:26,38-47 instead: q(z|x) a mixture-density layer of `n_components`
  (2 .. min(latent_dim, 8)) diagonal Gaussians, standard-normal prior, the same Monte-Carlo KL; `predict` / `encode` then report the
  mixture's mean and standard deviation per latent dimension.  [3P-recall: the
  published model; odin's mixture latent layer behind the reference's class is not citable.]"""
  _kind = "scale"

  def __init__(self, outputs, latents=RVmeta(10, "mixgaus", True, name="Latents"), n_components=10, covariance="none",
               tie_mixtures=False, tie_loc=False, tie_scale=False, mixture="prior", **kwargs):
    lat = [z.copy() for z in _flatten(latents)]
    for z in lat:
      if z.posterior[:3] != "mix":
        warnings.warn(f"SCALE only allow mixture distribution for latents  posterior, but given: {z.posterior}")
        z.posterior = "mixgaus"
    kw0 = lat[0].kwargs
    covariance = str(kw0.get("covariance", covariance))
    if covariance not in ("none", "diag", "tril", "full"):
      raise ValueError("SCALE is built for covariance='none' / 'diag' (diagonal components) and 'tril' / 'full' (a lower-triangular factor per component)")
    self._covariance = covariance
    self._ties = dict(tie_mixtures=bool(kw0.get("tie_mixtures", tie_mixtures)), tie_loc=bool(kw0.get("tie_loc", tie_loc)),
                      tie_scale=bool(kw0.get("tie_scale", tie_scale)))
    self._n_components = int(kw0.get("n_components", n_components))
    if not 2 <= self._n_components <= 32:
      raise ValueError(f"SCALE is built for 2..32 mixture components, given: {self._n_components}")
    # mixture='prior' (default): the published model's trainable mixture prior; 'posterior': scale.py:26,38-47 read literally -- q(z|x)
    # itself a mixture-density layer against a standard-normal prior
    mixture = str(kw0.get("mixture", mixture))
    if mixture not in ("prior", "posterior"):
      raise ValueError("mixture must be 'prior' (the published model) or 'posterior' (the literal reading of scale.py)")
    self._latent_mixture = mixture == "posterior"
    if self._latent_mixture:
      if covariance not in ("none", "diag") or any(self._ties.values()):
        raise ValueError("the mixture-density posterior is built with covariance='none' and no tied parameters")
      if not 2 <= self._n_components <= min(lat[0].event_shape, 8):
        raise ValueError("the mixture-density posterior takes 2 .. min(latent_dim, 8) components")
    if covariance in ("tril", "full"):
      if any(self._ties.values()):
        raise ValueError("tied mixture parameters are built for diagonal components only (covariance='none')")
      if lat[0].event_shape > 32:
        raise ValueError("full-covariance mixture components are built for at most 32 latent dimensions")
    super().__init__(outputs=outputs, latents=lat, **kwargs)
    self.init_args = dict(outputs=outputs, latents=latents, n_components=n_components, covariance=covariance,
                          tie_mixtures=tie_mixtures, tie_loc=tie_loc, tie_scale=tie_scale, mixture=mixture, **kwargs)


class SCALAR(SCALE):
  r"""SCALE with semi-supervised extension - "Single-Cell ATAC-seq analysis via Latent and ADT Recombination"
  (sisua/models/scale.py:52-59: `class SCALAR(SCALE, SISUA)`): SCALE's trainable mixture prior over z with SISUA's label
  heads on the decoder output (`labels`: NB / one-hot posteriors weighted by `alpha` and the per-cell label mask)."""
  _kind = "scale"

  def __init__(self, outputs, labels, **kwargs):
    super().__init__(outputs=outputs, labels=labels, **kwargs)
    self.init_args = dict(self.init_args, labels=labels)
    self._n_inputs = 1


class FVAE(SingleCellModel):
  r"""FactorVAE (sisua/models/fvae.py:9-12 -> odin `factorVAE`; Kim & Mnih 2018 "Disentangling by Factorising"): a VAE
  whose objective carries `gamma` x the total correlation of q(z), estimated by a discriminator on z that is trained in
  the same step to tell z from z with every dimension permuted over the minibatch (Algorithm 2).

    discriminator : dict(units=1000, n_hidden_layers=5) -- hidden width / depth (leaky ReLU 0.2, no BatchNorm)
    gamma         : weight of the total-correlation term (6.0)

  [3P-recall: odin's class is not in the reference tree; this follows the paper with odin's defaults.]  Both optimisers
  of the reference (VAE, discriminator; identical Adam settings) are one Adam over all tensors here -- clipnorm and
  Adam are per tensor, and each tensor takes one step per minibatch."""
  _kind = "fvae"

  def __init__(self, outputs, discriminator=None, gamma=6.0, **kwargs):
    disc = dict(units=1000, n_hidden_layers=5)
    disc.update(discriminator or {})
    unknown = set(disc) - {"units", "n_hidden_layers", "alpha"}
    if unknown:
      raise ValueError(f"discriminator options not built: {sorted(unknown)}")
    self._disc_cfg = dict(disc_units=int(disc["units"]), disc_layers=int(disc["n_hidden_layers"]), gamma=float(gamma),
                          disc_leak=float(disc.get("alpha", 0.2)))
    super().__init__(outputs=outputs, **kwargs)
    self.gamma = float(gamma)
    self.init_args = dict(outputs=outputs, discriminator=dict(disc), gamma=gamma, **kwargs)


class SemiFVAE(FVAE):
  r"""Semi-supervised FactorVAE (sisua/models/fvae.py:15-18 -> odin `SemifactorVAE`): the discriminator has one logit
  per class of every one-hot label variable (`labels`: one `RVmeta` or a list; 32 classes in all), its total-correlation
  logit is the logsumexp of all of them, and the labelled cells' masked cross-entropy (x `alpha`) -- summed over the
  variables, each under the softmax of its own logits -- joins both objectives.  [3P-recall: the reading of odin's
  `_tc_logits` / `supervised_loss` is frozen in oracle/sisua_oracle.py; label variables that are not categorical have
  no logits to take and are refused.]"""

  def __init__(self, outputs, labels, **kwargs):
    labs = _flatten(labels)
    if not 1 <= len(labs) <= 8 or any(l.posterior not in ("onehot", "categorical") or l.event_shape < 2 for l in labs) or sum(l.event_shape for l in labs) > 32:
      raise ValueError("SemiFVAE is built for 1..8 'onehot' label variables with at most 32 classes in all")
    super().__init__(outputs=outputs, labels=labels, **kwargs)
    self.init_args = dict(self.init_args, labels=labels)


class SCVI(SingleCellModel):
  r"""Single cell variational inference (sisua/models/scvi.py:20-171).

  Arguments:
    clip_library : `float` (default=`1e3`) clipping of the library latent before exp
  """
  _kind = "scvi"

  def __init__(self,
               outputs,
               latents=RVmeta(10, "diag", True, "Latents"),
               library=RVmeta(1, "normal", True, "Library"),
               encoder=NetConf([64, 64], batchnorm=True, dropout=0.1, name="Encoder"),
               encoder_l=NetConf([64], batchnorm=True, dropout=0.1, name="EncoderL"),
               clip_library=1e3,
               **kwargs):
    outs = _flatten(outputs)
    assert outs[0].posterior in ("zinbd", "nbd"), \
      "scVI only support transcriptomic distribution: 'zinbd' or 'nbd', " + "but given: %s" % str(outs)
    # scvi.py:55-56,66-86: 'full' = a Dense head per plane; otherwise NO head and the distribution layer keeps its own variable --
    # built as 'share' (alias 'gene', scVI's name): one trainable per-gene vector shared by every cell ([3P-recall] odin's
    # NegativeBinomialDispLayer(dispersion='share')); 'single': one trainable scalar for every cell and gene
    opts = {}
    for key in ("dispersion", "inflation"):
      v = str(outs[0].kwargs.get(key, "full")).lower()
      v = "share" if v in ("share", "gene") else v
      if v not in ("full", "share", "single"):
        raise ValueError(f"scVI {key}='{v}' is not built (supported: 'full', 'share' / 'gene', 'single')")
      opts[key] = v
    self._dispersion, self._inflation = opts["dispersion"], opts["inflation"] if outs[0].posterior == "zinbd" else "full"
    self.dispersion, self.inflation = self._dispersion, self._inflation
    super().__init__(outs, latents=[latents, library], encoder=[encoder, encoder_l], clip_library=clip_library, **kwargs)
    self.init_args = dict(outputs=outputs, latents=latents, library=library, encoder=encoder, encoder_l=encoder_l,
                          clip_library=clip_library, **kwargs)


class DeepCountAutoencoder(SingleCellModel):
  r"""Deep Count Autoencoder (sisua/models/dca.py:13-28): deterministic latent."""
  _kind = "dca"

  def __init__(self, outputs, latents=RVmeta(10, "relu", True, name="Latents"), **kwargs):
    lat = [z.copy() for z in _flatten(latents)]
    for z in lat:
      if not z.is_deterministic:
        warnings.warn("DeepCountAutoencoder only support deterministic latents, "
                      f"but given {z}, use default linear Dense layer for latents.")
        z.posterior = "linear"
    self._dca_activation = "relu" if lat[0].posterior == "relu" else "linear"
    super().__init__(outputs=outputs, latents=lat, **kwargs)
    self.init_args = dict(outputs=outputs, latents=latents, **kwargs)

  def _latent_activation(self):
    return self._dca_activation


# ---- registry (sisua/models/__init__.py:11-38) -------------------------------------------------------
def _registry() -> Dict[str, type]:
  """{lower-case class name: class} of every model family of this module, base class included (found through the class tree: a
  subclass a user defines elsewhere is a model too)."""
  found, todo = {}, [SingleCellModel]
  while todo:
    cls = todo.pop()
    found.setdefault(cls.__name__.lower(), cls)
    todo.extend(cls.__subclasses__())
  return found


def get_all_models() -> list:
  """Every model class, ordered by its short id (`vae`, `scvi`, `sisua`, ...; models/__init__.py:11-16)."""
  return sorted(_registry().values(), key=lambda cls: cls.id)


def get_model(model):
  """A model class by class object, class name or short id, case-insensitive (models/__init__.py:19-27)."""
  wanted = (model.__name__ if isinstance(model, type) else str(model)).lower()
  classes = _registry()
  hit = classes.get(wanted) or next((cls for cls in classes.values() if cls.id == wanted), None)
  if hit is None:
    raise RuntimeError(f"Cannot find SingleCellModel with type '{wanted}'")
  return hit


def load_model(filepath: str) -> SingleCellModel:
  r"""sisua/models/__init__.py:30-38.  The sidecar is a pickle (as in the reference): load trusted files only."""
  class_name, dataset, metadata, kwargs = read_metamodel(f"{filepath}.metamodel")
  model = get_model(class_name)(**kwargs)
  model.load_weights(filepath, raise_notfound=True)
  return model


# ---- the `.metamodel` sidecar as plain records ---------------------------------------------------------
_RECORDS = {"RVmeta": RVmeta, "NetConf": NetConf}


def _to_plain(v):
  """Config records -> tagged dicts, containers recursively; arrays and scalars stay."""
  if isinstance(v, (RVmeta, NetConf)):
    d = {k: _to_plain(getattr(v, k)) for k in v.__dataclass_fields__}
    d["__record__"] = type(v).__name__
    return d
  if isinstance(v, dict):
    return {k: _to_plain(x) for k, x in v.items()}
  if isinstance(v, (list, tuple)):
    return [_to_plain(x) for x in v]
  return v


def _from_plain(v):
  if isinstance(v, dict) and "__record__" in v:
    cls = _RECORDS[v["__record__"]]
    return cls(**{k: _from_plain(x) for k, x in v.items() if k != "__record__" and k in cls.__dataclass_fields__})
  if isinstance(v, dict):
    return {k: _from_plain(x) for k, x in v.items()}
  if isinstance(v, list):
    return [_from_plain(x) for x in v]
  return v


class _Shim:
  """Stand-in for a class this process cannot import (odin's RVmeta / NetConf inside a sidecar the REFERENCE
  wrote): pickle restores its attribute dict, `_shim_to_record` turns it into this package's record."""

  def __init__(self, *a, **kw):
    self.__dict__.update(kw)

  def __setstate__(self, state):
    self.__dict__.update(state if isinstance(state, dict) else {})


class _TolerantUnpickler(pickle.Unpickler):

  def find_class(self, module, name):
    try:
      return super().find_class(module, name)
    except Exception:
      return type(name, (_Shim,), {"_shim_name": name})


def _shim_to_record(v):
  if isinstance(v, _Shim):
    nm, st = getattr(v, "_shim_name", ""), dict(v.__dict__)
    if nm in ("RVmeta", "RVconf", "RandomVariable"):
      ev = st.get("event_shape", st.get("dim", 10))
      return RVmeta(event_shape=ev, posterior=st.get("posterior", "diag"), projection=bool(st.get("projection", True)),
                    name=st.get("name"), kwargs=dict(st.get("kwargs", {}) or {}))
    if nm in ("NetConf", "NetworkConfig"):
      return NetConf(units=st.get("units", (64, 64)), batchnorm=bool(st.get("batchnorm", True)),
                     dropout=float(st.get("dropout", 0.0) or 0.0), input_dropout=float(st.get("input_dropout", 0.0) or 0.0),
                     name=st.get("name"))
    return st
  if isinstance(v, dict):
    return {k: _shim_to_record(x) for k, x in v.items()}
  if isinstance(v, (list, tuple)):
    return [_shim_to_record(x) for x in v]
  return v


def read_metamodel(path: str):
  """[class_name, dataset, metadata, init_kwargs] of a `.metamodel` sidecar: this package's plain-record form, or
  one written by the reference itself (its odin RVmeta / NetConf instances are mapped onto the records here)."""
  with open(path, "rb") as f:
    class_name, dataset, metadata, kwargs = _TolerantUnpickler(f).load()
  return class_name, dataset, _shim_to_record(_from_plain(metadata)), _shim_to_record(_from_plain(kwargs))
