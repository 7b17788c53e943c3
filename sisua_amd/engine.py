"""Engine: the Python owner of one `smx_model` (one GPU, one stream).

Thin, mechanical wrapper over the C-ABI (include/sisua_hip.h): it converts numpy
arrays to pointers, keeps the tensor manifest, and raises `SmxError` on any
non-zero status.  All arithmetic happens in libsisua_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np

from sisua_amd import _hip
from sisua_amd._hip import SmxError, check, smx_config, smx_metrics
from sisua_amd.config import ModelConfig, init_params, label_planes, manifest

STREAM_INPUT_DROPOUT = 0
STREAM_ENC_DROPOUT = 16
STREAM_ENCL_DROPOUT = 32
STREAM_DEC_DROPOUT = 48
STREAM_EPS_Z = 64
STREAM_EPS_L = 65


def _fp(a: Optional[np.ndarray]):
  return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a, shape=None):
  a = np.ascontiguousarray(a, dtype=np.float32)
  if shape is not None and tuple(a.shape) != tuple(shape):
    raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
  return a


def make_smx_config(cfg: ModelConfig, max_batch: int) -> smx_config:
  c = smx_config()
  c.abi_version = _hip.SMX_ABI_VERSION
  c.model = _hip.MODEL_KINDS["scale_post" if cfg.latent_mixture else "scale_tril" if cfg.scale_tril else cfg.model]   # (SCALE's covariance='tril' changes a tensor's shape: a model kind of its own, no new field)
  c.likelihood = _hip.LIKELIHOODS[cfg.likelihood]
  c.n_genes, c.latent_dim = int(cfg.n_genes), int(cfg.latent_dim)
  for name, units in (("enc", cfg.enc_units), ("dec", cfg.dec_units), ("encl", cfg.encl_units if cfg.model == "scvi" else ())):
    if len(units) > _hip.SMX_MAX_LAYERS:
      raise ValueError(f"at most {_hip.SMX_MAX_LAYERS} layers per network")
    setattr(c, f"n_{name}", len(units))
    arr = getattr(c, f"{name}_units")
    for i, u in enumerate(units):
      arr[i] = int(u)
  if len(cfg.targets) > _hip.SMX_MAX_LABELS:
    raise ValueError(f"at most {_hip.SMX_MAX_LABELS} heads (extra outputs + label variables)")
  c.n_labels = len(cfg.targets)
  for j, (P, llk) in enumerate(cfg.targets):
    c.label_dim[j] = int(P)
    c.label_llk[j] = _hip.LABEL_LIKELIHOODS[llk[:-1] if llk.startswith("mix") else llk]
    c.label_components[j] = int(llk[-1]) if llk.startswith("mix") else 1
    c.label_observed[j] = 1 if j < len(cfg.extra_outputs) else 0
  c.scvi_dispersion = _hip.SCVI_PLANE_OPTIONS[cfg.dispersion]
  c.scvi_inflation = _hip.SCVI_PLANE_OPTIONS[cfg.inflation]
  c.batchnorm, c.log_norm = int(cfg.batchnorm), int(cfg.log_norm)
  c.latent_activation = _hip.ACTIVATIONS[cfg.latent_activation]
  c.dropout_enc, c.dropout_dec, c.input_dropout = cfg.dropout_enc, cfg.dropout_dec, cfg.input_dropout
  c.beta, c.alpha, c.clip_library = cfg.beta, cfg.alpha, cfg.clip_library
  c.bn_momentum, c.bn_eps = cfg.bn_momentum, cfg.bn_eps
  c.lr, c.adam_beta1, c.adam_beta2, c.adam_eps, c.clipnorm = cfg.lr, cfg.adam_beta1, cfg.adam_beta2, cfg.adam_eps, cfg.clipnorm
  c.n_components = int(cfg.n_components)
  c.disc_units, c.disc_layers = int(cfg.disc_units), int(cfg.disc_layers)
  c.gamma, c.disc_leak = float(cfg.gamma), float(cfg.disc_leak)
  c.max_batch = int(max_batch)
  c.seed = int(cfg.seed) & 0xFFFFFFFFFFFFFFFF
  return c


class Engine:

  def __init__(self, cfg: ModelConfig, max_batch: int = 256, device: int = 0, init: bool = True):
    self.lib = _hip.require_gpu(device)
    self.cfg = cfg
    self.max_batch = int(max_batch)
    self._h = C.c_void_p()
    self.last_comm_error = 0   # the hand-written exchange's error word as the last failed step left it (_step_check)
    c = make_smx_config(cfg, max_batch)
    check(self.lib.smx_model_create(C.byref(c), C.byref(self._h)))
    # manifest handshake: the library's tensor list must equal the host's
    self.names, self.shapes = [], {}
    buf = C.create_string_buffer(64)
    rows, cols = C.c_int32(), C.c_int32()
    for i in range(self.lib.smx_num_tensors(self._h)):
      check(self.lib.smx_tensor_info(self._h, i, buf, 64, C.byref(rows), C.byref(cols)))
      name = buf.value.decode()
      self.names.append(name)
      self.shapes[name] = (rows.value, cols.value) if (name.endswith("/W") or rows.value > 1) else (cols.value,)
    expect = manifest(cfg)
    if [(n, tuple(self.shapes[n])) for n in self.names] != [(n, tuple(s)) for n, s in expect]:
      raise SmxError("tensor manifest of libsisua_hip.so differs from sisua_amd.config.manifest")
    self.index = {n: i for i, n in enumerate(self.names)}
    self.n_cells = 0
    if cfg.model == "scale":   # scale.py:29-33: tied mixture parameters (model semantics carried as flags: no layout change of smx_config)
      for name in ("tie_mixtures", "tie_loc", "tie_scale"):
        if getattr(cfg, name, False):
          check(self.lib.smx_set_flag(self._h, name.encode(), 1))
    if init:
      self.set_params(init_params(cfg))

  # ---- lifetime ---------------------------------------------------------------
  def close(self):
    if getattr(self, "_h", None) is not None and self._h:
      self.lib.smx_model_destroy(self._h)
      self._h = C.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  # ---- parameters / state -------------------------------------------------------
  def get_params(self, which: int = 0) -> Dict[str, np.ndarray]:
    """which: 0 params, 1 gradients of the last step, 2 Adam m, 3 Adam v."""
    out = {}
    for i, n in enumerate(self.names):
      a = np.empty(self.shapes[n], dtype=np.float32)
      check(self.lib.smx_get_tensor(self._h, which, i, _fp(a)))
      out[n] = a
    return out

  def set_params(self, params: Dict[str, np.ndarray], which: int = 0):
    for n, v in params.items():
      a = _f32(v, self.shapes[n])
      check(self.lib.smx_set_tensor(self._h, which, self.index[n], _fp(a)))

  def get_bn(self) -> Dict[int, Dict[str, np.ndarray]]:
    out = {}
    w = C.c_int32()
    for i in range(self.lib.smx_num_bn_layers(self._h)):
      check(self.lib.smx_get_bn(self._h, i, 0, None, C.byref(w)))
      mean, var = np.empty(w.value, np.float32), np.empty(w.value, np.float32)
      check(self.lib.smx_get_bn(self._h, i, 0, _fp(mean), None))
      check(self.lib.smx_get_bn(self._h, i, 1, _fp(var), None))
      out[i] = dict(moving_mean=mean, moving_var=var)
    return out

  def set_bn(self, state: Dict[int, Dict[str, np.ndarray]]):
    for i, st in state.items():
      check(self.lib.smx_set_bn(self._h, int(i), 0, _fp(_f32(st["moving_mean"]))))
      check(self.lib.smx_set_bn(self._h, int(i), 1, _fp(_f32(st["moving_var"]))))

  def snapshot(self) -> dict:
    """Everything a training step changes -- parameters, both Adam moments, the BatchNorm moving statistics, the step counter -- as host
    arrays (what models.py keeps in a checkpoint; parallel.calibrate_forms restores it after its trial steps)."""
    return dict(params=self.get_params(0), m=self.get_params(2), v=self.get_params(3), bn=self.get_bn(), step=self.step)

  def restore(self, st: dict):
    self.set_params(st["params"], 0)
    self.set_params(st["m"], 2)
    self.set_params(st["v"], 3)
    self.set_bn(st["bn"])
    self.step = int(st["step"])

  @property
  def step(self) -> int:
    s = C.c_int32()
    check(self.lib.smx_get_step(self._h, C.byref(s)))
    return s.value

  @step.setter
  def step(self, value: int):
    check(self.lib.smx_set_step(self._h, int(value)))

  # ---- data ----------------------------------------------------------------------
  def upload(self, X, labels: Sequence[np.ndarray] = (), library=None, label_mask=None, cell_id_base: int = 0,
             storage: str = "f32"):
    """Make the cells x genes matrix resident in HBM.  storage='u16' keeps the counts as uint16 (half the bytes;
    integer counts <= 65535 only), storage='f32' is the reference's dense float32 layout, storage='csr' keeps the
    non-zeros only (8 bytes each; X may also be a (indptr, indices, data) triple or a scipy.sparse CSR matrix)."""
    if storage not in ("f32", "u16", "csr"):
      raise ValueError("storage must be 'f32', 'u16' or 'csr'")
    if storage == "csr":
      return self._upload_csr(X, labels, library, label_mask, cell_id_base)
    if storage == "u16":
      Xf = np.asarray(X)
      if Xf.size and (Xf.min() < 0 or Xf.max() > 65535 or not np.array_equal(Xf, np.floor(Xf))):
        raise ValueError("storage='u16' needs integer counts in [0, 65535]")
      X = np.ascontiguousarray(Xf, dtype=np.uint16)
    else:
      X = _f32(X)
    if X.ndim != 2 or X.shape[1] != self.cfg.n_genes:
      raise ValueError(f"X must be [n_cells, {self.cfg.n_genes}]")
    n = X.shape[0]
    labs = [_f32(y, (n, P)) for y, (P, _) in zip(labels, self.cfg.targets)]
    if len(labs) != len(self.cfg.targets):
      raise ValueError("one label matrix per label head is required")
    lab_ptrs = (C.POINTER(C.c_float) * max(1, len(labs)))(*[_fp(y) for y in labs]) if labs else None
    lib_arr = None if library is None else _f32(library, (n, 2))
    mask_arr = None if label_mask is None else np.ascontiguousarray(label_mask, dtype=np.uint8).reshape(n)
    mask_ptr = None if mask_arr is None else mask_arr.ctypes.data_as(C.POINTER(C.c_uint8))
    if storage == "u16":
      check(self.lib.smx_dataset_upload_u16(self._h, X.ctypes.data_as(C.POINTER(C.c_uint16)), n, lab_ptrs, _fp(lib_arr),
                                            mask_ptr, int(cell_id_base)))
    else:
      check(self.lib.smx_dataset_upload(self._h, _fp(X), n, lab_ptrs, _fp(lib_arr), mask_ptr, int(cell_id_base)))
    self.n_cells = n

  def _upload_csr(self, X, labels, library, label_mask, cell_id_base):
    G = self.cfg.n_genes
    if isinstance(X, tuple) and len(X) == 3:
      indptr, indices, data = X
    elif hasattr(X, "tocsr"):   # scipy.sparse
      if X.shape[1] != G:
        raise ValueError(f"X must be [n_cells, {G}]")
      X = X.tocsr()
      X.sort_indices()
      indptr, indices, data = X.indptr, X.indices, X.data
    else:
      Xd = _f32(X)
      if Xd.ndim != 2 or Xd.shape[1] != G:
        raise ValueError(f"X must be [n_cells, {G}]")
      nz = Xd != 0
      indptr = np.concatenate([[0], np.cumsum(nz.sum(1))])
      indices = np.nonzero(nz)[1]
      data = Xd[nz]
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    data = np.ascontiguousarray(data, dtype=np.float32)
    n = indptr.size - 1
    if n < 1 or indptr[0] != 0 or indptr[-1] != indices.size or indices.size != data.size:
      raise ValueError("inconsistent CSR arrays")
    labs = [_f32(y, (n, P)) for y, (P, _) in zip(labels, self.cfg.targets)]
    if len(labs) != len(self.cfg.targets):
      raise ValueError("one label matrix per label head is required")
    lab_ptrs = (C.POINTER(C.c_float) * max(1, len(labs)))(*[_fp(y) for y in labs]) if labs else None
    lib_arr = None if library is None else _f32(library, (n, 2))
    mask_arr = None if label_mask is None else np.ascontiguousarray(label_mask, dtype=np.uint8).reshape(n)
    mask_ptr = None if mask_arr is None else mask_arr.ctypes.data_as(C.POINTER(C.c_uint8))
    check(self.lib.smx_dataset_upload_csr(self._h, indptr.ctypes.data_as(C.POINTER(C.c_int64)), indices.ctypes.data_as(C.POINTER(C.c_int32)),
                                          _fp(data), n, lab_ptrs, _fp(lib_arr), mask_ptr, int(cell_id_base)))
    self.n_cells = n

  # ---- steps -----------------------------------------------------------------------
  @staticmethod
  def _ids(row_ids):
    return np.ascontiguousarray(row_ids, dtype=np.int32).reshape(-1)

  def train_step(self, row_ids, graph: bool = False, metrics: bool = True):
    ids = self._ids(row_ids)
    m = smx_metrics()
    fn = self.lib.smx_train_step_graph if graph else self.lib.smx_train_step
    self._step_check(fn(self._h, ids.ctypes.data_as(C.POINTER(C.c_int32)), ids.size, C.byref(m) if metrics else None))
    return m.as_dict() if metrics else None

  def train_steps(self, order, n_steps: int, batch: int, graph: bool = False, metrics: bool = False):
    """order = None: the row ids made resident by stage_steps(order, n_steps, batch) (no host-to-device copy in this call)."""
    if order is None:
      ptr = None
    else:
      ids = self._ids(order)
      if ids.size != n_steps * batch:
        raise ValueError("order must hold n_steps * batch row ids")
      ptr = ids.ctypes.data_as(C.POINTER(C.c_int32))
    m = smx_metrics()
    self._step_check(self.lib.smx_train_steps(self._h, ptr, int(n_steps), int(batch), int(graph), C.byref(m) if metrics else None))
    return m.as_dict() if metrics else None

  def _step_check(self, rc: int):
    """check() of a step's status; a failed collective of the hand-written exchange (SMX_ERR_COMM: a wait on a peer timed out, every rank's
    step is void) is reported ONCE -- the sticky error word is read and cleared here, so a caller that has dealt with the cause (ADVICE r04:
    it used to fail every later call until someone called comm_p2p_error) can go on: restore a checkpoint, re-attach the ranks."""
    try:
      check(rc)
    except SmxError as err:
      if getattr(err, "code", 0) == -4:
        try:
          self.last_comm_error = self.comm_p2p_error()   # (1: this rank's wait timed out, 2: a peer reported that its wait did)
        except SmxError:
          pass
      raise

  def stage_steps(self, order, n_steps: int, batch: int):
    """The next train_steps call's row ids, uploaded ahead of it (an input pipeline's prefetch)."""
    ids = self._ids(order)
    if ids.size != n_steps * batch:
      raise ValueError("order must hold n_steps * batch row ids")
    check(self.lib.smx_train_stage(self._h, ids.ctypes.data_as(C.POINTER(C.c_int32)), int(n_steps), int(batch)))

  def metrics_history(self, n_steps: int) -> Dict[str, np.ndarray]:
    """Per-step ELBO scalars of the last train_steps call: dict of arrays [n_steps]."""
    h = np.empty((int(n_steps), 8), np.float32)
    check(self.lib.smx_metrics_history(self._h, int(n_steps), _fp(h)))
    return {k: h[:, i].copy() for i, k in enumerate(("loss", "nllk_x", "nllk_y", "kl", "kl_l", "tc", "dtc_loss", "nllk_o"))}

  def eval_step(self, row_ids):
    ids = self._ids(row_ids)
    m = smx_metrics()
    self._step_check(self.lib.smx_eval_step(self._h, ids.ctypes.data_as(C.POINTER(C.c_int32)), ids.size, C.byref(m)))
    return m.as_dict()

  def forward(self, row_ids=None, x=None, library=None, sample_index: int = 0, training: bool = False,
              want_x_params: bool = True):
    """Eval-mode forward; returns dict of logical-shape arrays."""
    cfg = self.cfg
    if row_ids is not None:
      ids = self._ids(row_ids)
      B, idp, xp, lp = ids.size, ids.ctypes.data_as(C.POINTER(C.c_int32)), None, None
    else:
      xa = _f32(x)
      B, idp, xp = xa.shape[0], None, _fp(xa)
      la = None if library is None else _f32(library, (B, 2))
      lp = _fp(la)
    D, G, k = cfg.latent_dim, cfg.n_genes, cfg.k
    out = dict(z_mean=np.empty((B, D), np.float32), z_sample=np.empty((B, D), np.float32))
    out["z_scale"] = np.empty((B, D), np.float32) if cfg.stochastic else None
    if cfg.model == "scvi":
      out.update(l_mean=np.empty((B,), np.float32), l_scale=np.empty((B,), np.float32), l_sample=np.empty((B,), np.float32))
    if want_x_params:
      out["x_params"] = np.empty((k, B, G), np.float32)
    ys = [np.empty((B, label_planes(llk, P) * P), np.float32) for P, llk in cfg.head_labels]
    yptrs = (C.POINTER(C.c_float) * max(1, len(ys)))(*[_fp(y) for y in ys]) if ys else None
    check(self.lib.smx_forward(self._h, idp, xp, lp, B, int(sample_index), int(training), _fp(out["z_mean"]),
                               _fp(out.get("z_scale")), _fp(out["z_sample"]), _fp(out.get("l_mean")),
                               _fp(out.get("l_scale")), _fp(out.get("l_sample")), _fp(out.get("x_params")), yptrs))
    out["y_params"] = ys
    return out

  def forward_samples(self, n_samples: int, row_ids=None, x=None, library=None):
    """n_samples Monte-Carlo draws of one batch in one call (the encoders run once): arrays with a leading draw axis
    for z_sample / l_sample / x_params / y_params, one copy of the posterior means and scales."""
    cfg = self.cfg
    S = int(n_samples)
    if row_ids is not None:
      ids = self._ids(row_ids)
      B, idp, xp, lp = ids.size, ids.ctypes.data_as(C.POINTER(C.c_int32)), None, None
    else:
      xa = _f32(x)
      B, idp, xp = xa.shape[0], None, _fp(xa)
      la = None if library is None else _f32(library, (B, 2))
      lp = _fp(la)
    D, G, k = cfg.latent_dim, cfg.n_genes, cfg.k
    out = dict(z_mean=np.empty((B, D), np.float32), z_sample=np.empty((S, B, D), np.float32))
    out["z_scale"] = np.empty((B, D), np.float32) if cfg.stochastic else None
    if cfg.model == "scvi":
      out.update(l_mean=np.empty((B,), np.float32), l_scale=np.empty((B,), np.float32), l_sample=np.empty((S, B), np.float32))
    out["x_params"] = np.empty((S, k, B, G), np.float32)
    ys = [np.empty((S, B, label_planes(llk, P) * P), np.float32) for P, llk in cfg.head_labels]
    yptrs = (C.POINTER(C.c_float) * max(1, len(ys)))(*[_fp(y) for y in ys]) if ys else None
    check(self.lib.smx_forward_samples(self._h, idp, xp, lp, B, S, _fp(out["z_mean"]), _fp(out.get("z_scale")),
                                       _fp(out["z_sample"]), _fp(out.get("l_mean")), _fp(out.get("l_scale")),
                                       _fp(out.get("l_sample")), _fp(out["x_params"]), yptrs))
    out["y_params"] = ys
    return out

  def predict(self, x, library=None, n_samples: int = 1, batch: Optional[int] = None, want_x_params: bool = True):
    """Eval-mode forward of a whole host matrix in one call (smx_predict): arrays over ALL cells, with a leading draw
    axis for z_sample / l_sample / x_params / y_params -- the layout of forward_samples with n = every cell.
    want_x_params=False: everything but the gene output's parameter planes (what a lazy result keeps on the device side)."""
    cfg = self.cfg
    xa = _f32(x)
    N, S = xa.shape[0], int(n_samples)
    B = min(int(batch or self.max_batch), self.max_batch)
    la = None if library is None else _f32(library, (N, 2))
    D, G, k = cfg.latent_dim, cfg.n_genes, cfg.k
    out = dict(z_mean=np.empty((N, D), np.float32), z_sample=np.empty((S, N, D), np.float32))
    out["z_scale"] = np.empty((N, D), np.float32) if cfg.stochastic else None
    if cfg.model == "scvi":
      out.update(l_mean=np.empty((N,), np.float32), l_scale=np.empty((N,), np.float32), l_sample=np.empty((S, N), np.float32))
    out["x_params"] = np.empty((S, k, N, G), np.float32) if want_x_params else None
    ys = [np.empty((S, N, label_planes(llk, P) * P), np.float32) for P, llk in cfg.head_labels]
    yptrs = (C.POINTER(C.c_float) * max(1, len(ys)))(*[_fp(y) for y in ys]) if ys else None
    check(self.lib.smx_predict(self._h, _fp(xa), _fp(la), N, B, S, _fp(out["z_mean"]), _fp(out.get("z_scale")),
                               _fp(out["z_sample"]), _fp(out.get("l_mean")), _fp(out.get("l_scale")), _fp(out.get("l_sample")),
                               _fp(out["x_params"]), yptrs))
    out["y_params"] = ys
    return out

  STATS = {"mean": 0, "variance": 1, "mean_over_samples": 2, "log_prob": 3}

  def predict_stat(self, x, stat: str, library=None, n_samples: int = 1, batch: Optional[int] = None, count_only: bool = False,
                   target=None, out=None):
    """A statistic of the gene output over a whole host matrix (smx_predict_stat): the same passes and draws as predict(), but only the
    statistic leaves the device.  'mean' / 'variance' [n_samples, N, G]; 'mean_over_samples' [N, G]; 'log_prob' [n_samples, N] of `target`
    (default: of x itself).  count_only: the count distribution without the zero-inflation wrapper.  `out`: a float32 array of the
    result's shape to write into (a reused array saves the first-touch page faults of a fresh one)."""
    cfg = self.cfg
    xa = _f32(x)
    N, S, G = xa.shape[0], int(n_samples), cfg.n_genes
    B = min(int(batch or self.max_batch), self.max_batch)
    la = None if library is None else _f32(library, (N, 2))
    code = self.STATS[stat]
    shape = (N, G) if code == 2 else (S, N) if code == 3 else (S, N, G)
    if out is None:
      out = np.empty(shape, np.float32)
    elif out.dtype != np.float32 or tuple(out.shape) != shape or not out.flags.c_contiguous:
      raise ValueError(f"out must be a C-contiguous float32 array of shape {shape}")
    ta = None if target is None else _f32(target, (N, G))
    check(self.lib.smx_predict_stat(self._h, _fp(xa), _fp(la), N, B, S, code, int(bool(count_only)), _fp(ta), _fp(out)))
    return out

  def decode(self, z, l=None):
    """Decoder + output heads from given latents (eval mode)."""
    cfg = self.cfg
    za = _f32(z)
    B = za.shape[0]
    if za.shape[1] != cfg.latent_dim:
      raise ValueError(f"z must be [batch, {cfg.latent_dim}]")
    la = None if l is None else _f32(np.reshape(l, (B,)))
    xp = np.empty((cfg.k, B, cfg.n_genes), np.float32)
    ys = [np.empty((B, label_planes(llk, P) * P), np.float32) for P, llk in cfg.head_labels]
    yptrs = (C.POINTER(C.c_float) * max(1, len(ys)))(*[_fp(y) for y in ys]) if ys else None
    check(self.lib.smx_decode(self._h, _fp(za), _fp(la), B, _fp(xp), yptrs))
    return dict(x_params=xp, y_params=ys)

  def marginal_llk(self, row_ids=None, x=None, library=None, n_samples: int = 100):
    """Importance-weighted log p(x) per cell and the mean reconstruction log-likelihood (GPU)."""
    if row_ids is not None:
      ids = self._ids(row_ids)
      B, idp, xp, lp = ids.size, ids.ctypes.data_as(C.POINTER(C.c_int32)), None, None
    else:
      xa = _f32(x)
      B, idp, xp = xa.shape[0], None, _fp(xa)
      la = None if library is None else _f32(library, (B, 2))
      lp = _fp(la)
    mllk, llk = np.empty(B, np.float32), np.empty(B, np.float32)
    check(self.lib.smx_marginal_llk(self._h, idp, xp, lp, B, int(n_samples), _fp(mllk), _fp(llk)))
    return mllk, llk

  # ---- resident-matrix preprocessing ---------------------------------------
  def generate_lognormal(self, n_cells: int, seed: int = 8, rank: int = 0, storage: str = "u16", density: float = 0.14):
    """The rank's shard of BASELINE configs[4] generated on the device (smx_dataset_generate_lognormal): nothing crosses
    PCIe; rows are those of one virtual matrix keyed by the global cell id rank * n_cells + row."""
    if storage not in ("u16", "f32"):
      raise ValueError("storage must be 'u16' or 'f32'")
    check(self.lib.smx_dataset_generate_lognormal(self._h, int(seed), int(rank), int(n_cells), 1 if storage == "u16" else 0, float(density)))
    self.n_cells = int(n_cells)

  def dataset_library(self):
    """get_library_size (data/utils.py:231-263) over the resident matrix on the GPU; fills the resident
    library prior and returns (local_mean, local_var)."""
    st = np.zeros(2, np.float32)
    check(self.lib.smx_dataset_library(self._h, _fp(st)))
    return float(st[0]), float(st[1])

  def dataset_corrupt(self, dropout_rate: float = 0.2, retain_rate: float = 0.2, seed: int = 8) -> int:
    """'binomial' artificial corruption (data/utils.py:168-228) in place on the resident matrix with the
    counter RNG (oracle: corrupt_philox).  Returns the number of corrupted entries."""
    n = C.c_int64(0)
    check(self.lib.smx_dataset_corrupt(self._h, float(dropout_rate), float(retain_rate), int(seed), C.byref(n)))
    return int(n.value)

  def dataset_read(self, row0: int = 0, n_rows: int = None, library: bool = False):
    """Rows of the resident matrix, their constants sum_g lgamma(x+1), and optionally the library prior."""
    n = int(self.lib.smx_dataset_size(self._h)) - int(row0) if n_rows is None else int(n_rows)
    X = np.empty((n, self.cfg.n_genes), np.float32)
    rc = np.empty(n, np.float32)
    lb = np.empty((n, 2), np.float32) if library else None
    check(self.lib.smx_dataset_read(self._h, int(row0), n, _fp(X), _fp(rc), _fp(lb) if library else None))
    return (X, rc, lb) if library else (X, rc)

  def score_llk(self, targets, row_ids=None, x=None, library=None, n_samples: int = 10):
    """Posterior-predictive log-likelihood per cell (Posterior.cal_llk, posterior.py:919-938), on the GPU.
    `targets`: list of [B, G] matrices (None = the input cells).  Returns [len(targets), 2, B]:
    [:, 0] under the output distribution, [:, 1] under its count distribution without zero inflation."""
    if row_ids is not None:
      ids = self._ids(row_ids)
      B, idp, xp, lp = ids.size, ids.ctypes.data_as(C.POINTER(C.c_int32)), None, None
    else:
      xa = _f32(x)
      B, idp, xp = xa.shape[0], None, _fp(xa)
      la = None if library is None else _f32(library, (B, 2))
      lp = _fp(la)
    keep = [None if t is None else _f32(t, (B, self.cfg.n_genes)) for t in targets]
    arr = (C.POINTER(C.c_float) * len(keep))(*[_fp(t) if t is not None else C.POINTER(C.c_float)() for t in keep])
    out = np.empty((len(keep), 2, B), np.float32)
    check(self.lib.smx_score_llk(self._h, idp, xp, lp, arr, len(keep), B, int(n_samples), _fp(out)))
    return out

  # ---- noise injection (parity tests) ----------------------------------------------
  def set_noise(self, stream: int, data):
    a = _f32(data)
    if a.ndim == 1:
      a = a.reshape(-1, 1)
    check(self.lib.smx_set_noise(self._h, int(stream), _fp(a), a.shape[0], a.shape[1]))

  def clear_noise(self):
    check(self.lib.smx_clear_noise(self._h))

  # ---- data parallel ------------------------------------------------------------------
  @staticmethod
  def comm_unique_id() -> bytes:
    lib = _hip.load()
    buf = (C.c_uint8 * 128)()
    check(lib.smx_comm_unique_id(buf))
    return bytes(buf)

  def comm_init(self, rank: int, world: int, uid: bytes):
    buf = (C.c_uint8 * 128).from_buffer_copy(uid)
    check(self.lib.smx_comm_init(self._h, int(rank), int(world), buf))

  def comm_p2p_export(self, world: int) -> bytes:
    """This rank's two IPC handles (flat gradient buffer | communication region) for the hand-written two-shot all-reduce
    over peer-mapped buffers (smx_comm_p2p_export); gather every rank's 128 bytes in rank order, then comm_p2p_init."""
    buf = (C.c_uint8 * 128)()
    check(self.lib.smx_comm_p2p_export(self._h, int(world), buf))
    return bytes(buf)

  def comm_p2p_init(self, rank: int, world: int, all_handles: bytes):
    if len(all_handles) != 128 * world:
      raise ValueError("all_handles must hold world x 128 bytes in rank order")
    buf = (C.c_uint8 * len(all_handles)).from_buffer_copy(all_handles)
    check(self.lib.smx_comm_p2p_init(self._h, int(rank), int(world), buf))

  def comm_p2p_error(self) -> int:
    """Non-zero after a wait on a peer timed out in the peer-to-peer exchange (cleared by the call)."""
    e = C.c_int32(0)
    check(self.lib.smx_comm_p2p_error(self._h, C.byref(e)))
    return int(e.value)

  def comm_time_allreduce(self, iters: int = 50):
    """(microseconds per all-reduce of the flat gradient buffer ALONE, its bytes), through the path the steps take (every rank calls it)."""
    us, n = C.c_float(0.0), C.c_int64(0)
    check(self.lib.smx_comm_time_allreduce(self._h, int(iters), C.byref(us), C.byref(n)))
    return float(us.value), 4 * int(n.value)

  @property
  def world(self) -> int:
    return self.lib.smx_comm_world(self._h)

  @property
  def rank(self) -> int:
    return self.lib.smx_comm_rank(self._h)

  @property
  def comm_form(self) -> int:
    """How a training step exchanges its gradients: 0 no collective, 1 one all-reduce, 2 the two-bucket chain (the heads' bucket reduced,
    normed and applied on the communication stream), 3 the hand-written exchange (one launch per all-reduce), 4 its two-bucket form of
    round 4 (include/sisua_hip.h: smx_comm_form)."""
    return self.lib.smx_comm_form(self._h)

  FORM_NAMES = {0: "none", 1: "one all-reduce", 2: "two-bucket chain", 3: "hand-written exchange", 4: "hand-written exchange, two buckets"}

  def comm_set_form(self, form: int):
    """Ask for an exchange form (smx_comm_set_form: 1 / 2 / 3, or 0 = the library's rule).  A collective call for form 2: every rank."""
    check(self.lib.smx_comm_set_form(self._h, int(form)))

  def opt_gather(self):
    """Flag opt_shard: all-gather the heads' Adam moments (a collective: every rank calls it, between training calls); afterwards
    get_params(2 | 3) returns the job's moments on every rank (include/sisua_hip.h: smx_opt_gather)."""
    check(self.lib.smx_opt_gather(self._h))

  def set_sync_bn(self, on: bool = True):
    """SyncBatchNorm under data parallelism (global-batch statistics; one small extra all-reduce per BatchNorm pass)."""
    check(self.lib.smx_comm_set_sync_bn(self._h, int(bool(on))))

  @staticmethod
  def comm_library() -> dict:
    """Paths of the bound RCCL and of the HIP runtime both libraries run on, and the RCCL version code."""
    lib = _hip.load()
    a, b, v = C.create_string_buffer(4096), C.create_string_buffer(4096), C.c_int32()
    check(lib.smx_comm_library(a, 4096, b, 4096, C.byref(v)))
    return dict(rccl=a.value.decode(), hip=b.value.decode(), rccl_version=int(v.value))

  @staticmethod
  def comm_init_local(engines: Sequence["Engine"]):
    """Test hook: loopback communicator over engines of this process (rank i = engines[i]); afterwards each engine
    must be driven by its own host thread, all of them running the same number of equal-size steps."""
    lib = _hip.load()
    arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
    check(lib.smx_comm_init_local(arr, len(engines)))

  def set_flag(self, name: str, value: bool):
    """Code-path switch (smx_set_flag).  Training step: head_loss, head_fused, head_sweep, front, bwd_front, head_bwd, wgrad, scvi_fused, twin, label_ride,
    act_epilogue; data parallel: opt_shard (the heads' optimiser state sharded over the ranks; opt_gather()); scoring: stacked_scoring; "bf16x3" (products of the output head from bf16 MFMAs on three-way split operands):
    True / False, or -1 for the default (by the head's width)."""
    v = -1 if (name == "bf16x3" and not isinstance(value, bool) and int(value) < 0) else int(bool(value))
    check(self.lib.smx_set_flag(self._h, name.encode(), v))

  # ---- measurement ----------------------------------------------------------------------
  def timing_enable(self, kernel: Optional[str]):
    check(self.lib.smx_timing_enable(self._h, kernel.encode() if kernel else None))

  def timing_read(self):
    ms, n = C.c_double(), C.c_int64()
    check(self.lib.smx_timing_read(self._h, C.byref(ms), C.byref(n)))
    return ms.value, n.value

  def loss_bytes_per_cell(self) -> int:
    return int(self.lib.smx_loss_bytes_per_cell(self._h))

  def head_fused_bytes(self, batch: int) -> int:
    """Algorithmic bytes of one launch of the fused output head when a training step of `batch` cells takes it, else 0."""
    return int(self.lib.smx_head_fused_bytes(self._h, int(batch)))

  def synchronize(self):
    check(self.lib.smx_synchronize())


# ---- kernel-level helpers (tests) -----------------------------------------------------
def k_count_llk(likelihood: str, x, planes, direct: bool = False, want_grads: bool = True):
  lib = _hip.require_gpu()
  x = _f32(x)
  B, G = x.shape
  pl = _f32(planes)
  k = pl.shape[0]
  llk = np.empty(B, np.float32)
  grads = np.empty((k, B, G), np.float32) if want_grads else None
  check(lib.smx_k_count_llk(_hip.LIKELIHOODS[likelihood], int(direct), _fp(x), _fp(pl), B, G, _fp(llk), _fp(grads)))
  return llk, grads


def k_head_fused(likelihood: str, x, d, W, bias, grad_scale: float = 1.0, u16: bool = False, reps: int = 0):
  """The fused output head (smx_headfused.hip) over host arrays: x [B][G], d [B][128], W [128][k][G], bias [k][G] ->
  dict(llk [B], dW, db, dd [B][128], sumsq, us)."""
  lib = _hip.require_gpu()
  x, d, W, bias = _f32(x), _f32(d), _f32(W), _f32(bias)
  B, G = x.shape
  k = W.shape[1]
  assert d.shape == (B, 128) and W.shape == (128, k, G) and bias.shape == (k, G)
  llk = np.empty(B, np.float32)
  dW, db, dd = np.empty((128, k, G), np.float32), np.empty((k, G), np.float32), np.empty((B, 128), np.float32)
  sumsq, us = np.zeros(1, np.float32), np.zeros(1, np.float32)
  check(lib.smx_k_head_fused(_hip.LIKELIHOODS[likelihood], int(u16), _fp(x), _fp(d), _fp(W), _fp(bias), B, G, float(grad_scale), int(reps),
                             _fp(llk), _fp(dW), _fp(db), _fp(dd), _fp(sumsq), _fp(us)))
  return dict(llk=llk, dW=dW, db=db, dd=dd, sumsq=float(sumsq[0]), us=float(us[0]))


def k_head_fused_stress(likelihood: str, x, d, W, bias, launches: int, grad_scale: float = 1.0, u16: bool = False):
  """`launches` launches of the fused output head on the same inputs, each compared bit for bit on the device with the first:
  (launches that differed, index of the first differing output word or -1)."""
  lib = _hip.require_gpu()
  x, d, W, bias = _f32(x), _f32(d), _f32(W), _f32(bias)
  B, G = x.shape
  k = W.shape[1]
  assert d.shape == (B, 128) and W.shape == (128, k, G) and bias.shape == (k, G)
  n, first = C.c_int32(0), C.c_int64(-1)
  check(lib.smx_k_head_fused_stress(_hip.LIKELIHOODS[likelihood], int(u16), _fp(x), _fp(d), _fp(W), _fp(bias), B, G, float(grad_scale), int(launches),
                                    C.byref(n), C.byref(first)))
  return int(n.value), int(first.value)


def k_adam(params, grads, m, v, step: int, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7, clipnorm=100.0):
  """The optimiser launch by itself over lists of arrays (updated copies are returned): (params, m, v, norms)."""
  lib = _hip.require_gpu()
  sizes = np.array([int(np.size(p)) for p in params], dtype=np.int32)
  cat = lambda xs: np.ascontiguousarray(np.concatenate([np.asarray(x, np.float32).ravel() for x in xs]))
  P, G, M, V = cat(params), cat(grads), cat(m), cat(v)
  norms = np.empty(len(sizes), np.float32)
  check(lib.smx_k_adam(len(sizes), sizes.ctypes.data_as(C.POINTER(C.c_int32)), _fp(P), _fp(G), _fp(M), _fp(V), int(step),
                       float(lr), float(beta1), float(beta2), float(eps), float(clipnorm), _fp(norms)))
  cut = np.cumsum(sizes)[:-1]
  shp = [np.shape(p) for p in params]
  un = lambda flat: [a.reshape(sh) for a, sh in zip(np.split(flat, cut), shp)]
  return un(P), un(M), un(V), norms


def k_gemm(A, B, trans_a=False, trans_b=False, split_k=1, tile=0):
  lib = _hip.require_gpu()
  A, B = _f32(A), _f32(B)
  M, K = (A.shape[1], A.shape[0]) if trans_a else A.shape
  N = B.shape[0] if trans_b else B.shape[1]
  Cm = np.empty((M, N), np.float32)
  check(lib.smx_k_gemm(int(trans_a), int(trans_b), _fp(A), _fp(B), M, N, K, int(split_k), int(tile), _fp(Cm)))
  return Cm


def k_noise(seed, stream, step, cell_ids, width, p=0.0, sample=0):
  lib = _hip.require_gpu()
  ids = np.ascontiguousarray(cell_ids, dtype=np.int64)
  mult = np.empty((ids.size, width), np.float32)
  nrm = np.empty((ids.size, width), np.float32)
  check(lib.smx_k_noise(int(seed), int(stream), int(step), int(sample), ids.ctypes.data_as(C.POINTER(C.c_int64)),
                        ids.size, int(width), float(p), _fp(mult), _fp(nrm)))
  return mult, nrm
