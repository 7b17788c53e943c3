"""Host-side data semantics of the training path (NumPy; runs once per fit, not
per step): the pieces of the reference's data layer that decide WHICH cells and
WHICH values reach the step.

  split_indices      SingleCellOMIC.split            sisua/data/single_cell_dataset.py:43-81
  corrupt            apply_artificial_corruption     sisua/data/utils.py:168-228 (via _single_cell_analysis.py:78-111)
  library_size       get_library_size                sisua/data/utils.py:231-263
  label_mask / epoch_order / iter_batches
                     _OMICbase.create_dataset        sisua/data/_single_cell_base.py:539-602
  synthetic_*        shape/sparsity-matched stand-ins for the datasets of
                     description/dataset.html (the real ones need S3 downloads)

Bit-exactness of the first three against the reference's own code is pinned by
tests/golden/reference_data_fixtures.npz.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import ctypes as _C

import numpy as np


def split_indices(n_obs: int, train_percent: float = 0.8, seed: int = 1) -> Tuple[np.ndarray, np.ndarray]:
  train_percent = np.clip(train_percent, 0.0, 1.0)
  ids = np.random.RandomState(seed=seed).permutation(n_obs).astype("int32")
  n_train = int(train_percent * n_obs)
  return ids[:n_train], ids[n_train:]


def corrupt(x: np.ndarray, dropout_rate: float = 0.2, retain_rate: float = 0.2, distribution: str = "binomial",
            seed: int = 8, inplace: bool = False) -> np.ndarray:
  distribution = str(distribution).lower()
  dropout_rate = float(dropout_rate)
  if not 0 <= dropout_rate < 1:
    raise ValueError(f"dropout value must be >= 0 and < 1, given: {dropout_rate}")
  out = x if inplace else np.array(x, copy=True)
  if not (0.0 < dropout_rate < 1.0 or 0.0 < retain_rate < 1.0):
    return out
  rand = np.random.RandomState(seed=seed)
  i, j = np.nonzero(x)
  n_sel = int(np.floor(dropout_rate * len(i)))
  if n_sel == 0:   # nothing to corrupt (the reference's rand.choice / fancy indexing fails on this empty case)
    if distribution not in ("binomial", "uniform"):
      raise ValueError("Only support 2 corruption distribution: 'uniform' and 'binomial', "
                       f"but given: '{distribution}'")
    return out
  ix = rand.choice(range(len(i)), size=n_sel, replace=False)
  i, j = i[ix], j[ix]
  if distribution == "binomial":
    vals = rand.binomial(n=(x[i, j]).astype(np.int32), p=retain_rate)
  elif distribution == "uniform":
    vals = np.multiply(x[i, j], rand.binomial(n=np.ones(len(ix), dtype=np.int32), p=retain_rate))
  else:
    raise ValueError("Only support 2 corruption distribution: 'uniform' and 'binomial', "
                     f"but given: '{distribution}'")
  out[i, j] = vals
  return out


def library_size(x: np.ndarray):
  """-> (log_counts [N], local_mean, local_var); the model receives
  library = [[local_mean, local_var]] * N (data/_single_cell_base.py:568-570)."""
  if x.ndim != 2:
    raise ValueError("Only support 2-D matrix")
  total = x.sum(axis=1)
  log_counts = np.log(total + 1e-8)
  return log_counts, np.float32(np.mean(log_counts)), np.float32(np.var(log_counts))


def library_matrix(x: np.ndarray) -> np.ndarray:
  _, m, v = library_size(x)
  return np.tile(np.array([[m, v]], dtype=np.float32), (x.shape[0], 1))


def label_mask(n_obs: int, labels_percent: float, n_omics: int, seed: int = 1) -> np.ndarray:
  """Per-cell 'labelled' flag drawn once (the reference draws it in a tf.data map and
  freezes it with .cache(''); forced False with a single omic)."""
  labels_percent = float(np.clip(labels_percent, 0.0, 1.0))
  if n_omics <= 1 or labels_percent <= 0.0:
    return np.zeros(n_obs, dtype=bool)
  return np.random.RandomState(seed).uniform(size=n_obs) < labels_percent


_SHUFFLE_LIB = [False]


def _shuffle_lib():
  """libsisua_hip.so when it is built (its smx_shuffle_order is host-only code); None otherwise: the Python walk below
  is the definition, the library only runs it faster (tests/test_data.py checks they agree)."""
  if _SHUFFLE_LIB[0] is False:
    try:
      from sisua_amd import _hip
      _SHUFFLE_LIB[0] = _hip.load()
    except Exception:
      _SHUFFLE_LIB[0] = None
  return _SHUFFLE_LIB[0]


def epoch_order(n_obs: int, epoch: int, shuffle: int = 1000, seed: int = 1) -> np.ndarray:
  """Visit order of one epoch under a streaming shuffle buffer of size `shuffle`
  (tf.data .shuffle(1000) after .cache, before .batch)."""
  if not shuffle or shuffle <= 0:
    return np.arange(n_obs, dtype=np.int32)
  rng = np.random.RandomState(seed + epoch)
  out = np.empty(n_obs, dtype=np.int32)
  picks = rng.randint(0, 2 ** 31 - 1, size=n_obs)
  lib = _shuffle_lib()
  if lib is not None:   # the sequential walk in C (smx_shuffle_order): ~10 us instead of 2.3 ms per 3381-cell epoch
    p64 = np.ascontiguousarray(picks, dtype=np.int64)
    rc = lib.smx_shuffle_order(int(n_obs), int(shuffle), p64.ctypes.data_as(_C.POINTER(_C.c_int64)), out.ctypes.data_as(_C.POINTER(_C.c_int32)))
    if rc == 0:
      return out
  buf = list(range(min(shuffle, n_obs)))
  nxt = len(buf)
  for t in range(n_obs):
    k = picks[t] % len(buf)
    out[t] = buf[k]
    if nxt < n_obs:
      buf[k] = nxt
      nxt += 1
    else:
      buf[k] = buf[-1]
      buf.pop()
  return out


def iter_batches(order: np.ndarray, batch_size: int, drop_remainder: bool = True) -> List[np.ndarray]:
  n = len(order)
  end = (n // batch_size) * batch_size if drop_remainder else n
  return [order[s:s + batch_size] for s in range(0, end, batch_size)]


def shard_for_rank(ids: np.ndarray, rank: int, world: int) -> np.ndarray:
  """Data-parallel partition of the cells: rank r owns ids[r::world] truncated to a
  common length, so every rank runs the same number of steps (no data-path collective)."""
  n = (len(ids) // world) * world
  return ids[:n][rank::world]


def shard_range(n_obs: int, rank: int, world: int) -> Tuple[int, int]:
  """Contiguous data-parallel partition [lo, hi) of the cells (rank r owns N / world cells resident in
  its HBM), every rank the same count.  Contiguous so that a cell's GLOBAL id (the Philox key of its dropout masks
  and eps) is `lo + local row`: the noise of a cell does not depend on how the cells are sharded."""
  n = n_obs // world
  return rank * n, (rank + 1) * n


# ---------------------------------------------------------------------------
# synthetic stand-ins; generator seed 8 = the repo's habitual seed
# ---------------------------------------------------------------------------
def _counts(n, g, rng, gene_sd, size_sd, target_sparsity):
  m_g = rng.normal(0.0, gene_sd, size=g)
  s_c = rng.lognormal(0.0, size_sd, size=n)
  lo, hi = -14.0, 6.0
  gam = rng.gamma(2.0, 0.5, size=(n, g)).astype(np.float32)
  base = (s_c[:, None] * np.exp(m_g)[None, :]).astype(np.float32) * gam
  for _ in range(30):  # bisection on a global offset until P(x == 0) hits the target
    mid = 0.5 * (lo + hi)
    p0 = float(np.exp(-base * np.exp(mid)).mean())
    lo, hi = (lo, mid) if p0 < target_sparsity else (mid, hi)
  x = rng.poisson(base * np.exp(0.5 * (lo + hi))).astype(np.float32)
  return x


def synthetic_8kly(seed: int = 8, n: int = 4697, g: int = 1998, n_proteins: int = 12):
  """pbmc8k_ly-shaped: 4697 x 1998, sparsity ~0.93, mean non-zero count ~4.2, 12
  real-valued protein levels in [0.5, 9.1] (description/dataset.html:187)."""
  rng = np.random.default_rng(seed)
  x = _counts(n, g, rng, 2.9, 0.28, 0.93)
  x[x.sum(1) == 0, 0] = 1.0
  y = np.clip(rng.lognormal(0.9, 0.7, size=(n, n_proteins)), 0.5, 9.1).astype(np.float32)
  return x, y


def synthetic_eccly(seed: int = 8):
  """pbmcecc_ly-shaped: 2941 x 2000, sparsity ~0.89, 38 proteins (dataset.html:217)."""
  rng = np.random.default_rng(seed + 1)
  x = _counts(2941, 2000, rng, 2.6, 0.3, 0.89)
  x[x.sum(1) == 0, 0] = 1.0
  y = np.clip(rng.lognormal(0.9, 0.7, size=(2941, 38)), 0.5, 9.1).astype(np.float32)
  return x, y


def synthetic_cortex(seed: int = 8):
  """cortex-shaped: 3005 x 558, sparsity ~0.29, heavy tail, 7 one-hot cell types
  (dataset.html:31)."""
  rng = np.random.default_rng(seed + 2)
  x = _counts(3005, 558, rng, 1.8, 0.5, 0.29)
  x[rng.integers(0, 3005), rng.integers(0, 558)] = 10738.0
  x[x.sum(1) == 0, 0] = 1.0
  y = np.eye(7, dtype=np.float32)[rng.integers(0, 7, 3005)]
  return x, y


def synthetic_scalability(n: int, seed: int = 8):
  """The reference's own scalability shape: randint(0,100,(n,500)) genes and
  randint(0,10,(n,10)) proteins (tests/test_scalability.py:22-27)."""
  rng = np.random.default_rng(seed)
  return rng.integers(0, 100, size=(n, 500)).astype(np.float32), rng.integers(0, 10, size=(n, 10)).astype(np.float32)


# ---------------------------------------------------------------------------
# Minimal multi-omic container + minibatch description: what SingleCellModel.fit /
# predict need from SingleCellOMIC (data/_single_cell_base.py) without AnnData.
# ---------------------------------------------------------------------------
class BatchDataset:
  """What `SingleCellOMIC.create_dataset` returns in the reference (a tf.data pipeline of
  dict(inputs, library, mask) batches, _single_cell_base.py:539-602), as plain arrays plus
  the batching recipe; the arrays are uploaded to HBM once and minibatches are row ids."""

  def __init__(self, arrays: Sequence[np.ndarray], omics: Sequence[str], library: np.ndarray, mask: np.ndarray,
               batch_size: int = 64, drop_remainder: bool = False, shuffle: int = 1000, seed: int = 1):
    self.arrays = [np.ascontiguousarray(a, dtype=np.float32) for a in arrays]
    self.omics = list(omics)
    self.library = np.ascontiguousarray(library, dtype=np.float32)
    self.mask = np.ascontiguousarray(mask, dtype=bool)
    self.batch_size, self.drop_remainder, self.shuffle, self.seed = int(batch_size), bool(drop_remainder), int(shuffle or 0), int(seed)

  @property
  def n_obs(self):
    return self.arrays[0].shape[0]

  def epoch_batches(self, epoch: int) -> List[np.ndarray]:
    return iter_batches(epoch_order(self.n_obs, epoch, self.shuffle, self.seed), self.batch_size, self.drop_remainder)

  def steps_per_epoch(self) -> int:
    n = self.n_obs
    return n // self.batch_size if self.drop_remainder else -(-n // self.batch_size)

  def __iter__(self):
    for ids in self.epoch_batches(0):
      inputs = [a[ids] for a in self.arrays]
      yield dict(inputs=inputs[0] if len(inputs) == 1 else inputs, library=self.library[ids], mask=self.mask[ids])


class SingleCellOMIC:
  """Multi-omic cells x features container with the methods the training path calls on the
  reference's SingleCellOMIC: split, corrupt, library statistics, create_dataset, get_rv."""

  def __init__(self, X, var_names=None, name: str = "scOMICS", omic: str = "transcriptomic"):
    self.name = name
    self._data, self._vars = {}, {}
    self.add_omic(omic, X, var_names)

  def add_omic(self, omic: str, X, var_names=None):
    X = np.ascontiguousarray(X, dtype=np.float32)
    if self._data and X.shape[0] != self.n_obs:
      raise ValueError(f"Number of cell mismatch {self.n_obs} and {X.shape[0]}")
    self._data[str(omic)] = X
    self._vars[str(omic)] = np.array([f"{omic}{i}" for i in range(X.shape[1])]) if var_names is None else np.asarray(var_names)
    return self

  @property
  def omics(self):
    return list(self._data)

  @property
  def n_omics(self):
    return len(self._data)

  @property
  def n_obs(self):
    return next(iter(self._data.values())).shape[0]

  @property
  def n_vars(self):
    """Number of variables of the first OMIC (AnnData's n_vars; tests/test_singlecell_models.py:127)."""
    return next(iter(self._data.values())).shape[1]

  def get_omic(self, omic):
    return self._data[str(omic)]

  def numpy(self, omic=None):
    return self._data[str(omic) if omic is not None else self.omics[0]]

  def get_dim(self, omic):
    return self._data[str(omic)].shape[1]

  def get_var_names(self, omic):
    return self._vars[str(omic)]

  def __getitem__(self, ids):
    out = None
    for om in self.omics:
      if out is None:
        out = SingleCellOMIC(self._data[om][ids], self._vars[om], name=self.name, omic=om)
      else:
        out.add_omic(om, self._data[om][ids], self._vars[om])
    return out

  def copy(self):
    return self[np.arange(self.n_obs)]

  def split(self, train_percent=0.8, copy=True, seed=1):
    tr, te = split_indices(self.n_obs, train_percent, seed)
    return (None if len(tr) == 0 else self[tr]), (None if len(te) == 0 else self[te])

  def corrupt(self, dropout_rate=0.2, retain_rate=0.2, distribution="binomial", omic=None, inplace=True, seed=8):
    om = self if inplace else self.copy()
    key = str(omic) if omic is not None else om.omics[0]
    corrupt(om._data[key], dropout_rate, retain_rate, distribution, seed, inplace=True)
    return om

  def sparsity(self, omic=None):
    return float((self.numpy(omic) == 0).mean())

  def library_size(self, omic=None):
    return library_matrix(self.numpy(omic))

  def get_rv(self, omic, distribution=None):
    from sisua_amd.config import RVmeta
    omic = str(omic)
    if distribution is None:
      if omic in ("transcriptomic", "atac", "chromatin"):
        distribution = "zinb"
      elif omic == "proteomic":
        distribution = "nb"
      elif omic in ("celltype", "disease", "progenitor"):
        distribution = "onehot"
      else:
        raise ValueError(f"No default distribution for OMIC {omic}")
    return RVmeta(event_shape=self.get_dim(omic), posterior=distribution, projection=True, name=omic)

  create_rv = get_rv

  def create_dataset(self, omics=None, labels_percent=0, batch_size=64, drop_remainder=False, shuffle=1000, seed=1):
    if omics is None:
      omics = [self.omics[0]]
    if isinstance(omics, str):
      omics = [omics]
    omics = [str(o) for o in omics]
    arrays = [self.get_omic(o) for o in omics]
    return BatchDataset(arrays, omics, self.library_size(omics[0]), label_mask(self.n_obs, labels_percent, len(omics), seed),
                        batch_size, drop_remainder, shuffle, seed)

  def __repr__(self):
    return f"<SingleCellOMIC '{self.name}' n_obs={self.n_obs} " + " ".join(f"{o}:{self.get_dim(o)}" for o in self.omics) + ">"
