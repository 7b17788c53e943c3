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
  ix = rand.choice(range(len(i)), size=int(np.floor(dropout_rate * len(i))), replace=False)
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


def epoch_order(n_obs: int, epoch: int, shuffle: int = 1000, seed: int = 1) -> np.ndarray:
  """Visit order of one epoch under a streaming shuffle buffer of size `shuffle`
  (tf.data .shuffle(1000) after .cache, before .batch)."""
  if not shuffle or shuffle <= 0:
    return np.arange(n_obs, dtype=np.int32)
  rng = np.random.RandomState(seed + epoch)
  buf = list(range(min(shuffle, n_obs)))
  nxt = len(buf)
  out = np.empty(n_obs, dtype=np.int32)
  picks = rng.randint(0, 2 ** 31 - 1, size=n_obs)
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
