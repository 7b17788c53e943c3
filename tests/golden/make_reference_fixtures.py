#!/usr/bin/env python3
"""Generate tests/golden/reference_data_fixtures.npz by EXECUTING the reference's own
functions (read from /root/reference at generation time only; nothing from the
reference is copied into the repo -- the fixture holds inputs and outputs).

Functions run:
  * apply_artificial_corruption   sisua/data/utils.py:168-228
  * get_library_size              sisua/data/utils.py:231-263
  * SingleCellOMIC.split          sisua/data/single_cell_dataset.py:43-81

The modules they live in import odin/tensorflow/scanpy at module level (not
installable here), so each function's own source is lifted out of its module with
`ast` and executed against numpy/scipy only.  `split` is a method: it is run with
a minimal `self` that supplies what the method touches (`n_obs`, `_record`,
`copy`, `__getitem__` returning the selected ids).

Run in the build container:  python tests/golden/make_reference_fixtures.py
"""
import ast
import os
import warnings
from copy import deepcopy

import numpy as np
from scipy import sparse

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_data_fixtures.npz")


def lift(path, name, namespace):
  src = open(os.path.join(REF, path)).read()
  tree = ast.parse(src)
  for node in ast.walk(tree):
    if isinstance(node, ast.FunctionDef) and node.name == name:
      node.returns = None
      for a in node.args.args:
        a.annotation = None
      mod = ast.Module(body=[node], type_ignores=[])
      exec(compile(mod, path, "exec"), namespace)
      return namespace[name]
  raise KeyError(name)


def main():
  ns = dict(np=np, sparse=sparse, deepcopy=deepcopy, warnings=warnings)
  corrupt = lift("sisua/data/utils.py", "apply_artificial_corruption", ns)
  libsize = lift("sisua/data/utils.py", "get_library_size", ns)
  split = lift("sisua/data/single_cell_dataset.py", "split", dict(np=np))

  rng = np.random.default_rng(8)
  x = (rng.poisson(2.0, size=(48, 37)) * (rng.uniform(size=(48, 37)) < 0.45)).astype(np.float32)
  x[3, :] = 0  # an empty cell
  x[5, 7] = 10738  # heavy tail (cortex max, description/dataset.html:31)
  out = dict(x=x)
  for seed in (8, 1):
    out[f"corrupt_seed{seed}"] = corrupt(x, dropout=0.2, retain_rate=0.2, copy=True, seed=seed)
  out["corrupt_d35_r50_seed8"] = corrupt(x, dropout=0.35, retain_rate=0.5, copy=True, seed=8)
  xs = x.copy()
  xs[3, 0] = 1  # library size needs positive totals
  lc, lm, lv = libsize(xs, return_log_count=True)
  out.update(lib_x=xs, lib_log_counts=lc.astype(np.float64), lib_local_mean=lm, lib_local_var=lv)

  class _Self:  # the minimum SingleCellOMIC surface `split` touches
    def __init__(self, n):
      self.n_obs = n
    def _record(self, *a, **k):
      pass
    def copy(self):
      return self
    def __getitem__(self, ids):
      return np.asarray(ids)

  for n, pct, seed in ((100, 0.8, 1), (3005, 0.8, 1), (2404, 0.9, 1), (4697, 0.8, 1), (3757, 0.9, 1), (57, 0.5, 8)):
    tr, te = split(_Self(n), train_percent=pct, copy=True, seed=seed)
    out[f"split_{n}_{int(pct * 100)}_{seed}_train"] = tr
    out[f"split_{n}_{int(pct * 100)}_{seed}_test"] = te
  np.savez_compressed(OUT, **out)
  print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
  main()
