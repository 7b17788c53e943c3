"""Step time of MISA's label heads at BASELINE configs[3]'s shape (eccly: 2116 x 2000, 38 ADT label dimensions at 10 %, batch 256): SISUA's NB head,
mixtures of NB / zero-inflated NB per dimension, and ONE full-covariance Gaussian mixture over all 38 dimensions ('mixtril', on log1p of the labels)."""
import os, sys, time, dataclasses
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from sisua_amd.engine import Engine
cfg0, xt, batch, extra = bench.build_workload(0, 1, "eccly-sisua")
P = cfg0.labels[0][0]
order = bench.make_order(xt.shape[0], batch, 330)
for kind in ("nb", "mixnb2", "mixzinb2", "mixtril2"):
  cfg = dataclasses.replace(cfg0, labels=((P, kind),))
  labels = [np.log1p(extra["labels"][0]).astype(np.float32)] if kind.startswith("mixtril") else extra["labels"]
  e = Engine(cfg, max_batch=batch); e.upload(xt, labels, extra.get("library"), extra.get("label_mask"))
  e.train_steps(order[:30 * batch], 30, batch, graph=False); e.synchronize()
  t = time.perf_counter(); m = e.train_steps(order[30 * batch:], 300, batch, graph=False, metrics=True); e.synchronize(); dt = time.perf_counter() - t
  print(f"eccly labels {kind:9s} {dt / 300 * 1e6:7.1f} us/step  loss {m['loss']:.3f}", flush=True)
  e.close()
