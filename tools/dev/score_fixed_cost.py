"""What one marginal_log_prob call costs beside its head launch: the call by the host's clock at 1, 2, 10, 25, 50, 100 draws (8kly shape, 128 cells)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from sisua_amd import _hip
from sisua_amd.engine import Engine

cfg, xt, batch, _ = bench.build_workload(0, 1, "8kly")
e = Engine(cfg, max_batch=batch)
e.upload(xt)
e.train_steps(bench.make_order(xt.shape[0], batch, 20), 20, batch)
rows = np.arange(batch, dtype=np.int32)
for knobs in ({}, {"no_score_dec1": 1}):
  for k, v in knobs.items():
    _hip.set_tuning(k, v)
  for S in (1, 2, 10, 25, 50, 100, 200):
    for _ in range(20):
      e.marginal_llk(row_ids=rows, n_samples=S)
    ts = []
    for rep in range(3):
      t0 = time.perf_counter()
      for _ in range(200):
        e.marginal_llk(row_ids=rows, n_samples=S)
      ts.append((time.perf_counter() - t0) / 200 * 1e6)
    print(f"{knobs} draws {S:4d}: {min(ts):7.1f} us per call", flush=True)
  for k in knobs:
    _hip.clear_tuning(k)
t0 = time.perf_counter()
for _ in range(2000):
  e._ids(rows)
print("engine._ids alone: %.2f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
e.close()
