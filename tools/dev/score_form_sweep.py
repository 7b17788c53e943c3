"""The scoring head's form by the number of stacked rows: marginal_log_prob of 128 cells at S draws by the host's clock, the tile-per-workgroup form
(score_walk=0), the launcher's own choice, and the walk forced to 1 / 2 / 4 row ranges per gene tile."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from sisua_amd import _hip
from sisua_amd.engine import Engine

cfg, xt, batch, _ = bench.build_workload(0, 1, os.environ.get("WORKLOAD", "8kly"))
e = Engine(cfg, max_batch=batch)
e.upload(xt)
e.train_steps(bench.make_order(xt.shape[0], batch, 20), 20, batch)
rows = np.arange(batch, dtype=np.int32)

def t(S, knob):
  if knob is not None:
    _hip.set_tuning("score_walk", knob)
  try:
    for _ in range(10):
      e.marginal_llk(row_ids=rows, n_samples=S)
    ts = []
    for rep in range(3):
      t0 = time.perf_counter()
      for _ in range(100):
        e.marginal_llk(row_ids=rows, n_samples=S)
      ts.append((time.perf_counter() - t0) / 100 * 1e6)
  finally:
    _hip.clear_tuning("score_walk")
  return min(ts)

print("draws   tile   auto  walk1  walk2  walk4  walk8")
for S in [int(v) for v in os.environ.get("DRAWS", "1 2 4 6 10 15 20 25 35 50 75 100 128").split()]:
  print(f"{S:5d} " + " ".join(f"{t(S, k):6.1f}" for k in (0, None, 1, 2, 4, 8)), flush=True)
e.close()
