"""One stacked marginal_llk call (8kly shape, 128 cells x 1000 draws) for PMC passes over the scoring head kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from sisua_amd.engine import Engine

cfg, xt, batch, _ = bench.build_workload(0, 1, "8kly")
e = Engine(cfg, max_batch=batch)
e.upload(xt)
e.train_steps(bench.make_order(xt.shape[0], batch, 20), 20, batch)
rows = np.arange(batch, dtype=np.int32)
for _ in range(int(os.environ.get("CALLS", "2"))):
  e.marginal_llk(row_ids=rows, n_samples=int(os.environ.get("DRAWS", "1000")))
e.close()
