import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from sisua_amd.engine import Engine
cfg, xt, batch, _ = bench.build_workload(0, 1, "8kly")
e = Engine(cfg, max_batch=batch); e.upload(xt)
order = bench.make_order(xt.shape[0], batch, 400)
e.train_steps(order[:30 * batch], 30, batch, graph=False); e.synchronize()
for staged in (False, True, False, True):
  ts = []
  for rep in range(5):
    o = order[(30 + 20 * rep) * batch:(50 + 20 * rep) * batch]
    if staged: e.stage_steps(o, 20, batch)
    e.synchronize()
    t = time.perf_counter(); e.train_steps(None if staged else o, 20, batch, graph=False); e.synchronize(); ts.append((time.perf_counter() - t) / 20 * 1e6)
  print("staged" if staged else "passed", [round(v, 1) for v in ts])
