"""The fixed cost of one smx_train_steps call: K staged steps by the host's clock at K = 1, 5, 20, 100 (8kly shape), and the slope between them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from sisua_amd.engine import Engine

cfg, xt, batch, _ = bench.build_workload(0, 1, "8kly")
e = Engine(cfg, max_batch=batch)
e.upload(xt)
order = bench.make_order(xt.shape[0], batch, 400)
for _ in range(200):
  e.eval_step(order[:batch])
e.train_steps(order[: 50 * batch], 50, batch)
res = {}
for K in [int(v) for v in os.environ.get("KS", "1 5 20 100").split()]:
  ts = []
  for rep in range(int(os.environ.get("REPS", "15"))):
    e.stage_steps(order[: K * batch], K, batch)
    for _ in range(int(os.environ.get("EVAL_BEFORE", "0"))):   # (evaluation passes right in front of the bracket: the device's clocks are up when the clock starts)
      e.eval_step(order[:batch])
    e.synchronize()
    t0 = time.perf_counter()
    e.train_steps(None, K, batch)
    e.synchronize()
    ts.append((time.perf_counter() - t0) * 1e6)
  res[K] = float(np.median(ts))
  print("   calls in order:", " ".join(f"{t:.0f}" for t in ts))
  print(f"K = {K:4d}: {res[K]:9.1f} us per call (median), {res[K] / K:7.2f} us per step", flush=True)
ks = sorted(res)
if len(ks) >= 2:
  slope = (res[ks[-1]] - res[ks[-2]]) / (ks[-1] - ks[-2])
  print(f"slope {slope:.2f} us per step; fixed cost of a call at K = {ks[-2]}: {res[ks[-2]] - slope * ks[-2]:.1f} us")
e.close()
