"""How long does the host take to ENQUEUE a training step (smx_train_steps returns when its launches are queued) next to the time
the device takes to run it?  If the two are close the step is host-bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from sisua_amd.engine import Engine
for w in (sys.argv[1:] or ["8kly", "cortex-base"]):
  cfg, xt, batch, extra = bench.build_workload(0, 1, w)
  e = Engine(cfg, max_batch=batch); e.upload(xt, extra.get("labels", ()), extra.get("library"), extra.get("label_mask"))
  order = bench.make_order(xt.shape[0], batch, 630)
  e.train_steps(order[:30 * batch], 30, batch, graph=False); e.synchronize()
  for n in (300, 600):
    e.stage_steps(order[30 * batch:(30 + n) * batch], n, batch); e.synchronize()
    t0 = time.perf_counter(); e.train_steps(None, n, batch, graph=False); t1 = time.perf_counter(); e.synchronize(); t2 = time.perf_counter()
    print(f"{w}: {n} steps: enqueue returned after {(t1 - t0) / n * 1e6:.1f} us per step, device done after {(t2 - t0) / n * 1e6:.1f} us per step", flush=True)
  e.close()
