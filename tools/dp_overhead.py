"""Overhead of the data-parallel machinery on a 1-rank RCCL communicator (SMX_FORCE_ALLREDUCE=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sisua_amd.engine import Engine
import bench
os.environ.setdefault("SMX_FORCE_ALLREDUCE", "1")   # the 1-rank communicator really runs the data-parallel path
cfg, xt, batch, _ = bench.build_workload(0, 1, "8kly")
for comm in (False, True):
  for graph in (False, True):
    e = Engine(cfg, max_batch=batch); e.upload(xt)
    if comm: e.comm_init(0, 1, Engine.comm_unique_id())
    order = bench.make_order(xt.shape[0], batch, 330)
    e.train_steps(order[:30 * batch], 30, batch, graph=graph); e.synchronize()
    t = time.perf_counter(); e.train_steps(order[30 * batch:], 300, batch, graph=graph); e.synchronize(); dt = time.perf_counter() - t
    print(f"comm={comm} graph={graph} buckets={os.environ.get('SMX_DP_BUCKETS', '1')}: {dt / 300 * 1e6:.1f} us/step", flush=True)
    e.close()
