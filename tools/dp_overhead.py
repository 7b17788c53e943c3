"""Overhead of the data-parallel machinery with ONE rank (SMX_FORCE_ALLREDUCE=1: the whole data-parallel path runs, the
collective moves nothing): RCCL and the hand-written peer-to-peer exchange, one bucket against two -- since round 5 the two-bucket
step is the CHAIN on the communication stream (head bucket all-reduce -> norms -> clip + Adam, joined in front of the next output head;
the hand-written exchange keeps round 4's form).  What it prices is everything of the N > 1 step EXCEPT link time: the extra
launches, the norm pass after the collective, the second queue.   usage: dp_overhead.py [workload ...]   (default: 8kly c5-shard)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sisua_amd.engine import Engine
import bench
os.environ.setdefault("SMX_FORCE_ALLREDUCE", "1")   # the 1-rank communicator really runs the data-parallel path
for workload in (sys.argv[1:] or ["8kly", "c5-shard"]):
  cfg, xt, batch, _ = bench.build_workload(0, 1, workload)
  order = bench.make_order(xt.shape[0], batch, 330)
  for comm in ("none", "rccl", "p2p"):
    for buckets in ((1,) if comm == "none" else (1, 2)):
      os.environ["SMX_DP_BUCKETS"] = str(buckets)
      e = Engine(cfg, max_batch=batch); e.upload(xt, storage="u16" if workload.startswith("c5") else "f32")
      if comm == "rccl": e.comm_init(0, 1, Engine.comm_unique_id())
      if comm == "p2p": e.comm_p2p_init(0, 1, e.comm_p2p_export(1))
      e.train_steps(order[:30 * batch], 30, batch, graph=False); e.synchronize()
      best = 1e9
      for _ in range(3):
        t = time.perf_counter(); m = e.train_steps(order[30 * batch:], 300, batch, graph=False, metrics=True); e.synchronize(); best = min(best, time.perf_counter() - t)
      print(f"{workload}: collective={comm} buckets={buckets}: {best / 300 * 1e6:.1f} us/step   loss {m['loss']:.4f}", flush=True)
      if comm == "p2p": assert e.comm_p2p_error() == 0
      e.close()
