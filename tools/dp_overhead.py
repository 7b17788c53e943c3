"""Overhead of the data-parallel machinery with ONE rank (SMX_FORCE_ALLREDUCE=1: the whole data-parallel path runs, the
collective moves nothing): the RCCL all-reduce and the hand-written peer-to-peer exchange (two launches + two flag rounds
against itself), single collective and the two-bucket overlap, at the benchmark workload.  What it prices is everything of the
N > 1 step EXCEPT link time: the extra launches, the norm pass after the collective, the cross-stream events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sisua_amd.engine import Engine
import bench
os.environ.setdefault("SMX_FORCE_ALLREDUCE", "1")   # the 1-rank communicator really runs the data-parallel path
cfg, xt, batch, _ = bench.build_workload(0, 1, sys.argv[1] if len(sys.argv) > 1 else "8kly")
order = bench.make_order(xt.shape[0], batch, 330)
for comm in ("none", "rccl", "p2p"):
  for buckets in ((1,) if comm == "none" else (1, 2)):
    os.environ["SMX_DP_BUCKETS"] = str(buckets)
    e = Engine(cfg, max_batch=batch); e.upload(xt)
    if comm == "rccl": e.comm_init(0, 1, Engine.comm_unique_id())
    if comm == "p2p": e.comm_p2p_init(0, 1, e.comm_p2p_export(1))
    e.train_steps(order[:30 * batch], 30, batch, graph=False); e.synchronize()
    t = time.perf_counter(); e.train_steps(order[30 * batch:], 300, batch, graph=False); e.synchronize(); dt = time.perf_counter() - t
    print(f"collective={comm} buckets={buckets}: {dt / 300 * 1e6:.1f} us/step", flush=True)
    if comm == "p2p": assert e.comm_p2p_error() == 0
    e.close()
