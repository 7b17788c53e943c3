"""Overhead of the data-parallel machinery with ONE rank (SMX_FORCE_ALLREDUCE=1: the whole data-parallel path runs, the collective
moves nothing), per exchange FORM (Engine.comm_form): 1 one all-reduce through RCCL, 2 the two-bucket chain (head bucket all-reduce ->
norms -> clip + Adam on the communication stream), 3 the hand-written exchange (round 6: one launch per all-reduce).  What it prices is
everything of the N > 1 step EXCEPT link time: the extra launches, the norm pass behind the collective, the second queue.  Then
parallel.calibrate_forms on the same engine: the selection an N > 1 job makes, rehearsed on one rank.
usage: dp_overhead.py [workload ...]   (default: 8kly c5-shard)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sisua_amd.engine import Engine
from sisua_amd import parallel
import bench
os.environ.setdefault("SMX_FORCE_ALLREDUCE", "1")   # the 1-rank communicator really runs the data-parallel path


class Solo:   # a control plane of one rank
  rank, world = 0, 1
  def barrier(self): pass
  def max(self, v): return float(v)


for workload in (sys.argv[1:] or ["8kly", "c5-shard"]):
  cfg, xt, batch, _ = bench.build_workload(0, 1, workload)
  order = bench.make_order(xt.shape[0], batch, 330)
  e = Engine(cfg, max_batch=batch); e.upload(xt, storage="u16" if workload.startswith("c5") else "f32")
  for form in (0, 1, 2, 3):
    if form == 1:
      e.comm_init(0, 1, Engine.comm_unique_id())
      e.comm_p2p_init(0, 1, e.comm_p2p_export(1))
    if form:
      e.comm_set_form(form)
    e.train_steps(order[:30 * batch], 30, batch, graph=False); e.synchronize()
    best = 1e9
    for _ in range(3):
      t = time.perf_counter(); m = e.train_steps(order[30 * batch:], 300, batch, graph=False, metrics=True); e.synchronize(); best = min(best, time.perf_counter() - t)
    print(f"{workload}: form {form} ({'no communicator' if not form else Engine.FORM_NAMES[e.comm_form]}): {best / 300 * 1e6:.1f} us/step   loss {m['loss']:.4f}", flush=True)
    assert not form or e.comm_form == form, (form, e.comm_form)
    if form == 3: assert e.comm_p2p_error() == 0
  print(f"{workload}: calibrate_forms ->", parallel.calibrate_forms(e, Solo(), "auto", order, batch), flush=True)
  e.close()
