"""Timings of the SURVEY 8f rows built on the GPU: scoring path and preprocessing of the resident matrix."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from sisua_amd.engine import Engine

cfg, xt, batch, _ = bench.build_workload(0, 1, "8kly")
e = Engine(cfg, max_batch=batch)
e.upload(xt)
e.train_steps(bench.make_order(xt.shape[0], batch, 20), 20, batch)
rows = np.arange(batch, dtype=np.int32)
x_org = xt[:batch] + 1.0

def timed(f, reps):
  f(); e.synchronize()
  t = time.perf_counter()
  for _ in range(reps): f()
  e.synchronize()
  return (time.perf_counter() - t) / reps

for stacked in (False, True):
  e.set_flag("stacked_scoring", stacked)
  for draws in (100, 1000):
    t = timed(lambda: e.marginal_llk(row_ids=rows, n_samples=draws), 5)
    print(f"marginal_llk ({'stacked draws' if stacked else 'one decoder pass per draw'}): {batch} cells x {draws} draws: {t * 1e3:.2f} ms "
          f"-> {batch / t:.0f} cells/s, {batch * draws / t / 1e6:.2f} M draws/s")
t = timed(lambda: e.score_llk([x_org, None], row_ids=rows, n_samples=10), 10)
print(f"score_llk (2 targets x 2 distributions): {batch} cells x 10 draws: {t * 1e3:.2f} ms -> {batch / t:.0f} cells/s")
# the same for SCVI at the C3 shape (library latent per draw, softmax-rate head: raw planes + a row-local likelihood launch)
cfg_s, xs_, batch_s, extra_s = bench.build_workload(0, 1, "8kly-scvi")
es = Engine(cfg_s, max_batch=batch_s)
es.upload(xs_, library=extra_s["library"])
es.train_steps(bench.make_order(xs_.shape[0], batch_s, 20), 20, batch_s)
rows_s = np.arange(batch_s, dtype=np.int32)
def timed_s(f, reps):
  f(); es.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps): f()
  es.synchronize()
  return (time.perf_counter() - t0) / reps
for stacked in (False, True):
  es.set_flag("stacked_scoring", stacked)
  t = timed_s(lambda: es.marginal_llk(row_ids=rows_s, n_samples=100), 5)
  print(f"scvi marginal_llk ({'stacked draws' if stacked else 'one decoder pass per draw'}): {batch_s} cells x 100 draws: {t * 1e3:.2f} ms "
        f"-> {batch_s / t:.0f} cells/s, {batch_s * 100 / t / 1e6:.2f} M draws/s")
es.close()
t = timed(lambda: e.dataset_library(), 20)
print(f"dataset_library {xt.shape}: {t * 1e6:.0f} us ({xt.nbytes / t / 1e9:.0f} GB/s)")
t0 = time.perf_counter(); n = e.dataset_corrupt(0.2, 0.2, 8); e.synchronize(); t = time.perf_counter() - t0
print(f"dataset_corrupt {xt.shape}: {n} entries in {t * 1e3:.2f} ms (8 radix passes + apply + row constants)")
from sisua_amd import data
t0 = time.perf_counter(); data.corrupt(xt, 0.2, 0.2, seed=8); t = time.perf_counter() - t0
print(f"host corrupt (NumPy RandomState, reference stream): {t * 1e3:.1f} ms")
e.close()
