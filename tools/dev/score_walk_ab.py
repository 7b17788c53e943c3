"""A/B of the scoring head's two forms on one box: score_head_kernel (one 128 x 32 tile per workgroup; knob score_walk=0) against score_walk_kernel
(a workgroup walks 256-row blocks under its gene tile's resident W image), bit for bit and by the host's clock around smx_marginal_llk."""
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
S = int(os.environ.get("DRAWS", "100"))
N = int(os.environ.get("CALLS", "200"))

def run(label, knobs):
  for k, v in knobs.items():
    _hip.set_tuning(k, v)
  try:
    for _ in range(10):
      out = e.marginal_llk(row_ids=rows, n_samples=S)
    ts = []
    for rep in range(3):
      t0 = time.perf_counter()
      for _ in range(N):
        out = e.marginal_llk(row_ids=rows, n_samples=S)
      ts.append((time.perf_counter() - t0) / N * 1e6)
  finally:
    for k in knobs:
      _hip.clear_tuning(k)
  print(f"{label:34s} {min(ts):7.1f} us per call (of {' '.join(f'{t:.1f}' for t in ts)})", flush=True)
  return out

variants = [("tile per workgroup (score_walk=0)", {"score_walk": 0}), ("walk", {})]
for d in os.environ.get("DEPHASE", "").split():
  variants.append((f"walk, dephase {d}", {"score_walk_dephase": float(d)}))
variants.append(("walk, decoder in three launches", {"no_score_dec1": 1}))
for s in os.environ.get("SPLITS", "").split():
  variants.append((f"walk, {s} ranges", {"score_walk": float(s)}))
ref = None
for label, knobs in variants + variants[:2]:
  out = run(label, knobs)
  if ref is None:
    ref = out
  elif "three launches" in label:   # (the product's summation order differs: close, not equal)
    print("   max |difference| of log p(x):", float(np.abs(out[0] - ref[0]).max()), "of", float(np.abs(ref[0]).mean()))
    assert np.allclose(out[0], ref[0], rtol=1e-5, atol=1e-3)
  else:
    assert np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1]), label
print("every form: the same bits")
e.close()
