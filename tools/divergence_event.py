"""Why GPU and oracle trajectories separate: find the first optimiser step at which ANY gradient tensor differs between
the two by more than 1e-3 (relative L2; rounding leaves ~1e-6), and show the hidden unit / cell behind it: a ReLU whose
input is within float32 rounding of 0 is 'on' in one arithmetic and 'off' in the other (tools/divergence_trace.py shows
the consequence over the following steps).  The step at which this happens depends on the build (any change of
summation order moves it).  TRAJ_STEPS (default 320)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from oracle import sisua_oracle as so
from sisua_amd.engine import Engine
n_steps = int(os.environ.get("TRAJ_STEPS", "320"))
THRESH = float(os.environ.get("TRAJ_THRESH", "1e-3"))
if os.environ.get("TRAJ_WORKLOAD", "8kly") == "c5":   # the 20 000-gene trajectory of tests/golden/make_c5_trajectory.py
  from tests.golden import make_c5_trajectory as fxgen
  cfg, xt, batch, order, _ = fxgen.inputs()
  n_steps = min(n_steps, len(order) // batch)
else:
  cfg, xt, batch, _ = bench.build_workload(0, 1, "8kly")
  order = bench.make_order(xt.shape[0], batch, n_steps)
spec = so.Spec(**cfg.to_dict())
params = so.init_params(spec)
bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
e = Engine(cfg, max_batch=batch, init=False)
e.set_params(params); e.upload(xt)
x64 = xt.astype(np.float64)

captured = {}
_orig = so._mlp_fwd
def _capture(spec_, params_, bn_state, prefix, units, h, training, noise, stream0, p_drop, new_bn):
  out, caches = _orig(spec_, params_, bn_state, prefix, units, h, training, noise, stream0, p_drop, new_bn)
  captured[prefix] = caches
  return out, caches
so._mlp_fwd = _capture

for s in range(n_steps):
  rows = order[s * batch:(s + 1) * batch]
  p_before = {k: v.copy() for k, v in params.items()}
  ref = so.train_step(spec, params, bn, opt, x64[rows], so.PhiloxNoise(spec.seed, s, rows))
  e.train_step(rows)
  got = e.get_params(which=1)
  err = {k: np.linalg.norm(got[k] - ref["grads"][k]) / max(np.linalg.norm(ref["grads"][k]), 1e-30) for k in ref["grads"]}
  worst = max(err, key=err.get)
  print(f"step {s + 1}: worst gradient difference {err[worst]:.1e} ({worst})", flush=True)
  if err[worst] > THRESH:
    print(f"step {s + 1}: first gradient difference beyond rounding: " + ", ".join(f"{k} {v:.1e}" for k, v in sorted(err.items(), key=lambda kv: -kv[1])[:4])
          + f"; every tensor agreed to {prev_worst:.1e} at step {s}")
    # ReLU inputs of this step's forward pass (float64, the parameters before the update), nearest to 0
    for prefix, units in (("enc", spec.enc_units), ("dec", spec.dec_units)):
      for i, _ in enumerate(units):
        c = captured[prefix][i]
        y = p_before[f"{prefix}{i}/gamma"] * c["xhat"] + p_before[f"{prefix}{i}/beta"]
        pre = c["h_in"] @ p_before[f"{prefix}{i}/W"]
        b, u = np.unravel_index(np.abs(y).argmin(), y.shape)
        res = np.abs(pre[b, u]) * 6e-8 * abs(p_before[f"{prefix}{i}/gamma"][u]) * c["inv"][u]
        n_sum = c["h_in"].shape[1]
        print(f"  {prefix}{i}: ReLU input nearest to 0: cell {b}, unit {u}: y = {y[b, u]:+.3e} (typical |y| {np.abs(y).mean():.2f}); float32 resolution there: "
              f"|pre| {abs(pre[b, u]):.2f} x 6e-8 x gamma/std {abs(p_before[f'{prefix}{i}/gamma'][u]) * c['inv'][u]:.2f} = {res:.1e} per rounding, {n_sum} terms in the sum")
        # where the difference sits: the column of this unit in the layer's weight gradient
        g, r = got[f"{prefix}{i}/W"].astype(np.float64), ref["grads"][f"{prefix}{i}/W"]
        d = np.abs(g - r)
        scale = np.abs(r).max()
        col = (d > 1e-4 * scale)
        print(f"       weight-gradient entries off by more than 1e-4 of the largest: {int(col[:, u].sum())} in column {u}, {int(col.sum() - col[:, u].sum())} elsewhere")
    break
  prev_worst = err[worst]
else:
  print(f"no gradient difference beyond rounding in {n_steps} steps (worst {prev_worst:.1e})")
