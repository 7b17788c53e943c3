"""Why GPU and oracle trajectories separate: find the first optimiser step at which one encoder-weight gradient differs
by O(1) between the two, and show the state of the hidden unit / cell behind it (tools/divergence_trace.py shows the
consequence over the following steps).  TRAJ_STEPS (default 160)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from oracle import sisua_oracle as so
from sisua_amd.engine import Engine
n_steps = int(os.environ.get("TRAJ_STEPS", "160"))
cfg, xt, batch, _ = bench.build_workload(0, 1, "8kly")
spec = so.Spec(**cfg.to_dict())
params = so.init_params(spec)
bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
e = Engine(cfg, max_batch=batch, init=False)
e.set_params(params); e.upload(xt)
order = bench.make_order(xt.shape[0], batch, n_steps)
x64 = xt.astype(np.float64)
for s in range(n_steps):
  rows = order[s * batch:(s + 1) * batch]
  p_before = {k: v.copy() for k, v in params.items()}
  bn_before = {k: v.copy() for k, v in bn.items()}
  ref = so.train_step(spec, params, bn, opt, x64[rows], so.PhiloxNoise(spec.seed, s, rows))
  e.train_step(rows)
  g = e.get_params(which=1)["enc0/W"].astype(np.float64)
  r = ref["grads"]["enc0/W"]
  d = np.abs(g - r)
  scale = np.abs(r).max()
  i, j = np.unravel_index(d.argmax(), d.shape)
  if d.max() > 1e-3 * scale:
    print(f"step {s + 1}: enc0/W gradient [{i},{j}] gpu {g[i, j]:+.6e} oracle {r[i, j]:+.6e} (largest |g| {scale:.2e}); "
          f"all other entries agree to {np.sort(d.ravel())[-20] / scale:.1e} of the largest")
    # the oracle's ReLU input of unit j for every cell of the batch (training-mode BatchNorm of this batch)
    h0 = np.log1p(x64[rows])
    pre = h0 @ p_before["enc0/W"]
    mu, var = pre.mean(0), pre.var(0)
    y = p_before["enc0/gamma"] * (pre - mu) / np.sqrt(var + spec.bn_eps) + p_before["enc0/beta"]
    col = np.abs(g[:, j] - r[:, j])
    print(f"  column {j} (hidden unit {j}): {int((col > 1e-4 * scale).sum())} of {col.size} gene rows differ by more than 1e-4 of the largest gradient;"
          f" other columns: {int((np.delete(d, j, axis=1) > 1e-4 * scale).sum())} entries")
    order_y = np.argsort(np.abs(y[:, j]))
    print(f"  ReLU input of unit {j}, the three cells closest to 0 (typical |y| {np.abs(y[:, j]).mean():.2f}): "
          + ", ".join(f"cell {c}: {y[c, j]:+.3e}" for c in order_y[:3]))
    print(f"  float32 resolution of the pre-activation there: |pre| {np.abs(pre[order_y[0], j]):.2f} * 6e-8 * gamma/std {abs(p_before['enc0/gamma'][j]) / np.sqrt(var[j] + spec.bn_eps):.2f}"
          f" = {np.abs(pre[order_y[0], j]) * 6e-8 * abs(p_before['enc0/gamma'][j]) / np.sqrt(var[j] + spec.bn_eps):.1e}")
    break
else:
  print(f"no O(1) gradient difference in {n_steps} steps")
