"""ELBO trajectory of the benchmark configuration (8kly-shaped, vae zinb 128/32, batch 128, dropout 0.1) on the GPU
against the float64 oracle with the same Philox noise: relative difference of the loss per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from oracle import sisua_oracle as so
from sisua_amd.engine import Engine

n_steps = int(os.environ.get("TRAJ_STEPS", "300"))
cfg, xt, batch, _ = bench.build_workload(0, 1, "8kly")
spec = so.Spec(**cfg.to_dict())
params = so.init_params(spec)
bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
e = Engine(cfg, max_batch=batch, init=False)
e.set_params(params)
e.upload(xt)
order = bench.make_order(xt.shape[0], batch, n_steps)
x64 = xt.astype(np.float64)
rel = []
for s in range(n_steps):
  rows = order[s * batch:(s + 1) * batch]
  ref = so.train_step(spec, params, bn, opt, x64[rows], so.PhiloxNoise(spec.seed, s, rows))["metrics"]["loss"]
  got = e.train_step(rows)["loss"]
  rel.append(abs(got / ref - 1.0))
  if s in (0, 9, 49, 99, 199, 299, 499, 999) or s == n_steps - 1:
    print(f"step {s + 1:5d}: gpu {got:.5f} oracle {ref:.5f} rel {rel[-1]:.2e}  (max so far {max(rel):.2e})", flush=True)
# latent means of 256 cells after training: eval-mode encoders of both (moving BatchNorm statistics, no dropout)
rows = np.arange(256, dtype=np.int32)
zm = np.concatenate([e.forward(row_ids=rows[i:i + batch], want_x_params=False)["z_mean"] for i in range(0, len(rows), batch)])
ref = so.forward_backward(spec, params, bn, x64[rows], so.PhiloxNoise(spec.seed, 0, rows), training=False, backward=False)
num = np.linalg.norm(zm - ref["z_mean"]); den = np.linalg.norm(ref["z_mean"])
print(f"step {n_steps:5d}: latent means of 256 cells: rel-L2 {num / den:.2e}, max abs {np.abs(zm - ref['z_mean']).max():.2e} "
      f"(|z_mean| up to {np.abs(ref['z_mean']).max():.2f})", flush=True)
e.close()
