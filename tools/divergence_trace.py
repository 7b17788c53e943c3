import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from oracle import sisua_oracle as so
from sisua_amd.engine import Engine
n_steps = int(os.environ.get("TRAJ_STEPS", "400"))
cfg, xt, batch, _ = bench.build_workload(0, 1, "8kly")
spec = so.Spec(**cfg.to_dict())
params = so.init_params(spec)
bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
e = Engine(cfg, max_batch=256, init=False)
e.set_params(params); e.upload(xt)
order = bench.make_order(xt.shape[0], batch, n_steps)
x64 = xt.astype(np.float64)
probe = np.random.default_rng(0).permutation(xt.shape[0])[:256].astype(np.int32)
def rl2(a, b): return np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30)
for s in range(n_steps):
  rows = order[s * batch:(s + 1) * batch]
  ref = so.train_step(spec, params, bn, opt, x64[rows], so.PhiloxNoise(spec.seed, s, rows))
  got = e.train_step(rows)
  if (s + 1) % 20 == 0:
    gp = e.get_params()
    gm = e.get_params(2)
    d = {k: rl2(gp[k], params[k]) for k in gp}
    worst = sorted(d.items(), key=lambda kv: -kv[1])[:3]
    r = so.forward_backward(spec, params, bn, x64[probe], so.PhiloxNoise(spec.seed, 0, probe), training=False, backward=False)
    out = e.forward(row_ids=probe, want_x_params=False)
    # max abs diff per tensor, and where
    k0 = worst[0][0]; dd = np.abs(gp[k0] - params[k0]); idx = np.unravel_index(dd.argmax(), dd.shape)
    gb = e.get_bn()
    bnd = max(rl2(gb[i]["moving_var"], bn[f"{n}/moving_var"]) for i, (n, _) in enumerate(so.bn_manifest(spec)))
    print(f"step {s+1:4d} loss rel {abs(got['loss']/ref['loss']-1):.1e} zmean {rl2(out['z_mean'], r['z_mean']):.1e} bnvar {bnd:.1e} worst {[(k, f'{v:.1e}') for k, v in worst]} argmax {k0}{idx} {dd.max():.2e}", flush=True)
