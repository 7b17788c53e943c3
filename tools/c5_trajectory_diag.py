import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import sisua_oracle as so
from tests.golden import make_c5_trajectory as fxgen
from sisua_amd.engine import Engine
cfg, xt, B, order, probe = fxgen.inputs()
spec = so.Spec(**cfg.to_dict())
fx = np.load("tests/golden/oracle_c5_trajectory.npz")
for b3 in (1, 0):
  e = Engine(cfg, max_batch=B, init=False)
  e.set_flag("bf16x3", b3)
  e.set_params(so.init_params(spec)); e.upload(xt, storage="f32")
  errs = []
  for s in range(len(fx["loss"])):
    got = e.train_step(order[s * B:(s + 1) * B])
    errs.append([abs(got[k] / fx[k][s] - 1.0) for k in ("loss", "nllk_x", "kl")])
  errs = np.array(errs)
  print("bf16x3", b3, "max per key", errs.max(0), "argmax step", errs.argmax(0))
  print(" loss err by step:", " ".join(f"{v:.1e}" for v in errs[:, 0]))
  print(" kl   err by step:", " ".join(f"{v:.1e}" for v in errs[:, 2]))
  out = e.forward(row_ids=probe, want_x_params=False)
  for key in ("z_mean", "z_scale"):
    d = np.linalg.norm(out[key] - fx[key]) / np.linalg.norm(fx[key])
    print(" ", key, "rel l2", d)
  e.close()
