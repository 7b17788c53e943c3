#!/usr/bin/env python3
"""The fused output head by itself (smx_k_head_fused): parity against float64 NumPy at a few shapes, and its launch time at 128 x 20 000.
usage: headfused_try.py [--time-only LK] [--reps N]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sisua_oracle as so
from sisua_amd import engine as eng


def run(lk, B, G, u16, reps=0, check=True):
  rng = np.random.default_rng(B * 7 + G)
  k = so.n_params_per_gene(lk)
  x = (rng.poisson(3.0, size=(B, G)) * (rng.uniform(size=(B, G)) < 0.1)).astype(np.float32)
  d = np.maximum(rng.normal(size=(B, 128)), 0).astype(np.float32)
  W = (rng.normal(size=(128, k, G)) * 0.08).astype(np.float32)
  bias = (rng.normal(size=(k, G)) * 0.3).astype(np.float32)
  scale = -1.0 / B
  got = eng.k_head_fused(lk, x, d, W, bias, grad_scale=scale, u16=u16, reps=reps)
  if not check:
    print(lk, B, G, u16, "us", got["us"], flush=True)
    return
  d64, W64 = d.astype(np.float64), W.astype(np.float64)
  P = np.einsum("bh,hkg->kbg", d64, W64) + bias.astype(np.float64)[:, None, :]
  ref_e, ref_g = so.count_llk(x.astype(np.float64), list(P), lk)
  dP = scale * np.stack(ref_g)
  dW = np.einsum("bh,kbg->hkg", d64, dP); db = dP.sum(1); dd = np.einsum("kbg,hkg->bh", dP, W64)
  rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
  print(lk, B, G, u16, "llk", np.abs(got["llk"] - ref_e.sum(1)).max(), "dW", rel(got["dW"], dW), "db", rel(got["db"], db), "dd", rel(got["dd"], dd),
        "sumsq", got["sumsq"], (dW**2).sum(), "us", got["us"], flush=True)


if __name__ == "__main__":
  reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 50
  if "--time-only" in sys.argv:
    run(sys.argv[sys.argv.index("--time-only") + 1], 128, 20000, True, reps=reps, check=False)
  else:
    run("zinb", 128, 4096, False)
    run("zinb", 128, 4128, True)
    run("nb", 100, 4100, False)
    run("zinb", 128, 20000, True, reps=reps)
    run("nb", 128, 20000, True, reps=reps)
