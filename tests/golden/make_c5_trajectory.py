#!/usr/bin/env python3
"""The float64 oracle's 24-step trajectory at the per-GPU shape of BASELINE.json configs[4] (bench.py's "c5-shard" workload cut to
512 resident cells: 20 000 genes, batch 128, Philox noise keyed by (seed, step, cell)) -- the width at which the wide-panel kernels
run (bf16 x 3 products, smx_bigk.hip, smx_panel.h):

  oracle_c5_trajectory.npz   loss / nllk_x / kl of every step, the eval-mode latent means and scales of 128 probe cells after the
                             24 steps, the row order, and a checksum of the input matrix

tests/test_oracle_golden.py re-runs the first steps of the oracle against it; tests/test_gpu_configs.py holds the HIP path against it.

Run:  python tests/golden/make_c5_trajectory.py      (about a minute)
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import sisua_oracle as so  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
STEPS, N_CELLS = 24, 512
ORDER_SEED = 28   # chosen by `--search` (below): no ReLU input of these steps lies within 3e-6 of zero
KINK_MARGIN = 3e-6


def inputs(order_seed=None):
  cfg, xt, batch, _ = bench.build_workload(0, 1, "c5-shard", n_cells=N_CELLS)
  rng = np.random.default_rng(ORDER_SEED if order_seed is None else order_seed)
  order = np.concatenate([rng.permutation(N_CELLS)[:batch] for _ in range(STEPS)]).astype(np.int32)
  probe = rng.permutation(N_CELLS)[:128].astype(np.int32)
  return cfg, xt, batch, order, probe


def checksum(xt):
  return zlib.crc32(np.ascontiguousarray(xt, dtype=np.float32).tobytes())


def run(n_steps=STEPS, order_seed=None):
  """(the trajectory, the smallest |ReLU input| met on the way).  Two floating-point trajectories of this optimiser separate
  when a ReLU input lies within rounding of zero (at 20 000 genes such an input turns up about once in 40
  steps); a fixture for a 1e-4 comparison must not contain one -- `--search` tries row orders until none is nearer than
  KINK_MARGIN."""
  cfg, xt, B, order, probe = inputs(order_seed)
  so.KINK_LOG = []
  spec = so.Spec(**cfg.to_dict())
  params = so.init_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  out = dict(order=order, probe=probe, x_crc32=np.uint32(checksum(xt)), x_shape=np.array(xt.shape))
  loss, nllk, kl = [], [], []
  for s in range(n_steps):
    rows = order[s * B:(s + 1) * B]
    r = so.train_step(spec, params, bn, opt, xt[rows], so.PhiloxNoise(spec.seed, s, rows))
    loss.append(r["loss"]); nllk.append(r["metrics"]["nllk_x"]); kl.append(r["metrics"]["kl"])
  if n_steps == STEPS:
    f = so.forward_backward(spec, params, bn, xt[probe], so.PhiloxNoise(spec.seed, 0, probe), training=False, backward=False)
    out["z_mean"] = np.asarray(f["z_mean"], np.float64)
    out["z_scale"] = np.asarray(f["z_scale"], np.float64)
  out.update(loss=np.array(loss), nllk_x=np.array(nllk), kl=np.array(kl))
  margin = min(so.KINK_LOG) if so.KINK_LOG else 1.0
  so.KINK_LOG = None
  out["kink_margin"] = np.float64(margin)
  return out


if __name__ == "__main__":
  if "--search" in sys.argv:
    for seed in range(21, 200):
      out = run(order_seed=seed)
      print("order seed", seed, "smallest |ReLU input|", float(out["kink_margin"]), flush=True)
      if out["kink_margin"] > KINK_MARGIN:
        print("-> set ORDER_SEED =", seed)
        break
    sys.exit(0)
  out = run()
  assert out["kink_margin"] > KINK_MARGIN, ("a ReLU input within rounding of zero: run --search", float(out["kink_margin"]))
  np.savez_compressed(os.path.join(HERE, "oracle_c5_trajectory.npz"), **out)
  print("loss", out["loss"][0], "->", out["loss"][-1], "kink margin", float(out["kink_margin"]))
