#!/usr/bin/env python3
"""The float64 oracle's 300-step trajectory at BASELINE.json configs[1] (bench.py's "8kly" workload: 3381 x 1998 training
cells, batch 128, Philox noise keyed by (seed, step, cell)), committed so that the GPU parity test does not spend 40 s of
every run re-deriving it:

  oracle_c2_trajectory.npz   loss / nllk_x / kl of every step, the eval-mode latent means and scales of 256 probe cells
                             after 100 and after 300 steps, the row order, and a checksum of the input matrix

tests/test_oracle_golden.py re-runs the first steps of the oracle against it (the fixture stays the oracle's output);
tests/test_gpu_configs.py::test_latent_means_after_training_match_oracle holds the HIP path against it.

Run:  python tests/golden/make_c2_trajectory.py      (about a minute)
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
STEPS, PROBE_AT = 300, (100, 300)


def inputs():
  cfg, xt, batch, _ = bench.build_workload(0, 1, "8kly")
  order = bench.make_order(xt.shape[0], batch, STEPS)
  probe = np.random.default_rng(0).permutation(xt.shape[0])[:256].astype(np.int32)
  return cfg, xt, batch, order, probe


def checksum(xt):
  return zlib.crc32(np.ascontiguousarray(xt, dtype=np.float32).tobytes())


def run(n_steps=STEPS):
  cfg, xt, B, order, probe = inputs()
  spec = so.Spec(**cfg.to_dict())
  params = so.init_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  out = dict(order=order, probe=probe, x_crc32=np.uint32(checksum(xt)), x_shape=np.array(xt.shape))
  loss, nllk, kl = [], [], []
  for s in range(n_steps):
    rows = order[s * B:(s + 1) * B]
    r = so.train_step(spec, params, bn, opt, xt[rows], so.PhiloxNoise(spec.seed, s, rows))
    loss.append(r["loss"]); nllk.append(r["metrics"]["nllk_x"]); kl.append(r["metrics"]["kl"])
    if s + 1 in PROBE_AT:
      f = so.forward_backward(spec, params, bn, xt[probe], so.PhiloxNoise(spec.seed, 0, probe), training=False, backward=False)
      out[f"z_mean_{s + 1}"] = np.asarray(f["z_mean"], np.float64)
      out[f"z_scale_{s + 1}"] = np.asarray(f["z_scale"], np.float64)
  out.update(loss=np.array(loss), nllk_x=np.array(nllk), kl=np.array(kl))
  return out


if __name__ == "__main__":
  out = run()
  np.savez_compressed(os.path.join(HERE, "oracle_c2_trajectory.npz"), **out)
  print("loss", out["loss"][0], "->", out["loss"][-1])
