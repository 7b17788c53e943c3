#!/usr/bin/env python3
"""Known-answer fixtures produced by the float64 oracle, so the HIP path
and any later change of the oracle itself are held against committed numbers:

  oracle_step_fixture.npz        one full VAE/zinb step (B=8, G=64, H=16, D=4) with fixed parameters and
                                 INJECTED eps / dropout masks: loss, every gradient, post-Adam parameters
  oracle_trajectory_fixture.npz  a 50-step seeded (Philox) trajectory on a 512 x 200 synthetic matrix: loss per step

Run:  python tests/golden/make_oracle_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sisua_oracle as so  # noqa: E402
from tests.util import perturbed_params, synth_counts  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
STEP_KW = dict(model="vae", n_genes=64, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=4, input_dropout=0.25,
               dropout_enc=0.2, dropout_dec=0.2, seed=8)
TRAJ_KW = dict(model="vae", n_genes=200, likelihood="zinb", enc_units=(64,), dec_units=(64,), latent_dim=10, seed=8)


def main():
  # ---- one step, injected noise ----
  spec = so.Spec(**STEP_KW)
  rng = np.random.default_rng(2024)
  B = 8
  x = synth_counts(B, 64, sparsity=0.7, seed=11, max_count=181)
  params = perturbed_params(spec, scale=0.1, seed=5)
  def dmask(w, p):
    return (rng.uniform(size=(B, w)) >= p).astype(np.float32) / np.float32(1 - p)
  drop = {so.STREAM_INPUT_DROPOUT: dmask(64, 0.25), so.STREAM_ENC_DROPOUT: dmask(16, 0.2), so.STREAM_DEC_DROPOUT: dmask(16, 0.2)}
  eps = {so.STREAM_EPS_Z: rng.normal(size=(B, 4)).astype(np.float32)}
  p0 = {k: v.copy() for k, v in params.items()}
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  res = so.train_step(spec, params, bn, opt, x, so.InjectedNoise(drop, eps))
  out = dict(x=x, loss=res["loss"], nllk_x=res["metrics"]["nllk_x"], kl=res["metrics"]["kl"])
  for k in p0:
    out[f"p0/{k}"] = p0[k]
    out[f"g/{k}"] = res["grads"][k]
    out[f"p1/{k}"] = params[k]
  for s, v in drop.items():
    out[f"drop/{s}"] = v
  out[f"eps/{so.STREAM_EPS_Z}"] = eps[so.STREAM_EPS_Z]
  for k, v in bn.items():
    out[f"bn/{k}"] = v
  np.savez_compressed(os.path.join(HERE, "oracle_step_fixture.npz"), **out)
  # ---- 50-step trajectory, Philox noise ----
  spec = so.Spec(**TRAJ_KW)
  x = synth_counts(512, 200, sparsity=0.85, seed=0)
  params = so.init_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  Bt, steps = 64, 50
  order = np.concatenate([so.epoch_order(512, ep, shuffle=100, seed=1) for ep in range(8)])[: steps * Bt].astype(np.int32)
  losses, kls = [], []
  for s in range(steps):
    rows = order[s * Bt:(s + 1) * Bt]
    r = so.train_step(spec, params, bn, opt, x[rows], so.PhiloxNoise(spec.seed, s, rows))
    losses.append(r["loss"])
    kls.append(r["metrics"]["kl"])
  np.savez_compressed(os.path.join(HERE, "oracle_trajectory_fixture.npz"), x=x, order=order, loss=np.array(losses), kl=np.array(kls),
                      final_lat_W=params["lat/W"])
  print("loss", res["loss"], "trajectory", losses[0], "->", losses[-1])


if __name__ == "__main__":
  main()
