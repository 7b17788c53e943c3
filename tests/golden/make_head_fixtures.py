#!/usr/bin/env python3
"""Known-answer fixtures of the model variants added in round 3 (float64 oracle, Philox noise): one training step each at
off-initialisation parameters -- inputs, labels, mask, parameters before, loss terms and EVERY gradient.

  oracle_variants_fixture.npz
    misa_tril   SISUA with a 'mixtril2' head (ONE full-covariance Gaussian mixture over 6 label dimensions, vae.py:58) + a 'mixzinb2' head
                (MISA(zero_inflated=True), vae.py:76-84)
    scale_tril  SCALE with covariance='tril' (scale.py:28,35): a lower-triangular scale factor per mixture component

Run:  python tests/golden/make_head_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sisua_oracle as so  # noqa: E402
from tests.util import perturbed_params, synth_counts, synth_labels  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {
    "misa_tril": dict(model="sisua", n_genes=48, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=4,
                      labels=((6, "mixtril2"), (5, "mixzinb2")), alpha=10.0, seed=8),
    "scale_tril": dict(model="scale", n_genes=40, likelihood="nb", enc_units=(16,), dec_units=(16,), latent_dim=5, n_components=3,
                       covariance="tril", seed=8),
}
B, N, STEP, CELL_BASE = 24, 64, 0, 100   # (the first training step of a fresh model draws Philox step 0)


def inputs(name):
  spec = so.Spec(**CASES[name])
  x = synth_counts(N, spec.n_genes, sparsity=0.8, seed=21, max_count=300)
  ys = synth_labels(N, spec.labels, seed=4)
  mask = so.label_mask(N, 0.5, n_omics=1 + len(spec.labels), seed=2)
  rows = np.arange(7, 7 + B, dtype=np.int32)
  return spec, x, ys, mask, rows


def main():
  out = {}
  for name in CASES:
    spec, x, ys, mask, rows = inputs(name)
    params = perturbed_params(spec, scale=0.1, seed=6)
    res = so.forward_backward(spec, params, so.init_bn_state(spec), x[rows], so.PhiloxNoise(spec.seed, STEP, rows + CELL_BASE),
                              y=[y[rows] for y in ys], mask=mask[rows])
    out[f"{name}/x"] = x
    out[f"{name}/mask"] = mask
    for j, y in enumerate(ys):
      out[f"{name}/y{j}"] = y
    for k, v in params.items():
      out[f"{name}/p0/{k}"] = v
      out[f"{name}/g/{k}"] = res["grads"][k]
    for k in ("loss", "nllk_x", "nllk_y", "kl"):
      out[f"{name}/{k}"] = res["metrics"][k]
    print(name, {k: round(float(res["metrics"][k]), 6) for k in ("loss", "nllk_x", "nllk_y", "kl")})
  np.savez_compressed(os.path.join(HERE, "oracle_variants_fixture.npz"), **out)


if __name__ == "__main__":
  main()
