#!/usr/bin/env python3
"""Likelihood roofline away from the launch-latency regime: the standalone count_loss_kernel<ZINB> fwd+bwd and the
likelihood-attributable time of the fused output head (fused kernel - product only), both
at the per-GPU size of BASELINE.json configs[4] (G = 20000, 128 cells per GPU) and at larger batches.
HIP events on the model's stream (smx_timing_*), synthetic log-normal counts thinned to ~93 % zeros."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from sisua_amd.config import ModelConfig
from sisua_amd.engine import Engine

HBM_PEAK = 8000.0
out = []
SIZES = ((1998, 128), (20000, 128), (20000, 512), (20000, 1024))
if os.environ.get("LOSS_SIZES"):   # e.g. LOSS_SIZES=1998x256,5000x128
  SIZES = tuple(tuple(int(v) for v in t.split("x")) for t in os.environ["LOSS_SIZES"].split(","))
for G, B in SIZES:
  rng = np.random.default_rng(8)
  n = max(2 * B, 512)
  x = np.floor(rng.lognormal(0.0, 1.0, size=(n, G))).astype(np.float32) * (rng.uniform(size=(n, G)) < 0.12)
  x[:, 0] += 1
  cfg = ModelConfig(model="vae", n_genes=G, likelihood="zinb", enc_units=(128,), dec_units=(128,), latent_dim=32)
  e = Engine(cfg, max_batch=B)
  e.upload(x)
  order = (np.arange(40 * B) % n).astype(np.int32)
  e.train_steps(order[: 5 * B], 5, B, graph=False)

  def timed(label):
    e.timing_enable(label)
    e.train_steps(order, 40, B, graph=False)
    ms, cnt = e.timing_read()
    return 1e3 * ms / max(cnt, 1)

  null = timed("null")                       # event pair around nothing
  fused, prod = timed("out_head"), timed("out_head_product")   # 8 back-to-back launches per event pair (bench.py)
  e.set_flag("head_loss", False)
  alone = timed("loss")
  e.set_flag("head_loss", True)
  e.timing_enable(None)
  bytes_per_launch = e.loss_bytes_per_cell() * B
  us_alone, us_attr = (alone - null) / 8, (fused - prod) / 8
  row = dict(G=G, B=B, bytes_per_launch=bytes_per_launch,
             standalone_us=round(us_alone, 2), standalone_GBs=round(bytes_per_launch / (us_alone * 1e-6) / 1e9, 1),
             standalone_frac=round(bytes_per_launch / (us_alone * 1e-6) / 1e9 / HBM_PEAK, 4),
             fused_kernel_us=round((fused - null) / 8, 2), product_only_us=round((prod - null) / 8, 2),
             attributable_us=round(us_attr, 2), attributable_GBs=round(bytes_per_launch / (us_attr * 1e-6) / 1e9, 1),
             attributable_frac=round(bytes_per_launch / (us_attr * 1e-6) / 1e9 / HBM_PEAK, 4))
  out.append(row)
  print(row, flush=True)
  e.close()
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "loss_roofline.json"), "w"), indent=1)
