"""Step time of the model variants that bench.py has no workload for (same 8kly-shaped matrix, batch 128)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from sisua_amd.config import ModelConfig
from sisua_amd.engine import Engine

_, xt, batch, _ = bench.build_workload(0, 1, "8kly")
base = dict(n_genes=xt.shape[1], enc_units=(128,), dec_units=(128,), latent_dim=32, dropout_enc=0.1, dropout_dec=0.1, seed=8)
variants = {
  "vae zinb (bench)": dict(model="vae", likelihood="zinb"),
  "vae nb": dict(model="vae", likelihood="nb"),
  "vae zinbd": dict(model="vae", likelihood="zinbd"),
  "vae zinb no-BN": dict(model="vae", likelihood="zinb", batchnorm=False),
  "vae zinb input-dropout 0.3": dict(model="vae", likelihood="zinb", input_dropout=0.3),
  "dca zinb": dict(model="dca", likelihood="zinb"),
  "vae zinb units [256,128]": dict(model="vae", likelihood="zinb", enc_units=(256, 128), dec_units=(128, 256)),
  "vae zinb latent 10": dict(model="vae", likelihood="zinb", latent_dim=10),
  "vae zinb units [64,64]": dict(model="vae", likelihood="zinb", enc_units=(64, 64), dec_units=(64, 64)),
  "scale zinb (10 components)": dict(model="scale", likelihood="zinb", n_components=10),
  "scale zinb tied loc / scale": dict(model="scale", likelihood="zinb", n_components=10, tie_loc=True, tie_scale=True),
  "scale zinb covariance tril": dict(model="scale", likelihood="zinb", n_components=10, covariance="tril"),
  "scale zinb mixture posterior (8)": dict(model="scale", likelihood="zinb", n_components=8, latent_mixture=True),
  "fvae zinb (5 x 1000 discriminator)": dict(model="fvae", likelihood="zinb", disc_units=1000, disc_layers=5),
}
order = bench.make_order(xt.shape[0], batch, 330)
for name, kw in variants.items():
  cfg = ModelConfig(**dict(base, **kw))
  e = Engine(cfg, max_batch=batch); e.upload(xt)
  e.train_steps(order[:30 * batch], 30, batch, graph=False); e.synchronize()
  t = time.perf_counter(); m = e.train_steps(order[30 * batch:], 300, batch, graph=False, metrics=True); e.synchronize()
  dt = time.perf_counter() - t
  print(f"{name:32s} {dt / 300 * 1e6:7.1f} us/step  {batch * 300 / dt / 1e6:.2f} M cells/s  loss {m['loss']:.3f}", flush=True)
  e.close()
