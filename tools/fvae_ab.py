"""FactorVAE step (odin's 5 x 1000 discriminator at the C2 shape) with and without the direct-operand bf16 x 3 products of
smx_dgemm.hip / the 32 x 32-tile weight-gradient kernel (SMX_TUNING=no_dgemm: the LDS-tiled f32 kernel for every discriminator product)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from sisua_amd.config import ModelConfig
from sisua_amd.engine import Engine
_, xt, batch, _ = bench.build_workload(0, 1, "8kly")
cfg = ModelConfig(n_genes=xt.shape[1], enc_units=(128,), dec_units=(128,), latent_dim=32, dropout_enc=0.1, dropout_dec=0.1, seed=8, model="fvae", likelihood="zinb", disc_units=1000, disc_layers=5)
for env in ("", "1"):
  if env: os.environ["SMX_TUNING"] = "no_dgemm"
  e = Engine(cfg, max_batch=batch); e.upload(xt)
  order = bench.make_order(xt.shape[0], batch, 330)
  e.train_steps(order[:30 * batch], 30, batch, graph=False); e.synchronize()
  t = time.perf_counter(); m = e.train_steps(order[30 * batch:], 300, batch, graph=False); e.synchronize(); dt = time.perf_counter() - t
  print("no_dgemm=" + (env or "0"), f"{dt / 300 * 1e6:.1f} us/step  {batch / (dt / 300) / 1e6:.3f} M cells/s", flush=True)
  e.close()
