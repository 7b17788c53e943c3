import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from sisua_amd.config import ModelConfig
from sisua_amd.engine import Engine
_, xt, batch, _ = bench.build_workload(0, 1, "8kly")
cfg = ModelConfig(n_genes=xt.shape[1], enc_units=(128,), dec_units=(128,), latent_dim=32, dropout_enc=0.1, dropout_dec=0.1, seed=8, model="scale", likelihood="zinb", n_components=10, covariance="tril")
e = Engine(cfg, max_batch=batch); e.upload(xt)
order = bench.make_order(xt.shape[0], batch, 60)
e.train_steps(order, 60, batch, graph=False); e.synchronize(); e.close()
