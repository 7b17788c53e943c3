#!/usr/bin/env python3
"""Step time of a VAE zinb [128] / [128] at several panel widths with the fused output head (smx_headfused.hip) + the heads' optimiser update
as a background sweep (flag head_sweep), with the fused head alone, and with the separate launches (flag head_fused = 0): from which width
each form pays.  usage: head_fused_width_ab.py [genes ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sisua_amd.config import ModelConfig
from sisua_amd.engine import Engine
from tests.util import synth_counts

for G in [int(a) for a in sys.argv[1:]] or [4096, 6000, 8000, 12000, 20000]:
  cfg = ModelConfig(model="vae", n_genes=G, likelihood="zinb", enc_units=(128,), dec_units=(128,), latent_dim=32)
  x = synth_counts(1024, G, sparsity=0.92, seed=1)
  out = []
  for fused, sweep in ((1, 1), (1, 0), (0, 0)):
    e = Engine(cfg, max_batch=128)
    e.set_flag("head_fused", bool(fused))
    e.set_flag("head_sweep", bool(sweep))
    e.upload(x, storage="u16")
    order = np.concatenate([np.random.default_rng(s).permutation(1024)[:128] for s in range(330)]).astype(np.int32)
    e.train_steps(order[: 30 * 128], 30, 128, graph=False)
    e.synchronize()
    t = time.perf_counter()
    e.train_steps(order[30 * 128:], 300, 128, graph=False)
    e.synchronize()
    out.append(1e6 * (time.perf_counter() - t) / 300)
    e.close()
  print(f"genes {G:6d}: fused + background sweep {out[0]:7.1f} us per step, fused {out[1]:7.1f}, separate launches {out[2]:7.1f}", flush=True)
