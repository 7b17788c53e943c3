# development: odd minibatch sizes through the slab-summing BatchNorm launches -- default against knob no_bn_wide, bit for bit; eval and forward passes too
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import sisua_oracle as so
from sisua_amd import _hip
from sisua_amd.engine import Engine
from tests.util import make_pair, synth_counts
bad = 0
for G, units in ((700, 64), (1998, 128), (4200, 128)):
  spec, cfg = make_pair(model="vae", n_genes=G, likelihood="zinb", enc_units=(units,), dec_units=(units,), latent_dim=16)
  x = synth_counts(300, G, sparsity=0.92, seed=G, max_count=300)
  for B in (1, 2, 31, 33, 64, 65, 127, 128):
    rows = np.random.default_rng(B).permutation(300)[:B].astype(np.int32)
    out = []
    for off in (1, 0):
      _hip.set_tuning("no_bn_wide", off)
      e = Engine(cfg, max_batch=128, init=False)
      e.set_params(so.init_params(spec))
      e.upload(x, cell_id_base=9, storage="u16")
      m1 = e.train_step(rows); m2 = e.train_step(rows, graph=True); ev = e.eval_step(rows)
      f = e.forward(row_ids=rows)["x_params"]
      out.append((m1["loss"], m2["loss"], ev["loss"], f, e.get_params(0)))
      e.close()
    same = out[0][0] == out[1][0] and out[0][1] == out[1][1] and out[0][2] == out[1][2] and np.array_equal(out[0][3], out[1][3]) and all(np.array_equal(out[0][4][k], out[1][4][k]) for k in out[0][4])
    bad += not same
    print(G, units, B, "same" if same else "DIFFERENT", out[1][0], np.isfinite(out[1][0]))
_hip.set_tuning("no_bn_wide", 0)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
