# development: one training step with and without the wide-slab BatchNorm launches -- which gradients differ, by how much
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import sisua_oracle as so
from sisua_amd import _hip
from sisua_amd.engine import Engine
from tests.util import make_pair, synth_counts
G, B, lk = 4100, 100, "zinb"
spec, cfg = make_pair(model="vae", n_genes=G, likelihood=lk, enc_units=(128,), dec_units=(128,), latent_dim=16)
x = synth_counts(512, G, sparsity=0.92, seed=G + 1, max_count=700)
rows = np.random.default_rng(1).permutation(512)[:B].astype(np.int32)
out = []
for off in (1, 0):
  _hip.set_tuning("no_bn_wide", off)
  e = Engine(cfg, max_batch=128, init=False)
  e.set_params(so.init_params(spec))
  e.upload(x, cell_id_base=9, storage="u16")
  m = e.train_step(rows, graph=("graph" in sys.argv))
  out.append((m, e.get_params(1), e.get_params(0)))
  e.close()
print(out[0][0]["loss"], out[1][0]["loss"])
for which in (1, 2):
  for k in out[0][which]:
    a, b = out[0][which][k], out[1][which][k]
    if not np.array_equal(a, b):
      d = np.abs(a - b)
      print(("grad " if which == 1 else "param"), k, "differs in", int((a != b).sum()), "of", a.size, "max abs", d.max(), "max |a|", np.abs(a).max())
