"""Reads the stamp tables of a -DSMX_STAMPS build (tools/c2_stamps.sh) after 60 training steps of a bench.py workload and prints, per
launch, the phases of the stamped workgroup in shader cycles."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from sisua_amd import _hip
from sisua_amd.engine import Engine
workload = sys.argv[1] if len(sys.argv) > 1 else "8kly"
cfg, xt, batch, extra = bench.build_workload(0, 1, workload)
extra.pop("cell_id_base", None)
e = Engine(cfg, max_batch=batch); e.upload(xt, **extra)
order = bench.make_order(xt.shape[0], batch, 60)
for _ in range(100): e.eval_step(order[:batch])
e.train_steps(order, 60, batch, graph=False); e.synchronize()
lib = _hip.load()
NAMES = {("kernels", 0): ("bn_wide_fwd_kernel (encoder: column-major slabs summed, BatchNorm, ReLU, dropout; stamps of column 0's workgroup, from its first load)", ["entry", "", "", "slabs summed, partials in LDS", "", "column statistics", "normalise + stores issued"]),
         ("kernels", 1): ("bn_act_fwd_kernel<2,1> (latent sample + KL, first decoder product, BatchNorm)", ["entry", "latent + W tile in LDS (barrier)", "column of W -> registers", "dot products", "column sums", "column variances", "normalise + stores issued", "(from entry) the tile's loads issued", "(...) its first operands arrived", "(...) sample + KL computed, stores issued"]),
         ("kernels", 2): ("bn_wide_bwd_kernel (decoder: column-major slabs of d d summed, BatchNorm backward)", ["entry", "", "", "slabs summed, partials in LDS", "", "sum dy, sum dy xhat", "finish + stores issued"]),
         ("kernels", 3): ("bn_act_bwd_kernel<2,1> (d h product, encoder BatchNorm backward)", ["entry", "gradient tile + W rows in LDS (barrier)", "row of W -> registers", "dot products + mask + loads of out / xhat", "sum dy", "sum dy xhat", "finish + stores issued"]),
         ("headbwd", 4): ("out_head_bwd_kernel (workgroup 0: a dW tile)", ["entry", "products over the minibatch", "partial tiles summed, dW / db / sum of squares stored"])}
for unit in ("kernels", "headbwd"):
  fn = getattr(lib, "smx_dbg_stamps_" + unit)
  fn.restype = C.c_int
  buf = (C.c_longlong * 256)()
  assert fn(buf) == 0
  t = np.array(buf[:], dtype=np.int64).reshape(16, 16)
  for (u, slot), (name, phases) in NAMES.items():
    if u != unit: continue
    row = t[slot]
    ids = [i for i, p in enumerate(phases) if p and row[i] > 0]
    if len(ids) < 2:
      print(f"{name}: no stamps"); continue
    total = row[max(i for i in ids if i < 7)] - row[ids[0]]
    print(f"{name}: {total} cycles from entry to the last stamp")
    for a, b in zip(ids[:-1], ids[1:]):
      if b >= 7: print(f"    {phases[b]:58s} {row[b] - row[0]:7d} cycles after entry")
      else: print(f"    {phases[b]:58s} {row[b] - row[a]:7d} cycles")
e.close()
