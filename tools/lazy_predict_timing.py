"""predict() consumed through statistics: the eager path (parameter planes to the host, NumPy statistics) against the lazy handle
(distributions.LazyCountOutput: the statistic computed on the device, smx_predict_stat).  VERDICT r03 item 6.
   usage: python tools/lazy_predict_timing.py   (8kly-shaped synthetic counts, VAE zinb 128 / 32, on the GPU)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sisua_amd import models as M  # noqa: E402
from sisua_amd.data import SingleCellOMIC  # noqa: E402

cfg, xt, batch, _ = bench.build_workload(0, 1, "8kly")
sco = SingleCellOMIC(xt, name="8kly")
m = M.VAE(outputs=M.RVmeta(xt.shape[1], "zinb", name="transcriptomic"), latents=M.RVmeta(32, "diag", True, "Latents"),
          encoder=M.NetConf([128], batchnorm=True, dropout=0.1), decoder=M.NetConf([128], batchnorm=True, dropout=0.1))
m.fit(sco.create_dataset(batch_size=128, drop_remainder=True), metadata=sco, epochs=2)
test = xt[:940]
tsco = SingleCellOMIC(test, name="test")


def best(fn, n=5):
  fn()
  ts = []
  for _ in range(n):
    t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
  return min(ts)


print("## Posterior's defaults (posterior.py:114-115: batch 8, 10 draws) on 940 cells x 1998 genes: predict + the imputed mean over the draws")
def eager():
  pX, _ = m.predict(test, sample_shape=10, batch_size=8, verbose=False)
  return pX.distribution.count_distribution.mean().mean(0)
def lazy():
  pX, _ = m.predict(tsco, sample_shape=10, batch_size=8, verbose=False)
  return pX.distribution.count_distribution.mean_over_samples()
te, tl = best(eager), best(lazy)
assert np.allclose(eager(), lazy(), rtol=2e-5, atol=1e-6)
print(f"eager (planes to the host, NumPy mean)          {1e3 * te:8.2f} ms  {940 / te / 1e3:8.1f} k cells/s")
print(f"lazy  (mean over the draws on the device)       {1e3 * tl:8.2f} ms  {940 / tl / 1e3:8.1f} k cells/s")
def lazy_llk():
  pX, _ = m.predict(tsco, sample_shape=10, batch_size=8, verbose=False)
  return pX.log_prob()
tll = best(lazy_llk)
print(f"lazy  (log_prob of the counts, [10, 940])       {1e3 * tll:8.2f} ms  {940 / tll / 1e3:8.1f} k cells/s")
for B in (32, 128):
  def lz():
    pX, _ = m.predict(tsco, sample_shape=10, batch_size=B, verbose=False)
    return pX.distribution.count_distribution.mean_over_samples()
  t = best(lz)
  print(f"lazy, batch {B:3d} x 10 draws                        {1e3 * t:8.2f} ms  {940 / t / 1e3:8.1f} k cells/s")

print("## predict of means, one draw, batch 128, every training cell replicated to 54 096 cells x 1998 genes")
big = np.tile(xt, (16, 1))
bsco = SingleCellOMIC(big, name="big")
def eager_mean():
  pX, _ = m.predict(big, batch_size=128, verbose=False)
  return pX.mean()
pL, _ = m.predict(bsco, batch_size=128, verbose=False)
buf = np.empty((big.shape[0], big.shape[1]), np.float32)
te = best(eager_mean, 2)
tl = best(lambda: pL.mean(), 3)
tlo = best(lambda: pL.mean(out=buf), 3)
tp = best(lambda: m.predict(bsco, batch_size=128, verbose=False), 3)
print(f"eager predict + .mean()                          {1e3 * te:8.1f} ms  {big.shape[0] / te / 1e6:6.2f} M cells/s")
print(f"lazy predict (latents only leave the device)     {1e3 * tp:8.1f} ms  {big.shape[0] / tp / 1e6:6.2f} M cells/s")
print(f"lazy .mean() into a fresh array                  {1e3 * tl:8.1f} ms  {big.shape[0] / tl / 1e6:6.2f} M cells/s")
print(f"lazy .mean(out=reused array)                     {1e3 * tlo:8.1f} ms  {big.shape[0] / tlo / 1e6:6.2f} M cells/s")
