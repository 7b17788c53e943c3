"""Where SingleCellModel.predict spends its time (cProfile), 940 cells of the 8kly-shaped workload."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sisua_amd import data
from sisua_amd.models import VAE, NetConf, RVmeta

x, _ = data.synthetic_8kly(seed=8)
sco = data.SingleCellOMIC(x, name="8kly")
train, test = sco.split(0.8)
model = VAE(outputs=RVmeta(x.shape[1], "zinb", True, "transcriptomic"), latents=RVmeta(32, "diag", True, "Latents"),
            encoder=NetConf([128], batchnorm=True, dropout=0.1), decoder=NetConf([128], batchnorm=True, dropout=0.1))
ds = train.create_dataset(["transcriptomic"], labels_percent=0.1, batch_size=128, drop_remainder=True, shuffle=1000)
model.fit(ds, metadata=sco, epochs=2, learning_rate=1e-3, clipnorm=100)
xs = test.numpy() if hasattr(test, "numpy") else x[:940]
for bs, S in ((128, ()), (32, ()), (128, 10), (8, 10)):
  model.predict(xs, sample_shape=S, batch_size=bs, verbose=False)
  dt = 0.0
  for _ in range(3):
    t = time.perf_counter()
    X, Z = model.predict(xs, sample_shape=S, batch_size=bs, verbose=False)
    dt += (time.perf_counter() - t) / 3
    del X, Z   # (outside the timed region: returning 227 MB of touched pages to the OS takes as long as the call)
  print(f"predict {xs.shape[0]} cells batch {bs} sample_shape {S}: {dt * 1e3:.2f} ms -> {xs.shape[0] / dt:.0f} cells/s")
if "--profile" in sys.argv:
  S_prof = 10 if "--draws" in sys.argv else ()
  pr = cProfile.Profile(); pr.enable()
  for _ in range(5):
    model.predict(xs, sample_shape=S_prof, batch_size=128, verbose=False)
  pr.disable()
  s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
t = time.perf_counter(); Z = model.encode(xs[:128]); print(f"encode 128 cells: {(time.perf_counter() - t) * 1e3:.2f} ms")
