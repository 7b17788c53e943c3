# development: where an eager predict() call of a numpy matrix spends its time (cProfile, by cumulative time)
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sisua_amd import data
from sisua_amd.models import VAE, NetConf, RVmeta
x, _ = data.synthetic_8kly(seed=8)
sco = data.SingleCellOMIC(x, name="8kly")
train, test = sco.split(0.8)
model = VAE(outputs=RVmeta(x.shape[1], "zinb", True, "transcriptomic"), latents=RVmeta(32, "diag", True, "Latents"),
            encoder=NetConf([128], batchnorm=True, dropout=0.1), decoder=NetConf([128], batchnorm=True, dropout=0.1))
ds = train.create_dataset(["transcriptomic"], labels_percent=0.1, batch_size=128, drop_remainder=True, shuffle=1000)
model.fit(ds, metadata=sco, epochs=1, learning_rate=1e-3, clipnorm=100)
xs = test.numpy()
for bs in (32, 128):
  model.predict(xs, batch_size=bs, verbose=False)
  t = time.perf_counter(); model.predict(xs, batch_size=bs, verbose=False); print("batch", bs, "ms", 1e3 * (time.perf_counter() - t))
pr = cProfile.Profile(); pr.enable()
model.predict(xs, batch_size=128, verbose=False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
