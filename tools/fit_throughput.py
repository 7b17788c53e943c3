"""End-to-end SingleCellModel.fit throughput (host epoch preparation included) vs the raw step rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sisua_amd import data
from sisua_amd.models import VAE, NetConf, RVmeta

x, _ = data.synthetic_8kly(seed=8)
sco = data.SingleCellOMIC(x, name="8kly")
train, test = sco.split(0.8)
tr, va = train.split(0.9)
tr.corrupt(dropout_rate=0.2, retain_rate=0.2, inplace=True)
model = VAE(outputs=RVmeta(x.shape[1], "zinb", True, "transcriptomic"), latents=RVmeta(32, "diag", True, "Latents"),
            encoder=NetConf([128], batchnorm=True, dropout=0.1), decoder=NetConf([128], batchnorm=True, dropout=0.1))
ds_tr = tr.create_dataset(["transcriptomic"], labels_percent=0.1, batch_size=128, drop_remainder=True, shuffle=1000)
ds_va = va.create_dataset(["transcriptomic"], labels_percent=0.1, batch_size=128, drop_remainder=True, shuffle=1000)
model.fit(ds_tr, valid=ds_va, metadata=sco, epochs=2, learning_rate=1e-3, clipnorm=100, valid_freq=500)   # warm-up
for epochs in (50, 200):
  t = time.perf_counter()
  model.fit(ds_tr, valid=ds_va, metadata=sco, epochs=epochs, learning_rate=1e-3, clipnorm=100, valid_freq=500, earlystop_patience=10 ** 6)
  dt = time.perf_counter() - t
  steps = epochs * (tr.n_obs // 128)
  print(f"fit: {epochs} epochs = {steps} steps in {dt:.3f} s -> {steps * 128 / dt:.0f} cells/s ({dt / steps * 1e6:.1f} us/step)")

xs = test.numpy() if hasattr(test, "numpy") else x[:900]
for bs, S in ((32, ()), (128, ()), (8, 10), (128, 10)):
  t = time.perf_counter()
  X, Z = model.predict(xs, sample_shape=S, batch_size=bs, verbose=False)
  dt = time.perf_counter() - t
  print(f"predict: {xs.shape[0]} cells, batch {bs}, sample_shape {S}: {dt * 1e3:.1f} ms -> {xs.shape[0] / dt:.0f} cells/s")
