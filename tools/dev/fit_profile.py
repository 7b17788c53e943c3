"""cProfile of one SingleCellModel.fit call of 50 epochs at the 8kly shape (after a warm-up call): where the call's fixed ~40 ms go."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
model.fit(ds_tr, valid=ds_va, metadata=sco, epochs=2, learning_rate=1e-3, clipnorm=100, valid_freq=500)
pr = cProfile.Profile()
t = time.perf_counter()
pr.enable()
model.fit(ds_tr, valid=ds_va, metadata=sco, epochs=50, learning_rate=1e-3, clipnorm=100, valid_freq=500, earlystop_patience=10 ** 6)
pr.disable()
print(f"fit(50 epochs): {(time.perf_counter() - t) * 1e3:.1f} ms")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
