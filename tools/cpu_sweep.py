import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from threadpoolctl import threadpool_limits
import bench
cfg, xt, batch, _ = bench.build_workload(0, 1, "8kly")
for n in (1, 8, 16, 32, 64, 256):
    with threadpool_limits(limits=n):
        r = bench.cpu_baseline(cfg, xt, batch, budget_s=4.0)
    print(n, r["value"], flush=True)
