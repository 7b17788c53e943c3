"""Thread-count sweep of the CPU baseline (C/OpenMP port) on the GPU box's host."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
cfg, xt, batch, extra = bench.build_workload(0, 1, "8kly")
for n in (1, 8, 16, 32, 64, 128, 256):
  if n > (os.cpu_count() or 1):
    break
  r = bench.cpu_baseline(cfg, xt, batch, budget_s=3.0, threads=n)
  print(n, r["value"], flush=True)
