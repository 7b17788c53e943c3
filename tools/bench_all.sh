#!/bin/bash
# every single-GPU workload of bench.py, one JSON per workload under gpurun_out/<tag>/ (no CPU baseline)
TAG=${1:-all}
O=gpurun_out/$TAG
mkdir -p $O
for w in 8kly c5-shard 8kly-scvi eccly-sisua 8kly-2layer cortex-base; do
  python3 bench.py --workload $w --no-cpu-baseline --steps 300 --warmup 30 > $O/bench_$w.json 2> $O/bench_$w.err
  python3 - "$O/bench_$w.json" "$w" <<'PY'
import json, sys
try:
  d = json.load(open(sys.argv[1]))
  print(f"{sys.argv[2]:12s} {d['value']:10.0f} cells/s  {1e3 * d['ms_per_step']:7.1f} us/step  frac {d['roofline']['frac']:.3f}  {d['kernel_us']}")
except Exception as e:
  print(sys.argv[2], "FAILED", e)
PY
done
