#!/bin/bash
# quick A/B of loss-kernel vector width on the bench workload
for v in 1 2 4; do
  echo "SMX_LOSS_VEC=$v"; SMX_LOSS_VEC=$v timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['kernel_us'])"
done
