#!/bin/bash
# every single-GPU workload's step time on one box (A/B runs of a kernel change): us per step and the final loss
for w in 8kly 8kly-scvi eccly-sisua 8kly-2layer cortex-base c5-shard; do
  python3 bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline --no-c5-entry 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w', round(1e3 * d['ms_per_step'], 1), 'us', d['final_loss'], d['kernel_us'])"
done
