#!/bin/bash
# development A/B at the C5 width (bench.py --workload c5-shard): the heads' optimiser update as a background sweep on a second stream
# (SMX_TUNING adam_sweep = number of persistent workgroups; 0 = off: riders + the optimiser launch)
set -u
for v in "adam_sweep=0" "head_lazy=1" "adam_sweep=128" "head_lazy=1" "adam_sweep=0"; do
  export SMX_TUNING="$v"
  python bench.py --workload c5-shard --steps 200 --warmup 20 --no-cpu-baseline --no-c5-entry 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s' % ('$v' or 'default'), round(d['ms_per_step']*1e3, 1), 'us', d.get('final_loss'))"
done
