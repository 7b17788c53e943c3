#!/bin/bash
# development A/B at the C5 width (bench.py --workload c5-shard): the heads' optimiser update as a background sweep on a second stream
# (flag head_sweep; SMX_TUNING no_head_sweep = riders + the optimiser launch, adam_sweep_wgs = number of persistent workgroups)
set -u
for v in "no_head_sweep=1" "" "no_head_sweep=1" ""; do
  export SMX_TUNING="$v"
  python bench.py --workload c5-shard --steps 200 --warmup 20 --no-cpu-baseline --no-c5-entry 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s' % ('$v' or 'default'), round(d['ms_per_step']*1e3, 1), 'us', d.get('final_loss'))"
done
