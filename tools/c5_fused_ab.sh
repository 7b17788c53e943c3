#!/bin/bash
# development A/B at the C5 width (bench.py --workload c5-shard): the one-launch head against the separate launches it replaced
# (round 5's A/B against round 4's whole-tile kernel: profiles/r05_head_fused_experiments.txt)
set -u
for v in "" "no_head_fused=1" ""; do
  if [ -n "$v" ]; then export SMX_TUNING="$v"; else unset SMX_TUNING; fi
  python bench.py --workload c5-shard --steps 200 --warmup 20 --no-cpu-baseline --no-c5-entry 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-16s' % ('$v' or 'default'), round(d['ms_per_step']*1e3, 1), 'us', d.get('final_loss'))"
done
