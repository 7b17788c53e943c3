set -u
for f in 1 0; do
  if [ $f = 0 ]; then export SMX_TUNING="no_head_fused=1"; else unset SMX_TUNING; fi
  python bench.py --workload c5-shard --steps 200 --warmup 20 --no-cpu-baseline --no-c5-entry 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused' if $f else 'old  ', d['ms_per_step']*1e3, 'us', d.get('final_loss'))"
done
