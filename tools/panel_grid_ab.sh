for g in 512 100000 256 768; do
  export SMX_TUNING="panel_grid=$g"
  python bench.py --workload c5-shard --steps 300 --warmup 30 --no-cpu-baseline --no-c5-entry 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('grid cap $g:', d['ms_per_step'], d['kernel_us'])"
done
