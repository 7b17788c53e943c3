for v in "" "SMX_TUNING=split_deep=512" "SMX_TUNING=split_deep=256"; do
  env $v python3 bench.py --workload c5-shard --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', 1e3*d['ms_per_step'], d['kernel_us'])"
done
