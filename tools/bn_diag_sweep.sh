for d in 0 1 2 4 16 32 64 112; do
  SMX_BN_DIAG=$d python3 bench.py --workload eccly-sisua --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('diag $d', 1e3*d['ms_per_step'], d['kernel_us'])"
done
