for w in 8kly c5-shard eccly-sisua; do for v in 8 16 4; do
  SMX_HEAD_WAVES=$v python3 bench.py --workload $w --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w waves $v', 1e3*d['ms_per_step'], 'fused', r['fused_kernel_us'], 'product', r['product_only_us'], 'frac', r['frac'])"
done; done
