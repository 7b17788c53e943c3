#!/bin/bash
# same-box A/B of N builds of the library (_scratch/ab/lib_<name>.so, copied over sisua_amd/libsisua_hip.so in turn; the first name is restored at the end)
#   lib_ab3.sh "name1 name2 ..." [workloads...]     (default workloads: 8kly c5-shard)
NAMES=$1; shift; WL=${@:-8kly c5-shard}
cp sisua_amd/libsisua_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do
  for w in $WL; do
    for v in $NAMES; do
      cp _scratch/ab/lib_$v.so sisua_amd/libsisua_hip.so
      python3 bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline --no-c5-entry 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', '$v', round(d['ms_per_step']*1e3, 2), 'us', d.get('final_loss'), 'head', d['roofline']['avg_launch_us'], 'score', (d.get('scoring') or {}).get('marginal_llk_us'))"
    done
  done
done
cp /tmp/lib_keep.so sisua_amd/libsisua_hip.so
