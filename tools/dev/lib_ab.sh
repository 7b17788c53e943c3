#!/bin/bash
# same-box A/B of two builds of the library (tools/dev/libsisua_hip_old.so / _new.so, copied over sisua_amd/libsisua_hip.so in turn): bench.py --workload $1
W=${1:-c5-shard}; shift; EXTRA="$@"   # (further arguments go to bench.py, e.g. --storage u16)
for rep in 1 2 3; do
  for v in old new; do
    cp tools/dev/libsisua_hip_$v.so sisua_amd/libsisua_hip.so
    python3 bench.py --workload $W $EXTRA --steps 300 --warmup 30 --no-cpu-baseline --no-c5-entry 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step']*1e3, 2), 'us', d.get('final_loss'), d['roofline']['avg_launch_us'])"
  done
done
cp tools/dev/libsisua_hip_new.so sisua_amd/libsisua_hip.so
