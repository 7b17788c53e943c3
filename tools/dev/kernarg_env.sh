#!/bin/bash
# where the runtime puts the launches' argument segments: the C2 step (300 steps, us per step) under HIP_FORCE_DEV_KERNARG unset / 0 / 1, twice
mkdir -p gpurun_out
: > gpurun_out/kernarg_env.log
for rep in 1 2; do
  for v in unset 0 1; do
    if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
    python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-c5-entry 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('HIP_FORCE_DEV_KERNARG=$v', round(1e3*d['ms_per_step'],2), 'us per step', d['final_loss'], d['kernel_us'])" >> gpurun_out/kernarg_env.log
  done
done
cat gpurun_out/kernarg_env.log
