#!/bin/bash
# how the host waits for the device: the fixed cost of a smx_train_steps call (tools/dev/train_tail.py) and the 20-step bench line with the runtime's
# interrupt-driven signal waits (default) and with polling (HSA_ENABLE_INTERRUPT=0)
mkdir -p gpurun_out
: > gpurun_out/wait_mode.log
for v in default 0; do
  if [ $v = default ]; then unset HSA_ENABLE_INTERRUPT; else export HSA_ENABLE_INTERRUPT=$v; fi
  echo "== HSA_ENABLE_INTERRUPT=$v" >> gpurun_out/wait_mode.log
  KS="1 20 100" REPS=15 timeout -k 10 200 python3 tools/dev/train_tail.py 2>&1 | grep -v "calls in order" >> gpurun_out/wait_mode.log
  for r in 1 2; do
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c5-entry 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench 20 steps:', round(1e3*d['ms_per_step'],2), 'us per step;', d['final_loss'], 'scoring', d['scoring']['marginal_llk_us'])" >> gpurun_out/wait_mode.log
  done
done
cat gpurun_out/wait_mode.log
