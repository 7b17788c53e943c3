#!/bin/bash
# c5-shard step time over the share of the heads' optimiser chunks that rides with the encoder's BatchNorm-backward launch
# (SMX_ADAM_WIDE_SHARE; 0 = the optimiser launch keeps every chunk)
for rep in 1 2; do for sh in ${SHARES:-0 0.1 0.2 0.3 0.5}; do
  export SMX_ADAM_WIDE_SHARE=$sh
  python3 bench.py --workload c5-shard --steps 300 --warmup 30 --no-cpu-baseline --no-c5-entry 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_us']; print('share $sh:', round(1e3*d['ms_per_step'],1), 'us/step  bn_bwd', k['bn_bwd'], ' adam', k['adam'], d['final_loss'])"
done; done
