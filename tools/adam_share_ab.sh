#!/bin/bash
# c5-shard step time over the shares of the heads' optimiser chunks that ride with the latent head's backward product
# (knob adam_wide_share_b) and with the encoder's BatchNorm-backward launch (knob adam_wide_share); 0 = the optimiser launch keeps them
for rep in 1 2; do for sh in ${SHARES:-"0 0" "0 0.3" "0.1 0.3" "0.15 0.3" "0.2 0.3" "0.15 0.4"}; do
  set -- $sh
  export SMX_TUNING="adam_wide_share_b=$1,adam_wide_share=$2"
  python3 bench.py --workload c5-shard --steps 300 --warmup 30 --no-cpu-baseline --no-c5-entry 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_us']; print('shares $1 / $2:', round(1e3*d['ms_per_step'],1), 'us/step  bn_bwd', k['bn_bwd'], ' adam', k['adam'], d['final_loss'])"
done; done
