#!/bin/bash
# c5-shard step time over the big-K kernel's pipeline depth (knob bigk_stages = stage buffers in LDS: 4 x 32 KB fills a CU, 2 x 32 KB leaves room)
for rep in 1 2; do for sg in 4 3 2; do
  export SMX_TUNING="bigk_stages=$sg"
  python3 bench.py --workload c5-shard --steps 300 --warmup 30 --no-cpu-baseline --no-c5-entry 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_us']; print('stages $sg:', round(1e3*d['ms_per_step'],1), 'us/step  enc_fwd', k['gemm_enc_fwd'], ' out_bwd', k['gemm_out_bwd'], d['final_loss'])"
done; done
