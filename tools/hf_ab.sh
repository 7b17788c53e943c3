#!/bin/bash
# development: the one-launch output head by itself on one box -- parity of the half-tile kernel, then its launch time against round 4's
# whole-tile kernel (knob hf_v1) and against the table-reload form (knob hf_reload) at 128 x 20 000.   usage: hf_ab.sh [likelihoods ...]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
python3 tools/headfused_try.py --reps 50
for lk in ${@:-zinb nb zinbd nbd}; do
  python3 tools/headfused_try.py --time-only $lk --reps 200
  SMX_TUNING=hf_reload python3 tools/headfused_try.py --time-only $lk --reps 200
  SMX_TUNING=hf_v1 python3 tools/headfused_try.py --time-only $lk --reps 200
done
} 2>&1 | tee gpurun_out/hf_ab.txt
