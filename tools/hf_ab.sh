#!/bin/bash
# development: the one-launch output head by itself on one box -- parity at a few shapes, then its launch time at 128 x 20 000 for the given
# likelihoods.  (Round 5's A/B against round 4's whole-tile kernel -- removed since -- is in profiles/r05_head_fused_experiments.txt.)
# usage: hf_ab.sh [likelihoods ...]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
python3 tools/headfused_try.py --reps 50
for lk in ${@:-zinb nb zinbd nbd}; do
  python3 tools/headfused_try.py --time-only $lk --reps 200
done
} 2>&1 | tee gpurun_out/hf_ab.txt
