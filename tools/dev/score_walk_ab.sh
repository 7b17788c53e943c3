#!/bin/bash
# tools/dev/score_walk_ab.py on the GPU box, output under gpurun_out/
mkdir -p gpurun_out
timeout -k 10 400 python3 tools/dev/score_walk_ab.py > gpurun_out/score_walk_ab.log 2>&1
echo "rc $?" >> gpurun_out/score_walk_ab.log
tail -20 gpurun_out/score_walk_ab.log
