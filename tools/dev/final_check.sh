#!/bin/bash
# the last call of a session: the GPU suite, smoke + both bench lines (tools/dev/final_bench.sh), the storage formats' line of tools/make_profiles.sh
mkdir -p gpurun_out/final
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/final/gpu_tests.log 2>&1; tail -2 gpurun_out/final/gpu_tests.log
bash tools/dev/final_bench.sh
for s in f32 u16 csr; do python3 bench.py --storage $s --steps 300 --warmup 30 --no-cpu-baseline --no-c5-entry 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$s', d['ms_per_step'], d['final_loss'])"; done > gpurun_out/final/storage_formats.txt
cat gpurun_out/final/storage_formats.txt
