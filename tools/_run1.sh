python -m pytest tests/test_gpu_step.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/s3_t.log 2>&1; tail -3 gpurun_out/s3_t.log
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --no-c5-entry 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3, 2), 'us', round(d['ms_per_step_300']*1e3,2), d.get('final_loss'))"
done
bash tools/c2_stamps.sh 0 8kly 2>&1 | grep -A10 "bn_act_fwd_kernel<2,1>\|bn_act_bwd_kernel<2,1>" | head -30
