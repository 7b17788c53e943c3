python -m pytest tests -x -q -m gpu > gpurun_out/s3_t.log 2>&1; tail -3 gpurun_out/s3_t.log
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --no-c5-entry 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3, 2), 'us', round(d['ms_per_step_300']*1e3,2), d.get('final_loss'))"
done
