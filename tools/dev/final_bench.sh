mkdir -p gpurun_out/final
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1 &&
python3 bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_driver_settings.json 2> gpurun_out/final/bench_driver_settings.err &&
python3 bench.py --steps 300 --warmup 30 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -2 gpurun_out/final/smoke.log; python3 - <<'PY'
import json
for f in ("bench_driver_settings", "bench"):
  d = json.load(open(f"gpurun_out/final/{f}.json"))
  print(f, d["value"], d["ms_per_step"], d.get("value_300"), d["final_loss"], d["roofline"]["frac"], d["roofline"].get("c5_shard_step_us"), d["scoring"], d["cpu_baseline"])
PY
