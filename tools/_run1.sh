python - <<'PY'
import json, subprocess
for wl, extra in (("8kly", []), ("c5-shard", ["--steps", "200", "--warmup", "20", "--no-c5-entry"])):
  for rep in range(2):
    out = subprocess.run(["python", "bench.py", "--no-cpu-baseline", "--workload", wl] + extra, capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    print(wl, round(d["ms_per_step"] * 1e3, 2), round(d.get("ms_per_step_300", 0) * 1e3, 2), d.get("final_loss"), flush=True)
PY
