#!/usr/bin/env python3
"""One smx_marginal_llk call's kernels from a rocprofv3 --kernel-trace CSV directory of bench.py: the launches between two score_head_kernel launches
(start, duration, gap to the launch before)."""
import csv, glob, sys
d = sys.argv[1]
trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
sh = [i for i, r in enumerate(rows) if ("score_head_kernel" in r["Kernel_Name"] or "score_walk_kernel" in r["Kernel_Name"])]
a, b = sh[len(sh) // 2], sh[len(sh) // 2 + 1]
t0 = int(rows[a + 1]["Start_Timestamp"]); prev = int(rows[a]["End_Timestamp"])
print(f"(previous score_head ended {(t0 - prev) / 1e3:.1f} us before the next call's first launch)")
for r in rows[a + 1:b + 1]:
  s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
  print(f"  {r['Kernel_Name'][:70]:70s} t={(s - t0) / 1e3:7.2f} dur={(e - s) / 1e3:7.2f}us gap={(s - prev) / 1e3:6.2f} grid={r['Grid_Size_X']}")
  prev = e
print(f"  first launch to the end of score_head: {(prev - t0) / 1e3:.1f} us")
