#!/usr/bin/env python3
"""Reduce the PMC passes of tools/pmc_pass.sh to per-launch HBM bytes of the loss kernel.
Counter unit and the gfx950 FETCH_SIZE correction are CALIBRATED on copy kernels of known size
(64 MiB read + 64 MiB written, 16 B/lane and 4 B/lane) run in the same passes."""
import csv
import glob
import json
import sys

out = sys.argv[1]


def counter_rows(d):
  f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
  rows = []
  for p in f:
    rows += list(csv.DictReader(open(p)))
  return rows


res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
  cal = counter_rows(f"{out}/calib_{c}")
  byk = {}
  for r in cal:
    if r["Counter_Name"] == c:
      byk.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
  k16 = [v for k, v in byk.items() if "k_copy(" in k or k.startswith("k_copy(")]
  k4 = [v for k, v in byk.items() if "k_copy1" in k]
  last16 = k16[0][-1] if k16 else None
  last4 = k4[0][-1] if k4 else None
  rows = counter_rows(f"{out}/bench_{c}")
  vals = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == c and "count_loss_kernel" in r["Kernel_Name"]]
  res[c] = dict(calib_16B_per_lane=last16, calib_4B_per_lane=last4, loss_kernel_mean=sum(vals) / max(len(vals), 1), n=len(vals))
known = 64 * 1024 * 1024
summary = {"raw": res}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
  cal4 = res[c]["calib_4B_per_lane"]
  if cal4:
    bytes_per_unit = known / cal4   # calibrated on the loss kernel's own access width (4 B/lane)
    summary[c + "_bytes_per_unit_4B"] = bytes_per_unit
    summary[c + "_loss_bytes"] = res[c]["loss_kernel_mean"] * bytes_per_unit
  cal16 = res[c]["calib_16B_per_lane"]
  if cal16:
    summary[c + "_bytes_per_unit_16B"] = known / cal16
if "FETCH_SIZE_loss_bytes" in summary and "WRITE_SIZE_loss_bytes" in summary:
  summary["loss_kernel_hbm_bytes_per_launch"] = summary["FETCH_SIZE_loss_bytes"] + summary["WRITE_SIZE_loss_bytes"]
print(json.dumps(summary, indent=1))
json.dump(summary, open(out + "/pmc_summary.json", "w"), indent=1)
