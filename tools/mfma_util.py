#!/usr/bin/env python3
"""MFMA utilisation per kernel from two rocprofv3 outputs of the same command: a --pmc SQ_VALU_MFMA_BUSY_CYCLES pass
(counter_collection.csv) and a --kernel-trace --stats pass (kernel_stats.csv).  utilisation = busy cycles /
(1024 SIMDs x kernel duration x clock); clock taken as 2.4 GHz."""
import csv, glob, sys, collections
pmc_dir, stats_dir = sys.argv[1], sys.argv[2]
busy = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(pmc_dir + "/**/*counter_collection.csv", recursive=True)[0])):
  if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
    busy[r["Kernel_Name"]].append(float(r["Counter_Value"]))
dur = {}
for r in csv.DictReader(open(glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True)[0])):
  dur[r["Name"]] = float(r["AverageNs"])
print(f"{'kernel':72s} {'avg us':>8s} {'MFMA busy cyc':>14s} {'util of 1024 SIMDs':>19s}")
for k, v in sorted(busy.items(), key=lambda kv: -sum(kv[1])):
  b = sum(v) / len(v)
  if b <= 0 or k not in dur:
    continue
  util = b / (1024 * dur[k] * 1e-9 * 2.4e9)
  print(f"{k[:72]:72s} {dur[k] / 1e3:8.2f} {b:14.0f} {100 * util:18.1f}%")
