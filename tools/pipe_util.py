#!/usr/bin/env python3
"""Per-kernel pipe occupancy from rocprofv3 --pmc passes (each pass a directory with counter_collection.csv) and one
--kernel-trace --stats pass of the same command: every collected SQ counter as an average per launch and, for the cycle counters,
as a share of (1024 SIMDs x kernel duration x 2.4 GHz).  SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count QUAD-cycles (the fused head's
SQ_ACTIVE_INST_VALU per wave equals its vector instruction count, and a wave64 vector instruction holds its SIMD for 4 cycles): their
shares are printed x 4; SQ_VALU_MFMA_BUSY_CYCLES counts cycles (18 MFMAs x 32 cycles x waves, exactly).   usage: pipe_util.py <stats dir> <pmc dir> [<pmc dir> ...]"""
import csv, glob, sys, collections
stats_dir, pmc_dirs = sys.argv[1], sys.argv[2:]
dur = {}
for r in csv.DictReader(open(glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True)[0])):
  dur[r["Name"]] = float(r["AverageNs"])
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for d in pmc_dirs:
  for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
    vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in vals for c in vals[k]})
CYC = {"SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_FLAT",
       "SQ_INST_CYCLES_SALU", "SQ_ACTIVE_INST_MISC"}
for k in sorted(vals, key=lambda k: -dur.get(k, 0) * len(next(iter(vals[k].values())))):
  if k not in dur or dur[k] < 3000:
    continue
  budget = 1024 * dur[k] * 1e-9 * 2.4e9
  print(f"{k[:90]}   avg {dur[k] / 1e3:.2f} us")
  for c in names:
    if c in vals[k]:
      v = sum(vals[k][c]) / len(vals[k][c])
      quad = 4.0 if (c.startswith("SQ_ACTIVE_INST") or c == "SQ_INST_CYCLES_SALU") else 1.0
      share = f"  = {100 * quad * v / budget:5.1f} % of SIMD-cycles" + (" (x 4: quad-cycles)" if quad > 1 else "") if c in CYC else ""
      if c == "SQ_WAVE_CYCLES":
        share = f"  = {4 * v / budget:5.2f} waves resident per SIMD on average (x 4: quad-cycles)"
      print(f"    {c:28s} {v:14.0f}{share}")
