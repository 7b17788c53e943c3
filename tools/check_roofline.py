#!/usr/bin/env python3
"""Recomputes every roofline fraction of a bench line from a rocprofv3 --kernel-trace --stats summary of the SAME command and
fails if they disagree by more than 10 % (the figure must follow from the profile).

  python tools/check_roofline.py [bench.json] [kernel_stats.csv] [tag] [--info]
  --info: the bench line comes from a run WITHOUT the profiler -- the profiler serialises dispatches, so a few-microsecond kernel launched back
  to back (as bench.py times it) measures ~10 % longer under it (6.2 vs 6.9 us for the C2 fused head); differences are printed, not judged
  defaults: profiles/r03_bench.json  profiles/r03_rocprofv3_kernel_stats.csv  (tag = the bench line's workload)

For each entry of roofline.entries whose name ends in @<tag>: frac_profile = bytes_per_launch / AverageNs / 8 TB/s, with
AverageNs the calls-weighted mean over the profile rows whose kernel name matches the entry's kernel_regex.  The profile
must come from a run without the entries of another width (bench.py --no-c5-entry), since rocprofv3 averages by name.
The exit status follows the DOMINANT kernel (roofline.kernel, the headline frac); the other entries are printed with their
profile-derived fraction beside the bench's: a 3-5 us kernel's begin-to-end duration under rocprofv3 includes the ramp and
drain of an isolated dispatch (4.6 us for the standalone likelihood at 128 x 1998) that back-to-back launches overlap
(3.5 us per launch), so for such a kernel BOTH numbers are quoted and the lower one is the one to cite."""
import csv
import json
import re
import sys

PEAK = 8000.0   # GB/s, MI355X_MICROARCH.md
TOL = 0.10


def main():
  info = "--info" in sys.argv
  if info:
    sys.argv.remove("--info")
  bench = sys.argv[1] if len(sys.argv) > 1 else "profiles/r03_bench.json"
  stats = sys.argv[2] if len(sys.argv) > 2 else "profiles/r03_rocprofv3_kernel_stats.csv"
  line = [l for l in open(bench).read().splitlines() if l.strip().startswith("{")][-1]
  d = json.loads(line)
  tag = sys.argv[3] if len(sys.argv) > 3 else d["roofline"]["kernel"].split("@")[-1]
  rows = list(csv.DictReader(open(stats)))
  bad = 0
  for e in d["roofline"]["entries"]:
    if not e["name"].endswith("@" + tag):
      continue
    pat = re.compile(e["kernel_regex"])
    hit = [r for r in rows if pat.search(r["Name"])]
    if not hit:
      print(f"{e['name']}: no kernel in the profile matches /{e['kernel_regex']}/")
      bad += 1
      continue
    calls = sum(int(r["Calls"]) for r in hit)
    avg_us = sum(int(r["Calls"]) * float(r["AverageNs"]) for r in hit) / calls / 1e3
    frac_p = e["bytes_per_launch"] / (avg_us * 1e-6) / 1e9 / PEAK
    rel = abs(frac_p - e["frac"]) / max(frac_p, 1e-9)
    ok = rel <= TOL
    bad += (not ok) and e["name"] == d["roofline"]["kernel"] and not info
    print(f"{e['name']:44s} bench {e['avg_launch_us']:7.2f} us frac {e['frac']:.4f} | rocprofv3 {avg_us:7.2f} us ({calls} calls) frac {frac_p:.4f} | "
          f"diff {100 * rel:4.1f} % {'ok' if ok else ('differs (run without the profiler)' if info else 'MISMATCH' if e['name'] == d['roofline']['kernel'] else 'differs (not the headline kernel)')}")
  head = d["roofline"]
  print(f"headline: {head['kernel']} frac {head['frac']}  (= {head['bytes_per_launch']} B / {head['avg_launch_us']} us / {PEAK} GB/s)")
  sys.exit(1 if bad else 0)


if __name__ == "__main__":
  main()
