#!/usr/bin/env python3
"""Reduce the PMC passes of tools/pmc_pass.sh to per-launch HBM bytes of the likelihood kernels (fused, product-only, standalone).
Counter unit and the gfx950 FETCH_SIZE correction are CALIBRATED on copy kernels of known size
(64 MiB read + 64 MiB written, 16 B/lane and 4 B/lane) run in the same passes."""
import csv
import glob
import json
import sys

out = sys.argv[1]
# kernel families of the default bench command: the fused training kernel, its product-only timing variant, the
# standalone likelihood kernel (eval / flag head_loss = 0), the head's backward products
FAMILIES = {"fused": "out_head_loss_kernel<1, 0, 1", "product_only": "out_head_loss_kernel<1, 0, 0",
            "standalone": "count_loss_kernel<1, 0, 1", "head_bwd": "out_head_bwd_kernel",
            # the wide-panel forms (bench.py --workload c5-shard): the whole head in one launch, the optimiser, the encoder's wide products
            "head_fused": "head_fused_kernel", "adam": "adam_update_kernel", "enc_front": "bigk_kernel<", "enc_wgrad": "wgrad_panel_group_kernel",
            # the BatchNorm launches that sum the wide products' column-major slabs themselves (round 5: no reduce launch)
            "bn_wide_fwd": "bn_wide_fwd_kernel", "bn_wide_bwd": "bn_wide_bwd_kernel"}


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
  fam = {}
  for name, pat in FAMILIES.items():
    vals = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == c and pat in r["Kernel_Name"]]
    fam[name] = dict(mean=sum(vals) / max(len(vals), 1), n=len(vals))
  res[c] = dict(calib_16B_per_lane=last16, calib_4B_per_lane=last4, kernels=fam)
known = 64 * 1024 * 1024
summary = {"raw": res, "hbm_bytes_per_launch": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
  cal4, cal16 = res[c]["calib_4B_per_lane"], res[c]["calib_16B_per_lane"]
  if cal4:
    summary[c + "_bytes_per_unit_4B"] = known / cal4    # calibrated on 4 B / lane accesses (what these kernels issue)
  if cal16:
    summary[c + "_bytes_per_unit_16B"] = known / cal16
for name in FAMILIES:
  tot, ok = 0.0, True
  for c in ("FETCH_SIZE", "WRITE_SIZE"):
    u = summary.get(c + "_bytes_per_unit_4B")
    k = res[c]["kernels"][name]
    if not u or not k["n"]:
      ok = False
      break
    summary["hbm_bytes_per_launch"][f"{name}:{c}"] = k["mean"] * u
    tot += k["mean"] * u
  if ok:
    summary["hbm_bytes_per_launch"][name] = tot
print(json.dumps(summary, indent=1))
json.dump(summary, open(out + "/pmc_summary.json", "w"), indent=1)
