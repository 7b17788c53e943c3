#!/usr/bin/env python3
"""Microseconds per optimiser step of a bench workload, timed as bench.py times it (evaluation passes to bring the clocks up, warm-up steps, the
timed steps' ids staged, all of them queued by ONE library call) -- without the rest of the bench line: the quick A/B of a knob or a build.
    SMX_TUNING=no_fold_dz python tools/dev/step_time.py 8kly            # knobs: docs/LAB_NOTES.md
    python tools/dev/step_time.py c5-shard --storage u16 --steps 100
Prints three repetitions; under rocprofv3 (`rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 tools/dev/step_time.py ...`)
`tools/prof_summary.py DIR` gives the per-kernel table and one step's timeline."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("workload", nargs="?", default="8kly")
  ap.add_argument("--storage", default=None, help="f32 / u16 / csr (default: f32, u16 at the c5-shard width as bench.py's secondary entry)")
  ap.add_argument("--steps", type=int, default=0)
  ap.add_argument("--warmup", type=int, default=30)
  args = ap.parse_args()
  import bench
  from sisua_amd.engine import Engine
  cfg, x, b, extra = bench.build_workload(0, 1, args.workload)
  extra.pop("cell_id_base", None)
  steps = args.steps or (100 if args.workload.startswith("c5") else 300)
  e = Engine(cfg, max_batch=b, device=0)
  e.upload(x, storage=args.storage or ("u16" if args.workload.startswith("c5") else "f32"), **extra)
  o = bench.make_order(x.shape[0], b, steps + args.warmup)
  for _ in range(50):
    e.eval_step(o[:b])
  e.train_steps(o[: args.warmup * b], args.warmup, b, graph=False)
  for _ in range(3):
    e.stage_steps(o[args.warmup * b:], steps, b)
    e.synchronize()
    t = time.perf_counter()
    e.train_steps(None, steps, b, graph=False)
    e.synchronize()
    print("%s: %.1f us per step (%d steps of %d cells)" % (args.workload, 1e6 * (time.perf_counter() - t) / steps, steps, b), flush=True)
  e.close()


if __name__ == "__main__":
  main()
