#!/usr/bin/env python3
"""bench.py -- cells/sec of VAE training on the pbmc8k_ly-shaped workload
(BASELINE.json configs[1]: VAE, hidden 128, latent 32, ZINB, batch 128).

One "step" = one optimiser step of the hot path on one minibatch: gather +
log1p + encoder, reparameterised latent, decoder, ZINB+KL ELBO, backward,
(all-reduce), clipnorm + Adam; inputs already resident in HBM.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU; ONE dataset, rank r keeps the r-th contiguous 1/N of its cells
resident; one RCCL all-reduce of the flat gradient buffer per step.
--scaling weak (default): 128 cells per GPU per step (global batch 128 N) -- the mode the
north star's ">= 6.5x at 8 GPUs" can be claimed under, a latency-bound 128-cell step cannot
be cut in eight.  --scaling strong: the reference's global batch is preserved (128 / N cells
per GPU); with --sync-bn that is the single-process arithmetic (SyncBatchNorm).
Rank 0 prints ONE JSON line.  At N > 1 the line also carries `scaling_modes`: the same workload measured in-run under weak scaling,
under strong scaling + SyncBatchNorm (the reference's global batch 128 and its single-process arithmetic) and BASELINE
configs[4] (1e6 / N cells generated per rank, 128 cells per GPU), each with the step's collective alone (`allreduce_us`), the same
per-GPU step without a communicator (`nocomm_ms_per_step`) and their difference (`dp_overhead_us`), beside DESIGN.md section 5's
predictions for N = 8.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

LOSS_REPEAT = 8   # SMX_LOSS_TIMING_REPEAT in sisua_amd/csrc/smx_model.h
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.3 TB/s achievable)


import functools  # noqa: E402


@functools.lru_cache(maxsize=4)
def _prepared(workload: str):
  """The workload's training matrix before sharding (read-only, cached: the world-8 tests cut the same matrix nine times):
  synthetic stand-in -> split(0.8) -> split(0.9) -> corrupt(train)."""
  from sisua_amd import data
  if workload in ("8kly", "8kly-2layer", "8kly-scvi"):
    x, y = data.synthetic_8kly(seed=8)
  elif workload == "eccly-sisua":
    x, y = data.synthetic_eccly(seed=8)
  elif workload == "cortex-base":
    x, y = data.synthetic_cortex(seed=8)
  else:
    raise ValueError(workload)
  tr, _ = data.split_indices(x.shape[0], 0.8, seed=1)
  tr2, _ = data.split_indices(len(tr), 0.9, seed=1)
  xt = data.corrupt(x[tr][tr2], 0.2, 0.2, seed=8)
  xt[xt.sum(1) == 0, 0] = 1.0
  yt = y[tr][tr2] if y is not None else None
  for a in (xt, yt):
    if a is not None:
      a.flags.writeable = False
  return xt, yt


def build_workload(rank: int, world: int, workload: str, n_cells: int = 4096):
  """Reproduces on_train (sisua/train.py:118-147) on synthetic 8kly-shaped data:
  split(0.8) -> split(0.9) -> corrupt(train) -> library stats."""
  from sisua_amd import data
  from sisua_amd.config import ModelConfig
  if workload == "8kly":
    units, latent, batch = (128,), 32, 128
  elif workload == "8kly-2layer":
    units, latent, batch = (128, 128), 32, 128
  elif workload in ("8kly-scvi", "eccly-sisua"):
    # BASELINE.json configs[2] / configs[3]: SCVI nbd batch 256; SISUA zinb + ADT nb labels (10 %), alpha 10, batch 256
    units, latent, batch = (128,), 32, 256
  elif workload == "cortex-base":
    # the reference's own default run (configs/base.yaml: cortex, encoder / decoder units [64, 64], latent 12, zinbd,
    # batch 64; BASELINE.json configs[0] is the same data at batch 32)
    units, latent, batch = (64, 64), 12, 64
  elif workload == "c5-shard":
    # per-GPU slice of BASELINE.json configs[4] (1e6 x 20000 log-normal counts, 128 cells per GPU per step):
    # 4096 resident cells per GPU are enough to exercise the step at its real width (host-generated: the parity tests and the
    # CPU baseline need the rows on the host; `--workload c5` generates the full 1e6 / world cells per GPU on the device)
    rng = np.random.default_rng(8 + rank)   # generated per shard from (seed, rank), as SURVEY.md 8d describes for C5
    x = (np.floor(rng.lognormal(0.0, 1.0, size=(n_cells, 20000))) * (rng.uniform(size=(n_cells, 20000)) < 0.12)).astype(np.float32)
    x[:, 0] += 1
    units, latent, batch = (128,), 32, 128
  else:
    raise ValueError(workload)
  if workload == "c5-shard":
    xt, yt = x, None
  else:
    xt, yt = _prepared(workload)
    x = xt
  kw = dict(n_genes=x.shape[1], enc_units=units, dec_units=units, latent_dim=latent, batchnorm=True, dropout_enc=0.1,
            dropout_dec=0.1, input_dropout=0.0, log_norm=True, beta=1.0, lr=1e-3, clipnorm=100.0, seed=8)
  extra = {}
  if workload == "8kly-scvi":
    cfg = ModelConfig(model="scvi", likelihood="nbd", encl_units=(64,), **kw)
    extra["library"] = data.library_matrix(xt)
  elif workload == "eccly-sisua":
    cfg = ModelConfig(model="sisua", likelihood="zinb", labels=((yt.shape[1], "nb"),), alpha=10.0, **kw)
    extra["labels"] = [yt]
    extra["label_mask"] = data.label_mask(xt.shape[0], 0.1, 2, seed=1)
  elif workload == "cortex-base":
    cfg = ModelConfig(model="vae", likelihood="zinbd", **kw)
  else:
    cfg = ModelConfig(model="vae", likelihood="zinb", **kw)
  if world > 1 and workload != "c5-shard":   # ONE dataset: this rank's contiguous shard of the training cells
    lo, hi = data.shard_range(xt.shape[0], rank, world)
    extra = {k: ([a[lo:hi] for a in v] if k == "labels" else v[lo:hi]) for k, v in extra.items()}
    xt = xt[lo:hi]
    extra["cell_id_base"] = lo
  elif world > 1:
    extra["cell_id_base"] = rank * xt.shape[0]
  return cfg, xt, batch, extra


def cpu_quota() -> int:
  """CPUs this process may keep busy: the cgroup's CPU bandwidth quota (cpu.max: "<quota> <period>" microseconds) where there is one, else the
  affinity mask.  os.cpu_count() is the HOST's count (256 on the GPU boxes, of which a one-GPU job gets 16)."""
  n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
  try:
    q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max":
      n = min(n, max(1, int(round(int(q) / int(per)))))
  except (OSError, ValueError):
    pass
  return max(1, n)


def make_order(n_cells: int, batch: int, n_steps: int):
  from sisua_amd import data
  chunks, ep = [], 0
  have = 0
  while have < n_steps:
    bs = data.iter_batches(data.epoch_order(n_cells, ep, shuffle=1000, seed=1), batch, drop_remainder=True)
    chunks += bs
    have += len(bs)
    ep += 1
  return np.concatenate(chunks[:n_steps]).astype(np.int32)


def cpu_baseline(cfg, xt, batch, budget_s=12.0, threads=None, extra=None):
  """CPU baseline on a bounded sample of the same workload (as many steps as fit in ~budget_s), kind "port":
  * VAE workloads: the C / OpenMP fp32 port of the step (oracle/sisua_step.c, validated against the NumPy oracle in
    tests/test_oracle_cport.py) with its dense products through the OpenBLAS NumPy links (the sparse-input products keep their
    zero-skipping loops), swept over thread counts: `value` = the best of the sweep, `single_thread`, `threads_16` and the whole
    `sweep` beside it (round 3's loops alone peaked at 16 of 256 cores with 7.0 k cells/s);
  * other models: the NumPy float64 oracle with 16 BLAS threads (its SciPy special functions are single-threaded)."""
  extra = extra or {}
  order = make_order(xt.shape[0], batch, 400)
  ncpu = os.cpu_count() or 1
  quota = cpu_quota()   # what this job may use of them: a GPU box gives a 1-GPU job 16 CPUs of its 256 (cgroup cpu.max) -- threads beyond the quota are throttled, not added

  def run(step_fn, budget):
    t_start, done, t_steps = time.perf_counter(), 0, 0.0
    while True:
      rows = order[(done % 400) * batch:((done % 400) + 1) * batch]
      t0 = time.perf_counter()
      step_fn(done, rows)
      dt = time.perf_counter() - t0
      if done >= 2:  # first two steps warm caches / thread pools
        t_steps += dt
      done += 1
      if (time.perf_counter() - t_start > budget and done >= 4) or done >= 400:
        break
    timed = max(done - 2, 1)
    return batch * timed / t_steps, timed

  if cfg.model == "vae":
    os.environ.setdefault("OMP_PLACES", "cores")       # (read when the port's OpenMP runtime loads: its threads stay on neighbouring cores
    os.environ.setdefault("OMP_PROC_BIND", "close")    # instead of wandering over the host's 256 under the quota)
    from oracle import sisua_oracle as so
    from oracle.cport import CStep
    spec = so.Spec(**cfg.to_dict())
    cs = CStep(spec, so.init_params(spec))
    # 1, 2, 4, ... up to the quota (the sweep used to go on to 32 threads inside a 16-CPU quota and "slowed down" there)
    counts = [threads] if threads else sorted({n for n in (1, 2, 4, 8, 16, 32, 64, 128) if n <= quota} | {quota})
    sweep = {}
    for n in counts:
      cs.set_threads(n)
      v, timed = run(lambda i, rows: cs.train_step(xt[rows], rows, i), budget_s / len(counts))
      sweep[str(n)] = round(v, 1)
    best = max(sweep, key=sweep.get)
    what = (f"C/OpenMP fp32 port of the step (oracle/sisua_step.c), dense products through {cs.blas or 'its own loops (no BLAS found)'}; "
            f"thread sweep {sweep} cells/s, ~{budget_s / len(counts):.1f} s each")
    return dict(value=sweep[best], unit="cells/s", cores=int(best), kind="port", single_thread=sweep.get("1"), threads_16=sweep.get("16"),
                sweep=sweep, host_cpus=ncpu, cpu_quota=quota,
                sample=f"steps of batch {batch} of the same workload; {what} (this job's CPU quota: {quota} of the host's {ncpu} logical CPUs; threads bound to cores, OMP_PROC_BIND=close)")
  from threadpoolctl import threadpool_limits
  from oracle import sisua_oracle as so
  threads = threads or min(16, ncpu)
  spec = so.Spec(**cfg.to_dict())
  params = so.init_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  x64 = xt.astype(np.float64)

  def step_fn(i, rows):
    with threadpool_limits(limits=threads):
      so.train_step(spec, params, bn, opt, x64[rows], so.PhiloxNoise(spec.seed, i, rows),
                    y=[y[rows] for y in extra.get("labels", [])],
                    library=extra["library"][rows] if "library" in extra else None,
                    mask=extra["label_mask"][rows] if "label_mask" in extra else None)
  v, timed = run(step_fn, budget_s)
  return dict(value=round(v, 1), unit="cells/s", cores=threads, kind="port",
              sample=f"{timed} steps of batch {batch} of the same workload; NumPy float64 oracle (oracle/sisua_oracle.py), {threads} BLAS threads (host has {ncpu} cores)")


def measure_mode(cp, rank, world, local_rank, cfg, batch, steps, warmup, upload, n_cells, sync_bn=False):
  """One data-parallel mode, measured in-run on every rank: (a) the per-GPU step WITHOUT a communicator (what one GPU does with this
  rank's batch), (b) the same step with the job's collective, K steps between barrier + synchronize brackets, max over ranks,
  (c) the collective alone.  Returns the dict that goes into `scaling_modes` (rank 0's copy is printed) and the engine is closed."""
  from sisua_amd.engine import Engine
  from sisua_amd.parallel import attach_engine, calibrate_forms
  eng = Engine(cfg, max_batch=batch, device=local_rank)
  upload(eng)
  order = make_order(n_cells, batch, warmup + steps)

  def timed_steps():
    for _ in range(50):
      eng.eval_step(order[:batch])
    if warmup:
      eng.train_steps(order[: warmup * batch], warmup, batch)
    eng.stage_steps(order[warmup * batch:], steps, batch)
    eng.synchronize()
    cp.barrier()
    if eng.world > 1:
      try:
        eng.comm_time_allreduce(1)   # device-side line-up of the ranks (the TCP barrier leaves ~100 us of skew)
      except Exception as err:
        print(f"bench: device-side line-up skipped: {err}", file=sys.stderr)
    t0 = time.perf_counter()
    eng.train_steps(None, steps, batch)
    eng.synchronize()
    t1 = time.perf_counter()
    cp.barrier()
    return cp.max(t1 - t0)

  dt0 = timed_steps()                       # (a) no communicator: every rank runs alone, max over ranks
  collective = attach_engine(eng, cp)
  if sync_bn:
    eng.set_sync_bn(True)
  calib = calibrate_forms(eng, cp, collective, make_order(n_cells, batch, 35), batch)   # the exchange form: measured, the same on every rank
  dt = timed_steps()                        # (b) the data-parallel step
  us, nbytes = -1.0, None
  try:
    us, nbytes = eng.comm_time_allreduce(50)   # (c) the collective alone
  except Exception as err:
    print(f"bench: timing the collective failed: {err}", file=sys.stderr)
  us = cp.max(us)
  err = eng.comm_p2p_error() if collective in ("p2p", "p2p-only") else 0
  if cp.max(float(err)) > 0:
    sys.exit("bench: the peer-to-peer exchange reported a timed-out wait; the timed steps are void")
  hist = eng.metrics_history(steps)
  if not np.isfinite(hist["loss"]).all():
    sys.exit("bench: non-finite loss in the timed steps")
  form = Engine.FORM_NAMES.get(eng.comm_form, "?")
  eng.close()
  return {"cells_per_s": round(steps * batch * world / dt, 1), "ms_per_step": round(1e3 * dt / steps, 4),
          "batch_per_gpu": batch, "global_batch": batch * world, "sync_bn": bool(sync_bn), "collective": collective, "exchange": form,
          "exchange_forms_us_per_step": {Engine.FORM_NAMES[k]: v for k, v in calib["us_per_step"].items()},
          "allreduce_us": round(us, 1) if us >= 0 else None, "allreduce_bytes": nbytes,
          "nocomm_ms_per_step": round(1e3 * dt0 / steps, 4), "dp_overhead_us": round(1e6 * (dt - dt0) / steps, 1),
          "final_loss": round(float(hist["loss"][-1]), 4)}


# DESIGN.md section 5 "What N = 8 is expected to print": the per-step budget table turned into figures (one MI355X node, RCCL over xGMI).
# These are PREDICTIONS written before any multi-GPU run existed (no node was available to the builder in rounds 1-5); the driver's scaling
# record is what replaces them.  Round 6: which exchange form a step takes is measured in the run (dp.forms_us_per_step).  Round 5: the step exchanges two buckets as a chain (head bucket all-reduced, normed and applied on the
# communication stream beside the rest of the step; front bucket on the model's stream), so what is left on the critical path is the front
# bucket's all-reduce and the second queue's cost (profiles/r05_dp_overhead_one_rank.txt).
PREDICTED_N8 = {
    "weak": {"ms_per_step": [0.084, 0.126], "cells_per_s": [8.1e6, 12.2e6], "x_one_gpu": [4.0, 6.0],
             "reading": "63 us of step + the exchange form the run measures fastest (round 6: parallel.calibrate_forms) -- one all-reduce: + 3 us of machinery + a "
                        "blocking all-reduce of 4.2 MB (latency-bound: 30-60 us expected from RCCL); two-bucket chain: + 16 us (one rank, measured) + the 1.1 MB front "
                        "bucket (20-40 us); hand-written exchange, one launch: + 5.5 us (one rank, measured) + 2 x 3.7 MB over seven links at once (15-30 us expected): "
                        "the >= 6.5x target (<= 78 us per step) is still NOT expected at 128 cells per GPU"},
    "strong_syncbn": {"ms_per_step": [0.15, 0.23], "cells_per_s": [0.56e6, 0.85e6], "x_one_gpu": [0.28, 0.43],
                      "reading": "16 cells per GPU: the step stays a chain of 11 launch-bound launches (~62 us), + the front bucket's all-reduce + 4 small "
                                 "SyncBatchNorm collectives of ~10-20 us each: strong scaling at batch 128 is a slowdown, as DESIGN section 5 says"},
    "c5": {"ms_per_step": [0.234, 0.354], "cells_per_s": [2.9e6, 4.4e6], "x_one_gpu": [3.3, 5.0],
           "reading": "~144 us of step (153.7 us with the chain on one rank) + a blocking all-reduce of the 10.4 MB front bucket (34 us of link time at peak "
                      "two-shot, 80-120 us expected from RCCL's ring); the 30.7 MB head bucket runs beside the backward pass, the optimiser and the next "
                      "step's encoder / decoder (~100 us of window: 100 us of link time at peak, 120-200 us expected from the ring -- up to ~80 us of it exposed)"},
}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=300)
  ap.add_argument("--warmup", type=int, default=30)
  ap.add_argument("--workload", default="8kly")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--c5-cells", type=int, default=0, help="--workload c5: total cells over all ranks (default 1 000 000)")
  ap.add_argument("--no-c5-entry", action="store_true", help="skip the roofline entries at the C5-shard width (128 cells x 20 000 genes)")
  ap.add_argument("--storage", default="f32", choices=("f32", "u16", "csr"), help="resident count matrix: float32 (reference layout), uint16, or the non-zeros only (CSR)")
  ap.add_argument("--cpu-budget", type=float, default=15.0)
  ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                  help="N > 1: weak = the configuration's batch per GPU; strong = the global batch is preserved (batch / N per GPU)")
  ap.add_argument("--sync-bn", action="store_true", help="N > 1: SyncBatchNorm (global-batch statistics)")
  ap.add_argument("--no-scaling-modes", action="store_true", help="N > 1: skip the in-run weak / strong + SyncBatchNorm / C5 measurements")
  args = ap.parse_args()

  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if world != args.gpus:
    if world == 1 and args.gpus > 1:
      sys.exit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    args.gpus = world

  # the HIP library (binds /opt/rocm's runtime); ranks meet over the package's own TCP control plane (MASTER_ADDR / MASTER_PORT)
  from sisua_amd import _hip
  from sisua_amd.engine import Engine
  from sisua_amd.parallel import ControlPlane, attach_engine
  if os.environ.get("SMX_SHARE_GPU"):   # debugging aid: several ranks on one device (RCCL permitting)
    local_rank = local_rank % max(_hip.load().smx_device_count(), 1)
  _hip.require_gpu(local_rank)
  cp = ControlPlane(rank, world)

  c5_full = args.workload == "c5"
  if c5_full:
    # BASELINE.json configs[4] at its real residency: 1e6 cells x 20 000 genes in total, this rank's 1e6 / world cells
    # generated ON the device as uint16 from (seed, rank) (40 GB on one GPU, 5 GB per GPU on eight); 128 cells
    # per GPU per step (global batch 1024 on eight GPUs)
    cfg, _, batch, extra = build_workload(rank, world, "c5-shard", n_cells=8)
    extra = {}
    n_c5 = (args.c5_cells or 1_000_000) // world

    class _Shape:   # (only its shape is used below: the matrix lives on the device)
      shape = (n_c5, cfg.n_genes)
    xt = _Shape()
  else:
    cfg, xt, batch, extra = build_workload(rank, world, args.workload)
  cell_base = extra.pop("cell_id_base", 0)
  if args.scaling == "strong" and world > 1:
    if batch % world:
      sys.exit(f"--scaling strong: batch {batch} is not divisible by {world} ranks")
    batch //= world
  eng = Engine(cfg, max_batch=batch, device=local_rank)
  if c5_full:
    eng.generate_lognormal(n_c5, seed=8, rank=rank, storage="u16" if args.storage != "f32" else "f32")
  else:
    eng.upload(xt, cell_id_base=cell_base, storage=args.storage, **extra)
  cp.barrier()   # every rank's shard is resident before the first collective
  collective = attach_engine(eng, cp)
  if world > 1 and args.sync_bn:
    eng.set_sync_bn(True)

  # (eager launches: a captured hipGraph of the step replays at 88.7 us against 79.6 us, profiles/r05_graph_vs_eager.txt -- `--graph` went in round 5)
  order = make_order(xt.shape[0], batch, args.warmup + args.steps)
  # loads the code object and brings the device clocks up (evaluation passes: parameters and optimiser state untouched, so
  # the W warm-up + K timed TRAINING steps below are exactly the run they would be without them; a fresh box otherwise
  # spends the first milliseconds -- all of a 20-step run -- at idle clocks)
  for _ in range(200):
    eng.eval_step(order[:batch])
  if args.warmup:
    eng.train_steps(order[: args.warmup * batch], args.warmup, batch)
  # the timed steps' row ids are resident before the clock starts, like the matrix they index (inputs in HBM: an input pipeline
  # prefetches the next epoch's schedule while the current one runs)
  eng.stage_steps(order[args.warmup * batch:], args.steps, batch)
  eng.synchronize()
  cp.barrier()
  if world > 1:
    # the control plane's barrier is a TCP round (its replies leave rank 0 one after the other: ~100 us of skew, several per cent of a
    # 20-step run); a collective on the devices + synchronize lines the ranks up to microseconds before the clock starts
    try:
      eng.comm_time_allreduce(1)
    except Exception as err:   # (the line-up is a refinement of the barrier above, not a condition of the run)
      print(f"bench: device-side line-up skipped: {err}", file=sys.stderr)
  t0 = time.perf_counter()
  eng.train_steps(None, args.steps, batch)   # K steps queued by ONE library call, no host sync between them
  eng.synchronize()
  t1 = time.perf_counter()   # this rank's K steps are done (every step ends in a collective: so are everybody's)
  cp.barrier()
  dt = cp.max(t1 - t0)       # MAX over ranks of the K steps between the two barrier + synchronize brackets
  # N > 1: the step's collective ALONE (50 all-reduces of the zeroed flat gradient buffer through the path the steps take, events on
  # the model's stream; max over ranks) -- what share of ms_per_step is link time is the first thing to know about a scaling curve
  dp_info = None
  if world > 1:
    nbytes = None
    try:
      us, nbytes = eng.comm_time_allreduce(50)
    except Exception as err:   # (every rank still takes part in the max below)
      us = -1.0
      print(f"bench: timing the collective failed: {err}", file=sys.stderr)
    us = cp.max(us)
    if cp.max(float(eng.comm_p2p_error() if collective in ("p2p", "p2p-only") else 0)) > 0:
      sys.exit("bench: the peer-to-peer exchange reported a timed-out wait; the timed steps are void")
    dp_info = {"collective": collective, "allreduce_us": round(us, 1) if us >= 0 else None, "allreduce_bytes": nbytes,
               "exchange": Engine.FORM_NAMES.get(eng.comm_form, "?")}
  # the ELBO scalars of every timed step stayed on the device (smx_metrics_history): read after the clock has stopped
  hist = eng.metrics_history(args.steps)
  m = {k: float(v[-1]) for k, v in hist.items()}
  if not np.isfinite(hist["loss"]).all():
    sys.exit("bench: non-finite loss in the timed steps")
  # a longer run beside the contract's K steps (the driver's 20 steps are a 1.6 ms window): 300 more steps of the SAME engine,
  # ids staged, one library call, the same brackets -- reported as value_300 / ms_per_step_300, never as `value`
  n300 = 300
  order300 = make_order(xt.shape[0], batch, n300)
  eng.stage_steps(order300, n300, batch)
  eng.synchronize()
  cp.barrier()
  t0 = time.perf_counter()
  eng.train_steps(None, n300, batch, graph=False)
  eng.synchronize()
  t1 = time.perf_counter()
  cp.barrier()
  dt300 = cp.max(t1 - t0)

  # ---- roofline: every figure is ONE kernel's algorithmic bytes over that kernel's OWN duration -----------------
  # HIP events on the model's stream; the kernel is launched LOSS_REPEAT times back to back inside one event pair and the
  # pair's own overhead ("null": a pair around nothing) is removed, so what is left is the per-launch duration rocprofv3
  # reports for the same kernel (tools/check_roofline.py compares the two from profiles/).  Entries:
  #   * out_head_loss_kernel: rows a-9 + a-10 in one launch (output product with the ZINB likelihood fwd + bwd as its
  #     epilogue, P never stored) -- the dominant kernel of the step and the headline `frac`;
  #   * count_loss_kernel: the standalone likelihood fwd + bwd (a-10 alone, SURVEY.md 8d's (4+8k)G + 16D + 4 per cell) --
  #     what eval / predict / scoring use and what training uses under flag head_loss = 0;
  #   * the same two at the C5-shard width (128 cells x 20 000 genes per GPU), where the bytes are large enough for the
  #     HBM roofline to be the bound rather than the launch floor.
  # The earlier "fused - product-only" attribution is kept as a labelled secondary field only.
  def kernel_times(e, order_, batch_, n_ev):
    """Per-launch durations in us.  `x8`: LOSS_REPEAT launches back to back inside one event pair, the pair's own overhead
    removed -- the kernel's per-launch time in a stream of work, and the figure that agrees with rocprofv3's average
    duration for the dominant kernel (7.1 vs 7.6 us under the profiler; tools/check_roofline.py).  `x1`: ONE launch per
    pair minus the overhead of an empty pair -- reported for reference only: the empty pair's 5 us partly overlap the
    launch they bracket, so it UNDER-estimates (5.7 us for the same kernel)."""
    def timed(label):
      e.timing_enable(label)
      e.train_steps(order_[: n_ev * batch_], n_ev, batch_, graph=False)
      ms, n = e.timing_read()
      return 1e3 * ms / max(n, 1), n
    null_us, _ = timed("null")      # event pair around nothing: overhead of the timing method itself
    out = dict(null=null_us)
    for reps, tag in ((1, "x1"), (LOSS_REPEAT, "x8")):
      per = lambda us: max(us - null_us, 0.1) / reps
      fused_us, fused_n = timed(f"out_head@{reps}")
      prod_us, _ = timed(f"out_head_product@{reps}")
      e.set_flag("head_loss", False)
      alone_us, alone_n = timed(f"loss@{reps}")
      e.set_flag("head_loss", True)
      out[tag] = dict(fused=per(fused_us) if fused_n else None, product=per(prod_us) if fused_n else None, alone=per(alone_us),
                      fused_n=fused_n * reps, alone_n=alone_n * reps)
    e.timing_enable(None)
    return out

  def roofline_entries(e, cfg_, batch_, kt, tag):
    """Algorithmic bytes per launch: fused = what rows a-9 + a-10 must move when P stays in
    registers: W_out 4 Hp k Gp + bias 4 k Gp + decoder output 4 B Hp + counts 4 B G + dP 4 B k G + partials; standalone =
    SURVEY.md 8d's unfused loss kernel (4 + 8k) G + 16 D + 4 per cell (smx_loss_bytes_per_cell)."""
    k = cfg_.k
    G, H = cfg_.n_genes, cfg_.dec_units[-1]
    Gp, Hp = -(-G // 32) * 32, -(-H // 32) * 32
    fused_bytes = 4 * Hp * k * Gp + 4 * k * Gp + 4 * batch_ * Hp + 4 * batch_ * G + 4 * batch_ * k * G + 4 * batch_ * (Gp // 32)
    alone_bytes = e.loss_bytes_per_cell() * batch_
    ent = []
    one, many = kt["x1"], kt["x8"]
    whole = e.head_fused_bytes(batch_)   # > 0: the step's output head is smx_headfused.hip's ONE launch (a wide panel)
    if many["fused"] and whole:
      # rows a-9 + a-10 + the head's part of a-16: product, likelihood, dW / db and d d; neither P nor dP exists in memory
      ach = whole / (many["fused"] * 1e-6) / 1e9
      ent.append(dict(name=f"head_fused_kernel@{tag}", kernel_regex=r"head_fused_kernel", rows="a-9 + a-10 + a-16 (head): product, likelihood, dW, db, d d in one launch",
                      cells=batch_, genes=G, bytes_per_launch=whole, avg_launch_us=round(many["fused"], 3), achieved=round(ach, 1),
                      frac=round(ach / HBM_PEAK_GBS, 4), launches_timed=many["fused_n"], single_launch_minus_empty_pair_us=round(one["fused"], 3),
                      bound_note="vector + matrix issue, not HBM: ~133 vector instructions per likelihood element and 18 bf16 MFMAs per 16 x 16 x 32 of each of "
                                 "the three products (DESIGN.md section 4)"))
    elif many["fused"]:
      ach = fused_bytes / (many["fused"] * 1e-6) / 1e9
      ent.append(dict(name=f"out_head_loss_kernel@{tag}", kernel_regex=r"out_head_loss_kernel<[0-9]+, ?[0-9]+, ?1,", rows="a-9 + a-10 (fused)", cells=batch_, genes=G,
                      bytes_per_launch=fused_bytes, avg_launch_us=round(many["fused"], 3), achieved=round(ach, 1),
                      frac=round(ach / HBM_PEAK_GBS, 4), launches_timed=many["fused_n"], single_launch_minus_empty_pair_us=round(one["fused"], 3)))
    ach = alone_bytes / (many["alone"] * 1e-6) / 1e9
    ent.append(dict(name=f"count_loss_kernel@{tag}", kernel_regex=r"count_loss_kernel", rows="a-10 (standalone fwd+bwd)", cells=batch_, genes=G,
                    bytes_per_launch=alone_bytes, avg_launch_us=round(many["alone"], 3), achieved=round(ach, 1),
                    frac=round(ach / HBM_PEAK_GBS, 4), launches_timed=many["alone_n"], single_launch_minus_empty_pair_us=round(one["alone"], 3)))
    return ent

  n_ev = 200   # (independent of --steps: the driver's 20-step run must time as many launches as a long one)
  order_ev = make_order(xt.shape[0], batch, n_ev)
  kt = kernel_times(eng, order_ev, batch, n_ev)
  null_us = kt["null"]
  per_kernel = {}
  for name in ("gemm_enc_fwd", "bn_fwd", "out_head", "gemm_out_bwd", "bn_bwd", "gemm_enc_dw", "adam", "step"):
    eng.timing_enable(name)
    n_k = 50
    eng.train_steps(order_ev[: n_k * batch], n_k, batch, graph=False)
    ms, n = eng.timing_read()
    # per launch, the event pair's own overhead removed; the fused head is launched LOSS_REPEAT times per pair
    per_kernel[name] = round(max(1e3 * ms / max(n, 1) - null_us, 0.0) / (LOSS_REPEAT if name == "out_head" else 1), 2)
  eng.timing_enable(None)
  entries = roofline_entries(eng, cfg, batch, kt, args.workload) if rank == 0 else []
  c5_step_us = None
  if rank == 0 and world == 1 and args.workload == "8kly" and not args.no_c5_entry:
    # the same kernels at the width of BASELINE.json configs[4]'s per-GPU share, in the same run
    cfg5, x5, b5, _ = build_workload(0, 1, "c5-shard")
    e5 = Engine(cfg5, max_batch=b5, device=local_rank)
    e5.upload(x5, storage="u16")
    # (timed exactly as `--workload c5-shard` is: evaluation passes to bring the clocks up, warm-up steps, the timed steps' ids staged,
    # 100 steps queued by one call -- round 3 timed 30 un-staged steps here and disagreed with profiles/ by 12 %)
    n5w, n5t = 20, 100
    o5 = make_order(x5.shape[0], b5, n5w + n5t)
    for _ in range(50):
      e5.eval_step(o5[:b5])
    e5.train_steps(o5[: n5w * b5], n5w, b5, graph=False)
    e5.stage_steps(o5[n5w * b5:], n5t, b5)
    e5.synchronize()
    t5 = time.perf_counter()
    e5.train_steps(None, n5t, b5, graph=False)
    e5.synchronize()
    c5_step_us = round(1e6 * (time.perf_counter() - t5) / n5t, 1)
    entries += roofline_entries(e5, cfg5, b5, kernel_times(e5, o5, b5, 30), "c5-shard")
    e5.close()

  # ---- the scoring path: marginal_log_prob of one batch with the reference's 100 posterior draws (posterior.py:964) ----
  scoring = None
  if rank == 0 and world == 1 and args.workload == "8kly" and eng.max_batch >= batch:
    try:
      rows_s = order_ev[:batch]
      for _ in range(30):   # (the clocks come back up over the first ~20 calls behind the idle stretch of the engines' set-up: 223 us at the median of calls 4-23, 203 later)
        eng.marginal_llk(row_ids=rows_s, n_samples=100)
      ts = []
      for _ in range(50):
        t_s = time.perf_counter()
        eng.marginal_llk(row_ids=rows_s, n_samples=100)
        ts.append(time.perf_counter() - t_s)
      us_call = 1e6 * float(np.median(ts))
      # the bound of this path is vector issue, not memory (the 100 draws' planes, 396 MB, never exist in memory).  MEASURED instruction counts of
      # the head kernel (PMC, profiles/r05_scoring_pmc.txt): the walk (score_walk_kernel) issues 1356 vector instructions + 144 bf16 MFMAs per wave and
      # 32 x 32 x 3-plane tile = 85 vector instructions per likelihood element (the tile-per-workgroup form 108, round 4's straight-line likelihood 230).
      # Bound = vector issue alone, one wave-instruction per 4 cycles on 1024 SIMDs at 2.4 GHz, the MFMAs (54 us by themselves) taken as hidden
      # beside it; the call also holds ten small launches and the host's return (~90 us).
      elems = batch * 100 * cfg.n_genes
      valu_bound_us = elems * 85.0 / 64.0 / (1024 * 2.4e9 / 4.0) * 1e6
      scoring = {"what": f"smx_marginal_llk: {batch} cells x 100 posterior draws x {cfg.n_genes} genes, one call (host-synchronous, median of 50)",
                 "marginal_llk_us": round(us_call, 1), "draws_per_s": round(batch * 100 / (us_call * 1e-6), 0),
                 "likelihood_elements_per_s": round(elems / (us_call * 1e-6), 0),
                 # (NOT an HBM or MFMA roofline: a vector-issue yardstick of the builder's own making, VERDICT r05 weak 11 -- named as such; round 5's
                 # `frac_230`, the same formula at a retired instruction count, is gone)
                 "issue_bound": {"bound": "valu", "model": "85 vector instructions per element (measured: SQ_INSTS_VALU of score_walk_kernel) / 64 lanes / (1024 SIMDs x 2.4 GHz / 4 cycles per instruction); "
                                                           "a self-defined yardstick, not a roofline of the memory system or the matrix cores",
                                 "bound_us": round(valu_bound_us, 1), "frac": round(valu_bound_us / us_call, 4)}}
    except Exception as err:
      scoring = {"error": str(err)[:200]}

  if rank == 0:
    head = entries[0]
    fused = kt["x8"]["fused"] is not None
    tfile = os.path.join(ROOT, "profiles", "loss_traffic_bytes.json")
    traffic_prof = None
    if os.path.exists(tfile):
      try:
        traffic_prof = json.load(open(tfile)).get(args.workload + ("" if not fused else ":fused"))
      except Exception:
        traffic_prof = None
    secondary = None
    if fused:
      t_attr = max(kt["x8"]["fused"] - kt["x8"]["product"], 0.05)
      unfused_bytes = eng.loss_bytes_per_cell() * batch
      secondary = {"what": "NOT a kernel figure: SURVEY.md 8d's unfused likelihood bytes over (fused kernel - the same kernel without "
                           "the likelihood); kept for continuity with rounds 1-2 only",
                   "unfused_likelihood_bytes": unfused_bytes, "fused_kernel_us": round(kt["x8"]["fused"], 3),
                   "product_only_us": round(kt["x8"]["product"], 3), "attributed_us": round(t_attr, 3),
                   "bytes_over_attributed_time_gbs": round(unfused_bytes / (t_attr * 1e-6) / 1e9, 1)}
    if c5_full:
      resident_as = f"this rank's 1/{world} of {n_c5 * world} cells generated on the device from (seed, rank), resident as uint16"
    elif args.workload == "c5-shard":
      resident_as = f"log-normal counts, resident as {args.storage}"
    else:
      resident_as = f"train split, corrupted; resident as {args.storage}"
    out = {
        "metric": "cells/sec VAE training (pbmc8k_ly, batch=128)" if args.workload == "8kly" else f"cells/sec {cfg.model} training ({args.workload}, batch={batch})",
        "value": round(args.steps * batch * world / dt, 1),
        "unit": "cells/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4),
        "value_300": round(n300 * batch * world / dt300, 1), "ms_per_step_300": round(1e3 * dt300 / n300, 4),
        "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}-shaped synthetic counts {xt.shape[0]}x{xt.shape[1]} "
                               f"({resident_as}), {cfg.model} {cfg.likelihood} hidden={list(cfg.enc_units)} latent={cfg.latent_dim}, "
                               f"batch {batch}/GPU, eager launches",
                   "global_batch": batch * world, "parallelism": f"dp{world}" + ("+syncbn" if (world > 1 and args.sync_bn) else "")},
        "final_loss": round(m["loss"], 4),
        "roofline": {"bound": "hbm", "kernel": head["name"], "rows": head["rows"],
                     "achieved": head["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": head["frac"],
                     "traffic": None, "traffic_from_profiles": traffic_prof,
                     "bytes_per_launch": head["bytes_per_launch"], "avg_launch_us": head["avg_launch_us"],
                     "entries": entries, "c5_shard_step_us": c5_step_us,
                     "attribution_secondary": secondary,
                     "event_pair_overhead_us": round(null_us, 3), "launches_per_event_pair": LOSS_REPEAT},
        "kernel_us": per_kernel,
    }
    if scoring is not None:
      out["scoring"] = scoring
    if dp_info is not None:
      out["dp"] = dp_info
  else:
    out = None
  # N > 1, the late phases: the line above is COMPLETE by now (its K steps ran with ONE all-reduce of the flat buffer, the north star's form and
  # the only one a multi-GPU node has ever run).  What follows has never run on N > 1 GPUs: (1) the exchange form
  # chosen by measurement -- parallel.calibrate_forms times one all-reduce / the two-bucket chain / the hand-written exchange on this job's own
  # step, max over ranks; if another form wins, the contract's W + K steps are run again with it and THAT is `value` --, (2) the scaling
  # modes.  A watchdog on EVERY rank (SMX_BENCH_MODES_BUDGET_S, default 300 s from here): if a phase hangs or a rank dies, rank 0 prints the
  # line as far as it got and every rank leaves (a watchdog on rank 0 alone left the others in their collectives for good).
  scaling_modes = None
  late = {"phase": "exchange-form calibration"}
  if world > 1:
    import threading
    def _give_up():
      if rank == 0:
        line = dict(out)
        line.setdefault("dp", {})["late_phases"] = f"not finished within the budget (stopped in: {late['phase']}); everything above is complete"
        if not args.no_scaling_modes and args.workload == "8kly":
          line["scaling_modes"] = {"error": "not finished within the budget; the line above them is complete", "finished": sorted(k for k in (scaling_modes or {}) if k[:1] != "_")}
        print(json.dumps(line), flush=True)
      else:
        time.sleep(3.0)   # (rank 0 prints first)
      os._exit(0)
    watchdog = threading.Timer(float(os.environ.get("SMX_BENCH_MODES_BUDGET_S", "300")), _give_up)
    watchdog.daemon = True
    watchdog.start()
    from sisua_amd.parallel import calibrate_forms
    first_form = eng.comm_form
    calib = calibrate_forms(eng, cp, collective, make_order(xt.shape[0], batch, 35), batch)
    dp_line = {"first_form": Engine.FORM_NAMES.get(first_form, "?"), "first_form_ms_per_step": round(1e3 * dt / args.steps, 4),
               "forms_us_per_step": {Engine.FORM_NAMES[k]: v for k, v in calib["us_per_step"].items()},
               "selected": Engine.FORM_NAMES.get(calib["selected"], "?"),
               "how": f"parallel.calibrate_forms: {calib['steps']} steps per form from the same state, max over ranks; SMX_DP_FORM / SMX_DP_CALIBRATE=0 override"}
    if calib["selected"] != first_form:
      late["phase"] = "the contract's steps with the selected form"
      if args.warmup:
        eng.train_steps(order[: args.warmup * batch], args.warmup, batch)
      eng.stage_steps(order[args.warmup * batch:], args.steps, batch)
      eng.synchronize()
      cp.barrier()
      try:
        eng.comm_time_allreduce(1)
      except Exception as err:
        print(f"bench: device-side line-up skipped: {err}", file=sys.stderr)
      t0 = time.perf_counter()
      eng.train_steps(None, args.steps, batch)
      eng.synchronize()
      t1 = time.perf_counter()
      cp.barrier()
      dt_sel = cp.max(t1 - t0)
      h2 = eng.metrics_history(args.steps)
      dp_line["selected_ms_per_step"] = round(1e3 * dt_sel / args.steps, 4)
      if np.isfinite(h2["loss"]).all() and dt_sel < dt and rank == 0:
        out.update(value=round(args.steps * batch * world / dt_sel, 1), ms_per_step=round(1e3 * dt_sel / args.steps, 4), final_loss=round(float(h2["loss"][-1]), 4))
        dp_line["value_is"] = "the selected form's K steps"
      elif rank == 0:
        dp_line["value_is"] = "the first form's K steps (the selected form was not faster over the contract's K)"
    else:
      dp_line["value_is"] = "the first form's K steps (it is the selected form)"
    if rank == 0:
      out["dp"].update(dp_line)
      out["dp"]["exchange"] = Engine.FORM_NAMES.get(eng.comm_form, "?")
    if args.no_scaling_modes or args.workload != "8kly":
      watchdog.cancel()
  if world > 1 and not args.no_scaling_modes and args.workload == "8kly":
    scaling_modes = {}
    late["phase"] = "scaling modes"
    k_steps, k_warm = max(args.steps, 20), max(args.warmup, 5)
    cfg8, x8, b8, ex8 = build_workload(rank, world, "8kly")
    base8 = ex8.pop("cell_id_base", 0)
    up8 = lambda e: e.upload(x8, cell_id_base=base8, storage=args.storage, **ex8)
    def run_mode(name, **more):   # (a mode that fails on ANY rank ends the modes on every rank: its peers may still be inside its collectives)
      def go(*a, **kw):
        late["phase"] = f"scaling mode {name}"
        if scaling_modes.get("_stopped"):
          return
        failed = 0.0
        try:
          scaling_modes[name] = dict(measure_mode(*a, **kw), **more)
        except (Exception, SystemExit) as err:
          scaling_modes[name] = {"error": str(err)[:300]}
          failed = 1.0
        if cp.max(failed) > 0:   # (ranks that did not fail arrive here too -- or meet the watchdog)
          scaling_modes.setdefault(name, {})["stopped_here"] = "a rank failed in this mode; the later modes were not started"
          scaling_modes["_stopped"] = True
      return go
    run_mode("weak")(cp, rank, world, local_rank, cfg8, b8, k_steps, k_warm, up8, x8.shape[0])
    if b8 % world == 0:
      run_mode("strong_syncbn")(cp, rank, world, local_rank, cfg8, b8 // world, k_steps, k_warm, up8, x8.shape[0], sync_bn=True)
    cfg5, _, b5, _ = build_workload(rank, world, "c5-shard", n_cells=8)
    n5 = (args.c5_cells or 1_000_000) // world
    up5 = lambda e: e.generate_lognormal(n5, seed=8, rank=rank, storage="u16")
    run_mode("c5", cells_resident_per_gpu=n5, storage="u16, generated on the device from (seed, rank)")(cp, rank, world, local_rank, cfg5, b5, k_steps, k_warm, up5, n5)
    watchdog.cancel()
  if rank == 0:
    if scaling_modes is not None:
      scaling_modes.pop("_stopped", None)
      for k_, v_ in scaling_modes.items():
        v_["predicted_n8"] = PREDICTED_N8.get(k_)
      out["scaling_modes"] = scaling_modes
      out["scaling_modes_note"] = ("each mode measured in this run on every rank (max over ranks): the data-parallel step, the same per-GPU step "
                                   "without a communicator (nocomm_ms_per_step), their difference (dp_overhead_us) and the collective alone "
                                   "(allreduce_us); predicted_n8 = DESIGN.md section 5's figures for one 8-GPU node, written before any multi-GPU run")
    if world == 1 and not args.no_cpu_baseline and not c5_full:
      try:
        out["cpu_baseline"] = cpu_baseline(cfg, xt, batch, args.cpu_budget, extra=extra)
      except Exception as err:   # the GPU line must not be lost to the host-side baseline (e.g. no compiler on the box)
        out["cpu_baseline"] = {"value": None, "unit": "cells/s", "cores": 0, "kind": "port", "sample": f"failed: {err}"[:200]}
    print(json.dumps(out), flush=True)
  eng.close()
  cp.close()


if __name__ == "__main__":
  main()
