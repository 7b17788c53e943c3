"""Host-side pieces of bench.py (no GPU): workload construction, the minibatch order, the contract of the
JSON line's static fields."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.mark.parametrize("workload,batch,model,lk", [("8kly", 128, "vae", "zinb"), ("8kly-2layer", 128, "vae", "zinb"),
                                                      ("8kly-scvi", 256, "scvi", "nbd"), ("eccly-sisua", 256, "sisua", "zinb")])
def test_workloads_follow_baseline_configs(workload, batch, model, lk):
  cfg, xt, b, extra = bench.build_workload(0, 1, workload)
  assert b == batch and cfg.model == model and cfg.likelihood == lk and cfg.n_genes == xt.shape[1]
  assert xt.dtype == np.float32 and (xt >= 0).all() and np.array_equal(xt, np.floor(xt))
  assert (xt.sum(1) > 0).all()                               # library size is the log of the total
  if workload.startswith("8kly"):
    assert xt.shape == (3381, 1998)                          # split(0.8) -> split(0.9) of 4697 cells
    assert 0.90 < (xt == 0).mean() < 0.96
  if model == "scvi":
    assert extra["library"].shape == (xt.shape[0], 2)
  if model == "sisua":
    assert extra["labels"][0].shape[0] == xt.shape[0] and extra["label_mask"].shape == (xt.shape[0],)
    assert 0.05 < extra["label_mask"].mean() < 0.15          # labels_percent = 0.1


def test_ranks_shard_one_dataset():
  """ONE matrix, rank r keeps the r-th contiguous 1/world of its cells; cell ids stay global."""
  c, x, _, e = bench.build_workload(0, 1, "eccly-sisua")
  c0, x0, _, e0 = bench.build_workload(0, 2, "eccly-sisua")
  c1, x1, _, e1 = bench.build_workload(1, 2, "eccly-sisua")
  n = x.shape[0] // 2
  assert c0 == c1 == c and x0.shape == x1.shape == (n, x.shape[1])
  assert np.array_equal(x0, x[:n]) and np.array_equal(x1, x[n:2 * n])
  assert (e0["cell_id_base"], e1["cell_id_base"]) == (0, n)
  assert np.array_equal(e1["labels"][0], e["labels"][0][n:2 * n]) and np.array_equal(e1["label_mask"], e["label_mask"][n:2 * n])


def test_order_is_whole_batches_of_valid_rows():
  order = bench.make_order(3381, 128, 57)
  assert order.dtype == np.int32 and order.size == 57 * 128 and order.min() >= 0 and order.max() < 3381
  first_epoch = order[: 26 * 128]
  assert len(np.unique(first_epoch)) == first_epoch.size    # an epoch never repeats a cell


def test_world_size_mismatch_is_refused():
  env = dict(os.environ, WORLD_SIZE="1", RANK="0")
  r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
  assert r.returncode != 0 and "torch.distributed.run" in (r.stderr + r.stdout)


def test_rccl_banner_is_kept_off_stdout():
  """RCCL prints a five-line banner with printf to STDOUT when a communicator is created; bench.py's stdout is ONE JSON line.
  parallel.stdout_to_stderr points file descriptor 1 at stderr around the call, C stdio flushed on both sides."""
  code = ("import ctypes\nfrom sisua_amd.parallel import stdout_to_stderr\nlibc = ctypes.CDLL(None)\nprint('{\"before\": 1}', flush=True)\n"
          "with stdout_to_stderr():\n  libc.printf(b'RCCL version : x\\n')\n  print('inside')\nlibc.printf(b'after-c\\n')\nprint('after')\n")
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
  assert r.returncode == 0, r.stderr
  assert "RCCL version" not in r.stdout and "inside" not in r.stdout and "RCCL version : x" in r.stderr and "inside" in r.stderr
  assert r.stdout.splitlines()[0] == '{"before": 1}' and sorted(r.stdout.splitlines()[1:]) == ["after", "after-c"]
  kept = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, SMX_KEEP_RCCL_BANNER="1"))
  assert "RCCL version : x" in kept.stdout


def test_the_line_carries_a_longer_run_and_the_graph_flag_is_gone():
  """`value` / `ms_per_step` stay the contract's K steps; `value_300` / `ms_per_step_300` (300 further staged steps of
  the same engine) stand beside them.  The hipGraph replay of the step measured 88.7 us against 79.6 us of eager launches
  (profiles/r05_graph_vs_eager.txt): `--graph` left the command line (DESIGN.md section 6 says why; the capture entry points stay in the
  C-ABI for callers that embed the step in a graph of their own)."""
  src = open(os.path.join(ROOT, "bench.py")).read()
  assert '"value_300"' in src and '"ms_per_step_300"' in src
  r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
  assert r.returncode == 0 and "--graph" not in r.stdout and "--steps" in r.stdout

