"""Data-parallel path on CPU, world_size 2 (and 3): the package's own TCP control plane (sisua_amd/parallel.py: no torch)
and, beside it, a torch.distributed gloo group in the same processes as an independent check of every collective.
Covers the DP contract the HIP library implements with one all-reduce: sum of the per-rank flat buffers
[grads | BN batch stats | metrics] / world == the single-process result on the concatenated minibatch with
per-replica BN."""
import os
import socket

import numpy as np
import pytest

from oracle import sisua_oracle as so
from tests.util import perturbed_params, synth_counts


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, out_dir):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
  from sisua_amd import data
  from sisua_amd.parallel import ControlPlane, env_rank_world
  r, lr, w = env_rank_world()
  assert (r, w) == (rank, world)
  cp = ControlPlane(r, w)
  uid = cp.broadcast_bytes(lambda: bytes(range(128)))         # stands in for the RCCL unique id
  assert uid == bytes(range(128))
  assert cp.max(1.0 + rank) == float(world)
  # shard the cells; every rank runs the same number of equal-size steps
  spec = so.Spec(model="vae", n_genes=40, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=4)
  x = synth_counts(203, 40, sparsity=0.7, seed=0)
  ids = data.shard_for_rank(np.arange(203), rank, world)
  B = 16
  rows = ids[:B]
  params = perturbed_params(spec)
  bn = so.init_bn_state(spec)
  res = so.forward_backward(spec, params, bn, x[rows], so.PhiloxNoise(spec.seed, 0, rows))
  # the library scales the loss by 1/(world*B); the oracle call above used 1/B
  flat = np.concatenate([res["grads"][n].ravel() / world for n, _ in so.manifest(spec)] +
                        [res["new_bn"]["enc0/batch_mean"] / world, np.array([res["loss"] / world])])
  tot = cp.sum_array(flat)
  # the same reduction over gloo (torch.distributed is test infrastructure here, not a dependency of the package)
  import torch
  import torch.distributed as dist
  dist.init_process_group(backend="gloo", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
  t = torch.from_numpy(flat.copy())
  dist.all_reduce(t, op=dist.ReduceOp.SUM)
  assert np.allclose(tot, t.numpy(), rtol=1e-15, atol=1e-300)
  tm = torch.tensor([1.0 + rank], dtype=torch.float64)
  dist.all_reduce(tm, op=dist.ReduceOp.MAX)
  assert cp.max(1.0 + rank) == float(tm.item())
  dist.barrier()
  dist.destroy_process_group()
  np.save(os.path.join(out_dir, f"rank{rank}.npy"), tot)
  np.save(os.path.join(out_dir, f"rows{rank}.npy"), rows)
  cp.close()


def test_world2_gloo(tmp_path):
  import torch.multiprocessing as mp   # (lazily: the TCP workers below re-import this module and must stay torch-free)
  world, port = 2, _free_port()
  mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
  a, b = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
  assert np.array_equal(a, b)                                 # every rank holds the same reduced buffer
  r0, r1 = np.load(tmp_path / "rows0.npy"), np.load(tmp_path / "rows1.npy")
  assert not set(r0) & set(r1)                                # disjoint shards
  # serial emulation: the two replicas' minibatches, per-replica BN statistics, mean of the two losses
  spec = so.Spec(model="vae", n_genes=40, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=4)
  x = synth_counts(203, 40, sparsity=0.7, seed=0)
  params, bn = perturbed_params(spec), so.init_bn_state(spec)
  parts = [so.forward_backward(spec, params, bn, x[r], so.PhiloxNoise(spec.seed, 0, r)) for r in (r0, r1)]
  ref = np.concatenate([np.mean([p["grads"][n] for p in parts], 0).ravel() for n, _ in so.manifest(spec)] +
                       [np.mean([p["new_bn"]["enc0/batch_mean"] for p in parts], 0), [np.mean([p["loss"] for p in parts])]])
  assert np.allclose(a, ref, rtol=1e-12, atol=1e-14)
  # without BatchNorm the DP gradient IS the gradient of the concatenated minibatch
  spec2 = so.Spec(model="vae", n_genes=40, likelihood="zinb", enc_units=(16,), dec_units=(16,), latent_dim=4, batchnorm=False)
  p2 = perturbed_params(spec2)
  both = np.concatenate([r0, r1])
  g_all = so.forward_backward(spec2, p2, {}, x[both], so.PhiloxNoise(spec2.seed, 0, both))["grads"]
  g_dp = [so.forward_backward(spec2, p2, {}, x[r], so.PhiloxNoise(spec2.seed, 0, r))["grads"] for r in (r0, r1)]
  for n in g_all:
    assert np.allclose(g_all[n], 0.5 * (g_dp[0][n] + g_dp[1][n]), rtol=1e-10, atol=1e-13)


def test_control_plane_single_process_is_noop():
  from sisua_amd.parallel import ControlPlane
  cp = ControlPlane(0, 1)
  assert cp.broadcast_bytes(lambda: b"x") == b"x" and cp.max(3.0) == 3.0
  cp.barrier()
  cp.close()


def _tcp_worker(rank, world, port, out_dir):
  import sys
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                    SMX_RUN_ID=f"t{port}")
  from sisua_amd.parallel import ControlPlane
  if rank == world - 1:
    import time
    time.sleep(0.5)    # a late joiner: the others wait in the rendezvous
  cp = ControlPlane(rank, world)
  assert "torch" not in sys.modules, "the control plane must not import torch"
  for i in range(3):
    cp.barrier()
    assert cp.broadcast_bytes(lambda: bytes([rank, i]) * 64, src=i % world) == bytes([i % world, i]) * 64
    assert cp.max(float(rank * (i + 1))) == float((world - 1) * (i + 1))
    a = np.arange(7, dtype=np.float64).reshape(7, 1) * (rank + 1) + i
    assert np.array_equal(cp.sum_array(a), np.arange(7, dtype=np.float64).reshape(7, 1) * (world * (world + 1) / 2) + i * world)
  big = np.full(300000, 1.0 / (rank + 1))                    # 2.4 MB frames
  tot = cp.sum_array(big)
  assert tot.shape == big.shape and np.allclose(tot, sum(1.0 / (r + 1) for r in range(world)))
  cp.close()
  open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


@pytest.mark.parametrize("world", [2, 3])
def test_tcp_control_plane(tmp_path, world):
  """The ~100-line TCP star that replaced torch.distributed as the control plane: rendezvous through
  the published port file (MASTER_PORT itself stays free for the launcher), barrier / broadcast / max / sum, a late
  joiner, frames of a few MB, and no torch import in the workers."""
  import multiprocessing as pymp
  ctx = pymp.get_context("spawn")
  port = _free_port()
  ps = [ctx.Process(target=_tcp_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
  for p in ps:
    p.start()
  for p in ps:
    p.join(120)
  assert all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
  assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_tcp_control_plane_fixed_port_and_stale_file(tmp_path, monkeypatch):
  """SMX_CP_PORT pins the port (any topology); a stale rendezvous file of an earlier job is retried past, not trusted."""
  import threading
  from sisua_amd.parallel import ControlPlane
  port = _free_port()
  monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
  monkeypatch.setenv("MASTER_PORT", "1")
  monkeypatch.setenv("SMX_CP_PORT", str(port))
  out = [None, None]

  def run(r):
    cp = ControlPlane(r, 2)
    out[r] = cp.sum_array(np.array([r + 1.0]))[0]
    cp.close()
  ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
  [t.start() for t in ts]; [t.join(60) for t in ts]
  assert out == [3.0, 3.0]


class _FakeEngine:
  """Stands in for sisua_amd.engine.Engine in attach_engine: records the calls, fails ncclCommInitRank on the ranks it is told to."""

  def __init__(self, fail):
    self.fail, self.calls = fail, []

  def comm_init(self, rank, world, uid):
    self.calls.append("comm_init")
    if self.fail:
      raise RuntimeError("ncclCommInitRank failed: invalid usage")

  def comm_p2p_export(self, world):
    self.calls.append("p2p_export")
    return bytes(128)

  def comm_p2p_init(self, rank, world, handles):
    self.calls.append(f"p2p_init:{len(handles)}")

  def comm_set_form(self, form):
    self.calls.append(f"set_form:{form}")


def _attach_worker(rank, world, port, out_dir, failing, mode_env=None):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), SMX_RUN_ID=f"a{port}")
  for k in ("SMX_ALLREDUCE", "SMX_DP_FORM", "SMX_OPT_SHARD"):
    os.environ.pop(k, None)
  if mode_env:
    os.environ["SMX_ALLREDUCE"] = mode_env
  import warnings
  from sisua_amd import engine as eng_mod
  from sisua_amd.parallel import ControlPlane, attach_engine
  eng_mod.Engine.comm_unique_id = staticmethod(lambda: bytes(128))   # (no GPU, no librccl on this box's path: the id is opaque here)
  cp = ControlPlane(rank, world)
  e = _FakeEngine(rank in failing)
  try:
    with warnings.catch_warnings(record=True) as w:
      warnings.simplefilter("always")
      mode = attach_engine(e, cp)
    res = f"{mode}|{','.join(e.calls)}|{int(any('falling back' in str(x.message) for x in w))}"
  except RuntimeError as err:
    res = f"raised:{err}|{','.join(e.calls)}"
  cp.barrier()
  cp.close()
  open(os.path.join(out_dir, f"res{rank}"), "w").write(res)


@pytest.mark.parametrize("failing,expect", [((), "auto"), ((), "rccl"), ((0, 1), "p2p-only"), ((1,), "raised")])
def test_attach_engine_agrees_on_the_collective(tmp_path, failing, expect):
  """parallel.attach_engine over the TCP control plane with two ranks and a stand-in engine (no GPU): RCCL comes up on every rank ->
  'auto' (the hand-written exchange attached beside it, ONE all-reduce in force until calibrate_forms has measured) or, with
  SMX_ALLREDUCE=rccl, RCCL alone; on NO rank -> every rank takes the peer-to-peer exchange (handles of both ranks gathered), rank 0 warns;
  on SOME ranks only -> every rank stops (a half-built communicator cannot be repaired), nobody hangs."""
  import multiprocessing as pymp
  ctx = pymp.get_context("spawn")
  port = _free_port()
  ps = [ctx.Process(target=_attach_worker, args=(r, 2, port, str(tmp_path), tuple(failing), "rccl" if expect == "rccl" else None)) for r in range(2)]
  for p in ps:
    p.start()
  for p in ps:
    p.join(120)
  assert all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
  res = [(tmp_path / f"res{r}").read_text() for r in range(2)]
  if expect == "auto":
    assert res == ["auto|comm_init,p2p_export,p2p_init:256,set_form:1|0"] * 2
  elif expect == "rccl":
    assert res == ["rccl|comm_init,set_form:1|0"] * 2
  elif expect == "p2p-only":
    assert res == ["p2p-only|comm_init,p2p_export,p2p_init:256|1", "p2p-only|comm_init,p2p_export,p2p_init:256|0"]
  else:
    assert all(r.startswith("raised:") for r in res), res


# ---- parallel.calibrate_forms: the exchange form chosen by measurement, identically on every rank ----------------
class _FakeCalibEngine:
  """What calibrate_forms touches of an Engine: `cost` seconds per step and form on THIS rank."""
  def __init__(self, cost, refuse=(), world=2):
    self.cost, self.refuse, self.form, self.world = cost, set(refuse), 1, world
    self.state, self.log = {"params": 0}, []

  comm_form = property(lambda self: self.form)

  def comm_set_form(self, f):
    if f in self.refuse:
      raise RuntimeError("form not available on this rank")
    self.form = f
    self.log.append(("form", f))

  def snapshot(self):
    return dict(self.state)

  def restore(self, st):
    self.state = dict(st)
    self.log.append(("restore", st["params"]))

  def train_steps(self, order, n, batch, **kw):
    import time
    self.state["params"] += n          # training moves the state ...
    time.sleep(self.cost[self.form] * n)

  def stage_steps(self, order, n, batch):
    pass

  def synchronize(self):
    pass

  def comm_time_allreduce(self, n):
    return 0.0, 0

  def comm_p2p_error(self):
    return 0


def _calib_worker(rank, world, port, out_dir):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
  for k in ("SMX_DP_FORM", "SMX_DP_CALIBRATE", "SMX_DP_FAKE_SLOW"):
    os.environ.pop(k, None)
  import json
  from sisua_amd.parallel import ControlPlane, calibrate_forms
  cp = ControlPlane(rank, world)
  order = np.arange(4096, dtype=np.int32)
  out = {}
  # (1) form 3 is the fastest on rank 0 but the slowest on rank 1: the MAX over ranks decides, the same on both
  e = _FakeCalibEngine({1: 2e-3, 2: 1.5e-3, 3: 0.2e-3 if rank == 0 else 4e-3})
  out["max_over_ranks"] = calibrate_forms(e, cp, "auto", order, 8, steps=6, warmup=1)
  out["state_restored"] = e.state["params"] == 0 and e.form == out["max_over_ranks"]["selected"]
  # (2) a form that one rank cannot set is dropped on EVERY rank
  e = _FakeCalibEngine({1: 2e-3, 2: 0.2e-3, 3: 1e-3}, refuse=(2,) if rank == 1 else ())
  out["refused_on_one_rank"] = calibrate_forms(e, cp, "auto", order, 8, steps=6, warmup=1)
  # (3) within 2 % of the best the one all-reduce wins
  e = _FakeCalibEngine({1: 1.0e-3, 2: 0.995e-3, 3: 3e-3})
  os.environ["SMX_DP_FAKE_SLOW"] = "1:0,2:0"
  out["tie"] = calibrate_forms(e, cp, "rccl", order, 8, steps=12, warmup=0)["us_per_step"]
  # (4) the test hook that fakes a slow form
  os.environ["SMX_DP_FAKE_SLOW"] = "1:5000,2:5000"
  e = _FakeCalibEngine({1: 1e-3, 2: 1e-3, 3: 1e-3})
  out["faked"] = calibrate_forms(e, cp, "auto", order, 8, steps=4, warmup=0)
  os.environ.pop("SMX_DP_FAKE_SLOW")
  # (5) switched off: nothing is measured, the form in force stays
  os.environ["SMX_DP_CALIBRATE"] = "0"
  e = _FakeCalibEngine({1: 1e-3, 2: 1e-4, 3: 1e-4})
  out["off"] = calibrate_forms(e, cp, "auto", order, 8)
  out["off_log"] = len(e.log)
  with open(os.path.join(out_dir, f"calib{rank}.json"), "w") as f:
    json.dump(out, f)
  cp.close()


def test_exchange_form_is_chosen_by_measurement_identically_on_every_rank(tmp_path):
  import json
  import multiprocessing as mp
  world, port = 2, _free_port()
  ctx = mp.get_context("spawn")
  ps = [ctx.Process(target=_calib_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
  for p in ps:
    p.start()
  for p in ps:
    p.join(120)
    assert p.exitcode == 0
  a, b = [json.load(open(tmp_path / f"calib{r}.json")) for r in range(world)]
  assert a == b                                                        # every rank made the same choices from the same numbers
  r = a["max_over_ranks"]
  assert r["selected"] == 2 and set(r["us_per_step"]) == {"1", "2", "3"} and r["us_per_step"]["3"] > r["us_per_step"]["1"] > r["us_per_step"]["2"]
  assert a["state_restored"]
  r = a["refused_on_one_rank"]
  assert r["us_per_step"]["2"] is None and r["selected"] == 3
  assert a["faked"]["selected"] == 3
  assert a["off"]["us_per_step"] == {} and a["off_log"] == 0
