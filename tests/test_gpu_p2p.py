"""The hand-written two-shot all-reduce over IPC-mapped peer buffers (sisua_amd/csrc/smx_p2p.hip) with REAL
processes: 2 and 3 fresh child processes share the box's one GPU (each opens the others' gradient buffers and communication
regions through HIP IPC -- on an 8-GPU node the same handles map peer memory over xGMI), join over the package's TCP control
plane, and run optimiser steps whose collective is the peer-to-peer exchange (no RCCL communicator: RCCL refuses two ranks on
one device).  Checked against oracle.dp_train_step, and the ranks against each other bit for bit.  A second test lets a peer
die: the survivor's bounded waits must give up and report, not hang the device."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import sisua_oracle as so
from tests.util import adam_state_errors, grad_errors, make_pair, perturbed_params, synth_counts

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KW = dict(model="vae", n_genes=203, likelihood="zinb", enc_units=(48, 40), dec_units=(40,), latent_dim=10)
KW_SYNC = dict(model="vae", n_genes=120, likelihood="nb", enc_units=(32,), dec_units=(32,), latent_dim=6)

_CHILD = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
rank, world, out, sync_bn, die = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
from tests.util import make_pair, perturbed_params, synth_counts
from sisua_amd.engine import Engine
from sisua_amd.parallel import ControlPlane
kw = {kw!r}
spec, cfg = make_pair(**kw)
x = synth_counts(400, spec.n_genes, sparsity=0.85, seed=0)
e = Engine(cfg, max_batch=64, init=False)
e.set_params(perturbed_params(spec))
e.upload(x, cell_id_base=1000)
cp = ControlPlane(rank, world)
if os.environ.get("SMX_TEST_ATTACH"):   # the package's own way in: RCCL first, the exchange if no communicator comes up
  from sisua_amd.parallel import attach_engine
  attach_engine(e, cp)
else:
  handles = cp.allgather_bytes(e.comm_p2p_export(world))
  e.comm_p2p_init(rank, world, b"".join(handles))
assert e.world == world and e.rank == rank
e.set_sync_bn(bool(sync_bn))
cp.barrier()
if die and rank == 1:
  cp.close()
  os._exit(0)                       # a peer that never takes part in the exchange
rng = np.random.default_rng(5)
res = dict()
if os.environ.get("SMX_TEST_CALIBRATE"):   # parallel.calibrate_forms between real processes: trial steps, then the state it found
  from sisua_amd.parallel import calibrate_forms
  rep = calibrate_forms(e, cp, "p2p-only", rng.permutation(400)[:(2 + 5) * 48].astype(np.int32), 48, steps=5, warmup=2)
  res["calib_selected"], res["calib_us"] = rep["selected"], rep["us_per_step"].get(3) or -1.0
  rng = np.random.default_rng(5)
for step in range(3):
  rows = rng.permutation(400)[: 48 * world].astype(np.int32).reshape(world, 48)
  try:
    m = e.train_step(rows[rank])
  except Exception as err:            # a failed exchange is REPORTED by the step (SMX_ERR_COMM), not left in an unread word
    res["raised"] = str(err)
    break
  res[f"loss{{step}}"] = m["loss"]; res[f"kl{{step}}"] = m["kl"]; res[f"gn{{step}}"] = m["grad_norm_max"]
  if step == 0:
    for k, v in e.get_params(which=1).items():
      res["g/" + k] = v
  if die:
    break
res["err"] = getattr(e, "last_comm_error", 0) or e.comm_p2p_error()   # (the raise has read AND cleared the word: ADVICE r04)
res["err_after"] = e.comm_p2p_error()
for k, v in e.get_params().items():
  res["p/" + k] = v
for k, v in e.get_params(which=2).items():
  res["m/" + k] = v
np.savez(out, **res)
if not die:
  cp.barrier()
  cp.close()
e.close()
print("DONE", flush=True)
"""


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _run(tmp_path, world, kw, sync_bn=0, die=0, timeout=180, extra_env=None):
  script = tmp_path / "p2p_child.py"
  script.write_text(_CHILD.format(root=ROOT, kw=kw))
  port = _free_port()
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="1", SMX_CP_PORT=str(port), WORLD_SIZE=str(world), SMX_RUN_ID=f"p2p{port}",
             **(extra_env or {}))
  ps = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(tmp_path / f"r{r}.npz"), str(sync_bn), str(die)],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(env, RANK=str(r)), text=True) for r in range(world)]
  outs = []
  for p in ps:
    try:
      out, _ = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
      for q in ps:
        q.kill()
      pytest.fail("a rank of the peer-to-peer exchange hangs")
    outs.append((p.returncode, out))
  return outs


@pytest.mark.parametrize("world,sync_bn,buckets", [(2, 0, 1), (3, 0, 1), (2, 1, 1), (2, 0, 2), (2, 1, 2), (2, 0, "calibrate")])
def test_p2p_allreduce_between_processes_matches_oracle(tmp_path, world, sync_bn, buckets):
  """buckets = 2 (SMX_DP_BUCKETS=2): the heads' gradients are exchanged on the communication stream while the rest of the
  backward pass runs, the remainder afterwards -- two exchanges per step, the same numbers.  With SyncBatchNorm the exchange
  keeps ONE bucket whatever the switch says (its staging / flags are shared with SyncBatchNorm's small collectives: ADVICE r03)."""
  from sisua_amd import build
  build.build(verbose=False)
  kw = KW_SYNC if sync_bn else KW
  # "calibrate": parallel.calibrate_forms first (the exchange is the only form two processes on ONE device have: its trial steps run, the
  # ranks agree on the figure, and the three steps checked below start from the state it found)
  extra = {"SMX_TEST_CALIBRATE": "1"} if buckets == "calibrate" else {"SMX_DP_BUCKETS": str(buckets)}
  outs = _run(tmp_path, world, kw, sync_bn=sync_bn, extra_env=extra)
  for rc, out in outs:
    assert rc == 0 and "DONE" in out, out[-3000:]
  got = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
  if buckets == "calibrate":
    assert all(int(g["calib_selected"]) == 3 and float(g["calib_us"]) > 0 for g in got)
    assert float(got[0]["calib_us"]) == float(got[1]["calib_us"])   # the MAX over the ranks, the same figure on both
  spec, _ = make_pair(**kw)
  x = synth_counts(400, spec.n_genes, sparsity=0.85, seed=0)
  params = perturbed_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  rng = np.random.default_rng(5)
  for step in range(3):
    rows = rng.permutation(400)[: 48 * world].astype(np.int32).reshape(world, 48)
    ref = so.dp_train_step(spec, params, bn, opt, x, list(rows), step, cell_base=1000, sync_bn=bool(sync_bn))
    for r in range(world):
      assert int(got[r]["err"]) == 0
      assert np.isclose(float(got[r][f"loss{step}"]), ref["metrics"]["loss"], rtol=1e-4), (r, step)
      assert np.isclose(float(got[r][f"kl{step}"]), ref["metrics"]["kl"], rtol=1e-4, atol=1e-5), (r, step)
      assert np.isclose(float(got[r][f"gn{step}"]), max(ref["norms"].values()), rtol=1e-3), (r, step)
      assert float(got[r][f"loss{step}"]) == float(got[0][f"loss{step}"])          # the ranks agree bit for bit
    if step == 0:
      for r in range(world):
        g = {k[2:]: got[r][k] for k in got[r].files if k.startswith("g/")}
        worst = grad_errors(g, ref["grads"])
        assert max(worst.values()) < 1e-4, (r, sorted(worst.items(), key=lambda kv: -kv[1])[:3])
  for k in [k for k in got[0].files if k[:2] in ("p/", "m/")]:
    for r in range(1, world):
      assert np.array_equal(got[0][k], got[r][k]), (k, r)
  em = grad_errors({k[2:]: got[0][k] for k in got[0].files if k.startswith("m/")}, opt["m"])
  assert max(em.values()) < 1e-3, sorted(em.items(), key=lambda kv: -kv[1])[:3]


def test_p2p_exchange_gives_up_on_a_dead_peer(tmp_path):
  """Rank 1 leaves after the handles were exchanged; rank 0's step waits SMX_P2P_TIMEOUT_S (2 s here, 30 s by default) for its READY
  flag, gives up, and the STEP reports it (SMX_ERR_COMM from the metrics read-back; the Engine reads and clears the sticky word as it raises)
  -- the device is not left with a spinning kernel and no garbage gradient is applied silently."""
  from sisua_amd import build
  build.build(verbose=False)
  outs = _run(tmp_path, 2, KW, die=1, timeout=120, extra_env=dict(SMX_P2P_TIMEOUT_S="2"))
  assert outs[0][0] == 0 and "DONE" in outs[0][1], outs[0][1][-3000:]
  r0 = np.load(tmp_path / "r0.npz")
  assert int(r0["err"]) != 0 and "timed out" in str(r0["raised"])
  assert int(r0["err_after"]) == 0   # reported once: the Engine cleared the sticky word when it raised


def test_attach_engine_falls_back_to_the_exchange_when_rccl_refuses(tmp_path):
  """parallel.attach_engine with the default SMX_ALLREDUCE: RCCL refuses two ranks on the box's one device on EVERY rank, the
  ranks agree on that over the control plane and carry on over the peer-to-peer exchange -- same results as joining it directly."""
  from sisua_amd import build
  build.build(verbose=False)
  direct = tmp_path / "direct"; direct.mkdir()
  fb = tmp_path / "fallback"; fb.mkdir()
  outs = _run(direct, 2, KW)
  assert all(rc == 0 and "DONE" in out for rc, out in outs), outs[0][1][-2000:]
  outs = _run(fb, 2, KW, extra_env=dict(SMX_TEST_ATTACH="1"))
  assert all(rc == 0 and "DONE" in out for rc, out in outs), outs[0][1][-3000:]
  assert "falling back to SMX_ALLREDUCE=p2p-only" in outs[0][1]
  a, b = np.load(direct / "r0.npz"), np.load(fb / "r0.npz")
  b1 = np.load(fb / "r1.npz")
  for k in a.files:
    np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    if k.startswith(("p/", "m/", "g/")):
      np.testing.assert_array_equal(b[k], b1[k], err_msg=k)
  # and the switch that forbids the fallback keeps the old behaviour: the job stops with RCCL's error
  outs = _run(fb, 2, KW, extra_env=dict(SMX_TEST_ATTACH="1", SMX_NO_COMM_FALLBACK="1"))
  assert all(rc != 0 for rc, _ in outs)


_FIT_CHILD = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import sisua_amd.models as api
from sisua_amd.data import SingleCellOMIC
from tests.util import synth_counts
rank, out = int(os.environ["RANK"]), sys.argv[1]
sco = SingleCellOMIC(synth_counts(640, 120, sparsity=0.8, seed=3), name="toy")
m = api.VAE(outputs=api.RVmeta(120, "zinb", name="transcriptomic"), latents=api.RVmeta(8, "diag", True, "Latents"),
            encoder=api.NetConf([32], batchnorm=True, dropout=0.1), decoder=api.NetConf([32], batchnorm=True, dropout=0.1))
m.fit(sco, epochs=3, batch_size=32, verbose=False, distributed="auto")
rep = m.dp_report
np.savez(out, selected=rep["selected"], us=rep["us_per_step"].get(3) or -1.0, loss=np.asarray(m.train_history["loss"], np.float64),
         **{{"p/" + k: v for k, v in m._engine.get_params().items()}})
print("DONE", flush=True)
"""


def test_fit_between_processes_measures_its_exchange_form(tmp_path):
  """SingleCellModel.fit(distributed='auto') in two fresh processes sharing the box's GPU (RCCL refuses the duplicate device on both: the
  job goes on over the hand-written exchange): the first fit on a communicator measures the exchange form on its own step
  (parallel.calibrate_forms, `model.dp_report`), training then runs as if it had not -- the replicas end bit-identical, the loss falls."""
  from sisua_amd import build
  build.build(verbose=False)
  script = tmp_path / "fit_child.py"
  script.write_text(_FIT_CHILD.format(root=ROOT))
  port = _free_port()
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="1", SMX_CP_PORT=str(port), WORLD_SIZE="2", LOCAL_RANK="0", SMX_RUN_ID=f"fit{port}")
  for k in ("SMX_ALLREDUCE", "SMX_DP_FORM", "SMX_DP_CALIBRATE"):
    env.pop(k, None)
  ps = [subprocess.Popen([sys.executable, str(script), str(tmp_path / f"f{r}.npz")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         env=dict(env, RANK=str(r)), text=True) for r in range(2)]
  for p in ps:
    try:
      out, _ = p.communicate(timeout=240)
    except subprocess.TimeoutExpired:
      for q in ps:
        q.kill()
      pytest.fail("a rank of the two-process fit hangs")
    assert p.returncode == 0 and "DONE" in out, out[-3000:]
  a, b = np.load(tmp_path / "f0.npz"), np.load(tmp_path / "f1.npz")
  assert int(a["selected"]) == int(b["selected"]) == 3 and float(a["us"]) == float(b["us"]) > 0
  for k in [k for k in a.files if k.startswith("p/")]:
    assert np.array_equal(a[k], b[k]), k
  assert np.isfinite(a["loss"]).all() and a["loss"][-1] < a["loss"][0]
