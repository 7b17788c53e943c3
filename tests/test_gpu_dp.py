"""Data-parallel arithmetic of the HIP library with world > 1 on ONE GPU.

`smx_comm_init_local` joins several models of this process into a loopback communicator (events + a summing
kernel in place of RCCL; every model is driven by its own host thread).  Everything else of the step is the
production path: loss scaled by 1 / (batch * world), ONE all-reduce of [grads | BN batch stats | metrics], the
norm of the REDUCED gradient for per-tensor clipnorm, averaged moving statistics, Adam -- and, opt-in,
SyncBatchNorm with its per-layer statistics all-reduce.  Checked against oracle.dp_train_step."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from oracle import sisua_oracle as so
from tests.util import adam_state_errors, grad_errors, make_pair, masked_move_error, perturbed_params, rel_l2, synth_counts, synth_labels

pytestmark = pytest.mark.gpu
RTOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def Engine():
  from sisua_amd import build
  build.build(verbose=False)
  from sisua_amd.engine import Engine
  return Engine


def run_ranks(fns, timeout=120):
  """One host thread per rank (the ctypes calls release the GIL); re-raises the first failure."""
  errs, outs = [None] * len(fns), [None] * len(fns)

  def wrap(i):
    try:
      outs[i] = fns[i]()
    except BaseException as e:  # noqa: BLE001
      errs[i] = e

  ts = [threading.Thread(target=wrap, args=(i,), daemon=True) for i in range(len(fns))]
  for t in ts:
    t.start()
  for t in ts:
    t.join(timeout)
  assert not any(t.is_alive() for t in ts), "a rank thread hangs"
  for e in errs:
    if e is not None:
      raise e
  return outs


CASES = {
    "vae_zinb": dict(model="vae", n_genes=203, likelihood="zinb", enc_units=(48, 40), dec_units=(40,), latent_dim=10),
    "vae_clip": dict(model="vae", n_genes=120, likelihood="nb", enc_units=(32,), dec_units=(32,), latent_dim=6, clipnorm=0.05,
                     lr=5e-3),
    "sisua": dict(model="sisua", n_genes=180, likelihood="zinb", enc_units=(64,), dec_units=(64,), latent_dim=9,
                  labels=((12, "nb"), (7, "onehot"))),
    "scvi_zinbd": dict(model="scvi", n_genes=160, likelihood="zinbd", enc_units=(48,), dec_units=(48,), latent_dim=6,
                       encl_units=(16,)),
    "fvae": dict(model="fvae", n_genes=110, likelihood="zinb", enc_units=(32,), dec_units=(32,), latent_dim=8, disc_units=60, disc_layers=2),
    "vae_nobn": dict(model="vae", n_genes=64, likelihood="nb", enc_units=(32,), dec_units=(32,), latent_dim=5, batchnorm=False),
    # round-3 variants through the same flat-buffer collective: MISA's full-covariance / zero-inflated mixture heads, SCALE's priors
    "misa_mix": dict(model="sisua", n_genes=90, likelihood="zinb", enc_units=(32,), dec_units=(32,), latent_dim=6,
                     labels=((7, "mixtril2"), (5, "mixzinb3")), alpha=10.0),
    "scale": dict(model="scale", n_genes=100, likelihood="zinb", enc_units=(32,), dec_units=(32,), latent_dim=7, n_components=4),
    "scale_tril": dict(model="scale", n_genes=100, likelihood="nb", enc_units=(32,), dec_units=(32,), latent_dim=6, n_components=3, covariance="tril"),
    "scale_post": dict(model="scale", n_genes=100, likelihood="zinb", enc_units=(32,), dec_units=(32,), latent_dim=6, n_components=3, latent_mixture=True),
    # round 4: outputs[1:] as observed heads beside label heads; scvi with per-gene dispersion / inflation vectors and a second output
    "sisua_extra_output": dict(model="sisua", n_genes=120, likelihood="zinb", enc_units=(40,), dec_units=(40,), latent_dim=7,
                               extra_outputs=((9, "nbd"),), labels=((8, "nb"), (5, "onehot")), alpha=10.0),
    "scvi_share_two_outputs": dict(model="scvi", n_genes=130, likelihood="zinbd", enc_units=(40,), dec_units=(40,), latent_dim=5, encl_units=(16,),
                                   dispersion="share", inflation="share", extra_outputs=((7, "zinbd"),)),
}


def _problem(kw, n=400):
  spec, cfg = make_pair(**kw)
  x = synth_counts(n, spec.n_genes, sparsity=0.85, seed=0)
  ys = synth_labels(n, spec.extra_outputs + spec.labels)
  _, lm, lv = so.library_size(x)
  lib = np.tile(np.array([[lm, lv]], dtype=np.float32), (n, 1))
  mask = so.label_mask(n, 0.4, n_omics=1 + len(spec.labels), seed=1)
  return spec, cfg, x, ys, lib, mask


@pytest.mark.parametrize("name,world,sync_bn", [("sisua_extra_output", 2, False), ("sisua_extra_output", 3, True), ("scvi_share_two_outputs", 2, False),
                                                ("scvi_share_two_outputs", 2, True), ("vae_zinb", 2, False), ("vae_zinb", 2, True), ("vae_clip", 2, False),
                                                ("vae_clip", 3, True), ("sisua", 2, False), ("sisua", 2, True),
                                                ("scvi_zinbd", 2, False), ("scvi_zinbd", 2, True), ("vae_nobn", 4, False), ("fvae", 2, False),
                                                ("misa_mix", 2, False), ("scale", 2, True), ("scale_tril", 3, False), ("scale_post", 2, True)])
def test_world_n_steps_match_oracle(Engine, name, world, sync_bn):
  """3 optimiser steps of `world` replicas, every rank holding the WHOLE matrix but drawing its own rows: loss /
  metrics / reduced gradients / gradient norms / parameters / moving statistics of EVERY rank equal the oracle's
  data-parallel contract."""
  spec, cfg, x, ys, lib, mask = _problem(CASES[name])
  params = perturbed_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  B, steps, base = 48, 3, 1000
  engines = []
  for r in range(world):
    e = Engine(cfg, max_batch=64, init=False)
    e.set_params(params)
    e.upload(x, ys, lib, mask, cell_id_base=base)
    engines.append(e)
  Engine.comm_init_local(engines)
  for e in engines:
    assert e.world == world
    e.set_sync_bn(sync_bn)
  assert [e.rank for e in engines] == list(range(world))
  rng = np.random.default_rng(5)
  for step in range(steps):
    rows = rng.permutation(x.shape[0])[: B * world].astype(np.int32).reshape(world, B)
    p_before = {k: v.copy() for k, v in params.items()}
    ref = so.dp_train_step(spec, params, bn, opt, x, list(rows), step, cell_base=base, y=ys, library=lib, mask=mask,
                           sync_bn=sync_bn)
    ms = run_ranks([lambda r=r: engines[r].train_step(rows[r]) for r in range(world)])
    for r, (e, m) in enumerate(zip(engines, ms)):
      assert m["nan_flag"] == 0 and m["step"] == step + 1
      for key in ("loss", "nllk_x", "kl") + (("nllk_y",) if spec.labels else ()) + (("kl_l",) if spec.model == "scvi" else ()) + \
          (("nllk_o",) if spec.extra_outputs else ()) + \
          (("tc", "dtc_loss") if spec.model == "fvae" else ()):   # (fvae: z is permuted within each rank's minibatch)
        assert np.isclose(m[key], ref["metrics"][key], rtol=RTOL, atol=1e-5), (r, step, key, m[key], ref["metrics"][key])
      assert np.isclose(m["grad_norm_max"], max(ref["norms"].values()), rtol=1e-3), (r, step)
      if step == 0:   # the reduced gradient itself (later steps start from fp32-rounded parameters)
        worst = grad_errors(e.get_params(which=1), ref["grads"])
        assert max(worst.values()) < RTOL, (r, sorted(worst.items(), key=lambda kv: -kv[1])[:3])
  # after 3 steps: every rank holds the same parameters and optimiser state, and they are the oracle's.  The moments are
  # what is held to a tight tolerance (m linear, v quadratic in the gradients); the weights are judged where the gradient is
  # far above float32 rounding -- elsewhere Adam turns rounding noise into steps of up to lr
  finals = [e.get_params() for e in engines]
  for k in finals[0]:
    for r in range(1, world):
      assert np.array_equal(finals[0][k], finals[r][k]), (k, r)
  em, ev, where = adam_state_errors(engines[0], opt)
  assert em < 1e-3 and ev < 2e-3, (em, ev, where)
  for k in finals[0]:
    err = masked_move_error(finals[0][k], p_before[k], params[k], ref["grads"][k], spec.lr)
    assert err is None or err < 1e-2, (k, err)
    assert np.abs(finals[0][k] - params[k]).max() <= 3.01 * 1.6 * spec.lr, k   # (3 steps of at most lr_t <= 1.6 lr each: a sanity bound, not the parity check)
  names = [p for p, _ in so.bn_manifest(spec)]
  for e in engines:
    for i, st in e.get_bn().items():
      assert np.allclose(st["moving_mean"], bn[f"{names[i]}/moving_mean"], rtol=1e-4, atol=1e-6)
      assert np.allclose(st["moving_var"], bn[f"{names[i]}/moving_var"], rtol=1e-4, atol=1e-6)
  if name == "vae_clip":
    assert max(ref["norms"].values()) > 10 * spec.clipnorm   # the threshold was exceeded: the norm path is what ran
  for e in engines:
    e.close()


@pytest.mark.parametrize("name,world,sync_bn", [("vae_zinb", 2, False), ("vae_zinb", 3, True), ("sisua", 2, False), ("sisua_extra_output", 3, False),
                                                ("misa_mix", 2, True), ("scale", 2, False)])
def test_two_bucket_chain_matches_oracle(Engine, monkeypatch, name, world, sync_bn):
  """The two-bucket step of round 5 (smx_step.hip: dp_chain_start; forced here with SMX_DP_BUCKETS=2, by default taken from 3 MB of head
  gradients): the heads' bucket is all-reduced, normed and APPLIED on the communication stream -- started where the heads' gradients are
  final (with label heads whose weight gradient rides with the last launch of the backward pass: in front of the optimiser), joined in front of
  the next step's output head --, the front bucket on the model's stream.  Four steps through ONE multi-step call and one single-step call
  (the chain of step n is still under way when step n + 1 starts): losses of every step, the reduced gradients of the last one, parameters,
  Adam moments and moving statistics equal the oracle's data-parallel contract; every rank ends bit-identical; and the one-bucket form of
  the same run gives the same losses to rounding."""
  spec, cfg, x, ys, lib, mask = _problem(CASES[name])
  B, steps, base = 48, 4, 1000
  rng = np.random.default_rng(5)
  rows = [rng.permutation(x.shape[0])[: B * world].astype(np.int32).reshape(world, B) for _ in range(steps + 1)]
  hist = {}
  for buckets in ("2", "1"):
    monkeypatch.setenv("SMX_DP_BUCKETS", buckets)
    params = perturbed_params(spec)
    bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
    engines = []
    for r in range(world):
      e = Engine(cfg, max_batch=64, init=False)
      e.set_params(params)
      e.upload(x, ys, lib, mask, cell_id_base=base)
      engines.append(e)
    Engine.comm_init_local(engines)
    for e in engines:
      e.set_sync_bn(sync_bn)
      assert e.comm_form == int(buckets)
    refs = [so.dp_train_step(spec, params, bn, opt, x, list(rows[s]), s, cell_base=base, y=ys, library=lib, mask=mask, sync_bn=sync_bn) for s in range(steps)]
    orders = [np.concatenate([rows[s][r] for s in range(steps)]) for r in range(world)]
    ms = run_ranks([lambda r=r: engines[r].train_steps(orders[r], steps, B, graph=False, metrics=True) for r in range(world)])
    hs = [e.metrics_history(steps) for e in engines]
    for r in range(world):
      for s in range(steps):
        assert np.isclose(hs[r]["loss"][s], refs[s]["metrics"]["loss"], rtol=RTOL, atol=1e-5), (buckets, r, s)
      assert np.isclose(ms[r]["grad_norm_max"], max(refs[-1]["norms"].values()), rtol=1e-3), (buckets, r)
    em, ev, where = adam_state_errors(engines[0], opt)
    assert em < 2e-3 and ev < 4e-3, (buckets, em, ev, where)
    # one more step as a call of its own: every gradient of it
    p_before = {k: v.copy() for k, v in params.items()}
    ref = so.dp_train_step(spec, params, bn, opt, x, list(rows[steps]), steps, cell_base=base, y=ys, library=lib, mask=mask, sync_bn=sync_bn)
    ms = run_ranks([lambda r=r: engines[r].train_step(rows[steps][r]) for r in range(world)])
    for r, m in enumerate(ms):
      assert m["step"] == steps + 1 and np.isclose(m["loss"], ref["metrics"]["loss"], rtol=RTOL, atol=1e-5), (buckets, r)
    finals = [e.get_params() for e in engines]
    for k in finals[0]:
      for r in range(1, world):
        assert np.array_equal(finals[0][k], finals[r][k]), (buckets, k, r)
    names = [p for p, _ in so.bn_manifest(spec)]
    for i, st in engines[-1].get_bn().items():
      assert np.allclose(st["moving_mean"], bn[f"{names[i]}/moving_mean"], rtol=1e-4, atol=1e-6)
      assert np.allclose(st["moving_var"], bn[f"{names[i]}/moving_var"], rtol=1e-4, atol=1e-6)
    hist[buckets] = np.asarray(hs[0]["loss"])
    for e in engines:
      e.close()
  assert np.allclose(hist["2"], hist["1"], rtol=2e-6)


@pytest.mark.parametrize("name,world,sync_bn", [("vae_zinb", 2, False), ("vae_zinb", 3, True), ("sisua", 2, False), ("sisua_extra_output", 3, False),
                                                ("vae_zinb", 8, False), ("scale", 2, False)])
def test_sharded_optimiser_state_matches_oracle(Engine, name, world, sync_bn):
  """Flag opt_shard (smx_step.hip: dp_chain_start): the heads' bucket is reduce-SCATTERED, every rank clips and applies
  Adam to its 1 / world slice only (slices cut at 64-float boundaries, through optimiser chunks and tensors where they fall; the per-tensor
  norm from the ranks' partial sums of squares), and the updated parameters are all-gathered.  Four steps through one multi-step call + a
  single-step call: losses of every step, the largest gradient norm, the parameters and -- after smx_opt_gather -- both Adam moments equal
  the oracle's data-parallel contract; every rank ends bit-identical, moments included; reading a head's moments before the gather is
  refused; the unsharded chain of the same run gives the same losses to rounding."""
  from sisua_amd._hip import SmxError
  spec, cfg, x, ys, lib, mask = _problem(CASES[name])
  B, steps, base = 24 if world == 8 else 48, 4, 1000
  rng = np.random.default_rng(5)
  rows = [rng.permutation(x.shape[0])[: B * world].astype(np.int32).reshape(world, B) for _ in range(steps + 2)]
  hist = {}
  for shard in (True, False):
    params = perturbed_params(spec)
    bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
    engines = []
    for r in range(world):
      e = Engine(cfg, max_batch=64, init=False)
      e.set_params(params)
      e.upload(x, ys, lib, mask, cell_id_base=base)
      engines.append(e)
    Engine.comm_init_local(engines)
    for e in engines:
      e.set_sync_bn(sync_bn)
      e.set_flag("opt_shard", shard)
      if shard:
        assert e.comm_form == 2   # (the flag takes the chained form whatever the bucket's size)
    refs = [so.dp_train_step(spec, params, bn, opt, x, list(rows[s]), s, cell_base=base, y=ys, library=lib, mask=mask, sync_bn=sync_bn) for s in range(steps)]
    orders = [np.concatenate([rows[s][r] for s in range(steps)]) for r in range(world)]
    ms = run_ranks([lambda r=r: engines[r].train_steps(orders[r], steps, B, graph=False, metrics=True) for r in range(world)])
    hs = [e.metrics_history(steps) for e in engines]
    for r in range(world):
      for s in range(steps):
        assert np.isclose(hs[r]["loss"][s], refs[s]["metrics"]["loss"], rtol=RTOL, atol=1e-5), (shard, r, s)
      assert np.isclose(ms[r]["grad_norm_max"], max(refs[-1]["norms"].values()), rtol=1e-3), (shard, r)
    if shard:
      with pytest.raises(SmxError, match="opt_gather"):
        engines[0].get_params(which=2)
      run_ranks([lambda r=r: engines[r].opt_gather() for r in range(world)])
    em, ev, where = adam_state_errors(engines[world - 1], opt)
    assert em < 2e-3 and ev < 4e-3, (shard, em, ev, where)
    ref = so.dp_train_step(spec, params, bn, opt, x, list(rows[steps]), steps, cell_base=base, y=ys, library=lib, mask=mask, sync_bn=sync_bn)
    ms = run_ranks([lambda r=r: engines[r].train_step(rows[steps][r]) for r in range(world)])
    for r, m in enumerate(ms):
      assert m["step"] == steps + 1 and np.isclose(m["loss"], ref["metrics"]["loss"], rtol=RTOL, atol=1e-5), (shard, r)
    # one more step with the flag switched OFF and no gather by the caller: the library brings the stale moments in before an unsharded step
    for e in engines:
      e.set_flag("opt_shard", False)
    ref = so.dp_train_step(spec, params, bn, opt, x, list(rows[steps + 1]), steps + 1, cell_base=base, y=ys, library=lib, mask=mask, sync_bn=sync_bn)
    ms = run_ranks([lambda r=r: engines[r].train_step(rows[steps + 1][r]) for r in range(world)])
    for r, m in enumerate(ms):
      assert m["step"] == steps + 2 and np.isclose(m["loss"], ref["metrics"]["loss"], rtol=RTOL, atol=1e-5), (shard, r)
    finals = [(e.get_params(), e.get_params(which=2), e.get_params(which=3)) for e in engines]
    em, ev, where = adam_state_errors(engines[0], opt)
    assert em < 2e-3 and ev < 4e-3, (shard, em, ev, where)
    worst = grad_errors(finals[0][0], params)
    assert max(worst.values()) < 1e-4, sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    for which in range(3):
      for k in finals[0][which]:
        for r in range(1, world):
          assert np.array_equal(finals[0][which][k], finals[r][which][k]), (shard, which, k, r)
    hist[shard] = np.asarray(hs[0]["loss"])
    for e in engines:
      e.close()
  assert np.allclose(hist[True], hist[False], rtol=2e-6)


def test_sync_bn_equals_single_process_on_the_global_batch(Engine):
  """SURVEY 8e caveat (i): with SyncBatchNorm two replicas of 32 cells ARE one process on the 64 cells."""
  spec, cfg, x, ys, lib, mask = _problem(CASES["vae_zinb"])
  one = Engine(cfg, max_batch=64)
  one.upload(x, ys, lib, mask)
  two = [Engine(cfg, max_batch=64) for _ in range(2)]
  for e in two:
    e.upload(x, ys, lib, mask)
  Engine.comm_init_local(two)
  for e in two:
    e.set_sync_bn(True)
  rng = np.random.default_rng(0)
  for step in range(4):
    rows = rng.permutation(x.shape[0])[:64].astype(np.int32)
    m1 = one.train_step(rows)
    m2 = run_ranks([lambda r=r: two[r].train_step(rows[r * 32:(r + 1) * 32]) for r in range(2)])
    assert np.isclose(m1["loss"], m2[0]["loss"], rtol=2e-6) and m2[0]["loss"] == m2[1]["loss"]
  a, b = one.get_params(), two[0].get_params()
  for k in a:   # different summation order (2 x 32 vs 64 rows) only
    assert np.allclose(a[k], b[k], rtol=1e-4, atol=2e-5), k
  for e in [one] + two:
    e.close()


def test_k_adam_matches_oracle(Engine):
  """The optimiser kernel by itself (smx_k_adam): per-tensor clipnorm + Keras Adam to 1e-6, several steps, with
  tensors above and below the clip threshold and sizes that are not multiples of the 4096-float chunk."""
  from sisua_amd.engine import k_adam
  rng = np.random.default_rng(0)
  shapes = [(33, 17), (5,), (128, 96), (4097,), (1, 3)]
  spec = so.Spec(model="vae", n_genes=8, enc_units=(4,), dec_units=(4,), latent_dim=2, clipnorm=2.0, lr=3e-3)
  names = [f"t{i}" for i in range(len(shapes))]
  params = {n: rng.normal(size=s) for n, s in zip(names, shapes)}
  params = {k: v.astype(np.float32).astype(np.float64) for k, v in params.items()}
  opt = so.init_opt_state(params)
  gp = [params[n].astype(np.float32) for n in names]
  gm = [np.zeros(s, np.float32) for s in shapes]
  gv = [np.zeros(s, np.float32) for s in shapes]
  for t in range(1, 6):
    scale = [0.01, 3.0, 0.05, 0.2, 10.0]
    grads = {n: (sc * rng.normal(size=s)).astype(np.float32).astype(np.float64) for n, s, sc in zip(names, shapes, scale)}
    before = {n: params[n].copy() for n in names}
    norms = so.adam_update(spec, params, grads, opt)
    gp, gm, gv, gn = k_adam(gp, [grads[n] for n in names], gm, gv, t, lr=spec.lr, beta1=spec.adam_beta1,
                            beta2=spec.adam_beta2, eps=spec.adam_eps, clipnorm=spec.clipnorm)
    assert any(v > spec.clipnorm for v in norms.values()) and any(v < spec.clipnorm for v in norms.values())
    for i, n in enumerate(names):
      assert np.isclose(gn[i], norms[n], rtol=1e-6), (t, n)
      step_ref, step_got = params[n] - before[n], gp[i].astype(np.float64) - before[n]
      # the update itself, not the parameter: rel-L2 of the MOVE (fp32 storage of the parameter: 6e-8 * |p| absolute)
      assert np.linalg.norm(step_got - step_ref) <= 1e-5 * np.linalg.norm(step_ref) + 2e-7 * np.linalg.norm(before[n]), (t, n)
      # fp32 storage: b1 * m + (1 - b1) * g rounds twice, and the two terms may cancel
      assert np.allclose(gm[i], opt["m"][n], rtol=1e-6, atol=3e-7 * np.abs(opt["m"][n]).max()), (t, n)
      # 1 - beta2 in fp32 (0.999f = 0.99900001...) is 1.3e-5 off the float64 value; it cancels against the same factor
      # in lr_t, which is why the MOVE above agrees to 1e-5
      assert np.allclose(gv[i], opt["v"][n], rtol=3e-5, atol=3e-7 * np.abs(opt["v"][n]).max()), (t, n)
    # the oracle continues from the GPU's fp32 state so that the comparison stays per-step
    for i, n in enumerate(names):
      params[n] = gp[i].astype(np.float64); opt["m"][n] = gm[i].astype(np.float64); opt["v"][n] = gv[i].astype(np.float64)


def test_comm_library_is_resolved_beside_the_hip_runtime(Engine):
  info = Engine.comm_library()
  assert os.path.isabs(info["rccl"]) and os.path.exists(info["rccl"]) and "rccl" in os.path.basename(info["rccl"])
  assert os.path.exists(info["hip"]) and "amdhip64" in os.path.basename(info["hip"])
  # deterministic rule: RCCL is the sibling of the HIP runtime the process runs on (unless SMX_RCCL_PATH overrides)
  if not os.environ.get("SMX_RCCL_PATH"):
    assert os.path.dirname(info["rccl"]) == os.path.dirname(info["hip"]), info
  assert info["rccl_version"] > 20000


def test_exchange_form_requests_are_checked(Engine, monkeypatch):
  """smx_comm_set_form refuses what is not attached: a form without a communicator, the hand-written exchange without its peer mapping, a
  form number that does not exist -- and leaves the model as it was."""
  from sisua_amd import SmxError
  from tests.util import make_pair
  monkeypatch.setenv("SMX_FORCE_ALLREDUCE", "1")
  _, cfg = make_pair(model="vae", n_genes=40, likelihood="nb", enc_units=(16,), dec_units=(16,), latent_dim=4)
  e = Engine(cfg, max_batch=16)
  assert e.comm_form == 0
  for form in (1, 2, 3, 7, -1):
    with pytest.raises(SmxError):
      e.comm_set_form(form)
  e.comm_set_form(0)
  e.comm_init(0, 1, Engine.comm_unique_id())
  e.comm_set_form(1)
  assert e.comm_form == 1
  with pytest.raises(SmxError):
    e.comm_set_form(3)          # (no peer mapping)
  assert e.comm_form == 1
  e.comm_p2p_init(0, 1, e.comm_p2p_export(1))
  e.comm_set_form(3)
  assert e.comm_form == 3
  e.comm_set_form(1)
  assert e.comm_form == 1
  e.close()


def test_every_exchange_form_on_one_rank_and_the_measured_choice(Engine, monkeypatch):
  """the step's exchange form is chosen by measurement (parallel.calibrate_forms), not by the 3 MB constant.  One
  rank with SMX_FORCE_ALLREDUCE (the whole data-parallel path runs, the collective moves nothing), RCCL and the hand-written exchange
  both attached, at a wide panel (the chain's head bucket exists): (a) forms 1 / 2 / 3 each report themselves and train the SAME numbers
  as no communicator at all (a world of one: bit for bit); (b) calibrate_forms measures all three, leaves parameters, moments, BatchNorm
  statistics and the step counter exactly as it found them, and sets what it measured fastest; (c) with two forms faked slow
  (SMX_DP_FAKE_SLOW) it takes the third, whichever that is."""
  from sisua_amd import parallel
  from tests.util import make_pair, synth_counts
  monkeypatch.setenv("SMX_FORCE_ALLREDUCE", "1")
  for k in ("SMX_DP_FORM", "SMX_DP_CALIBRATE", "SMX_DP_BUCKETS", "SMX_DP_FAKE_SLOW"):
    monkeypatch.delenv(k, raising=False)
  spec, cfg = make_pair(model="vae", n_genes=4160, likelihood="zinb", enc_units=(128,), dec_units=(128,), latent_dim=16)
  x = synth_counts(512, 4160, sparsity=0.92, seed=3)
  rng = np.random.default_rng(2)
  order = np.concatenate([rng.permutation(512)[:128] for _ in range(40)]).astype(np.int32)

  class Solo:
    rank, world = 0, 1
    def barrier(self): pass
    def max(self, v): return float(v)

  def run(form):
    e = Engine(cfg, max_batch=128, init=False)
    e.set_params(so.init_params(spec))
    e.upload(x, storage="u16")
    if form:
      e.comm_init(0, 1, Engine.comm_unique_id())
      e.comm_p2p_init(0, 1, e.comm_p2p_export(1))
      e.comm_set_form(form)
      assert e.comm_form == form
    e.train_steps(order[:6 * 128], 6, 128, graph=False)
    h = e.metrics_history(6)["loss"].copy()
    if form == 3:
      assert e.comm_p2p_error() == 0
    return e, h, e.get_params(0)

  e0, h0, p0 = run(0)
  e0.close()
  for form in (1, 2, 3):
    e, h, p = run(form)
    assert np.array_equal(h, h0), (form, h, h0)
    for k in p0:
      assert np.array_equal(p[k], p0[k]), (form, k)
    if form < 3:
      e.close()
  # e: forms 1-3 attached, six steps in
  before = e.snapshot()
  rep = parallel.calibrate_forms(e, Solo(), "auto", order, 128, steps=12, warmup=2)
  assert set(rep["us_per_step"]) == {1, 2, 3} and all(v and v > 0 for v in rep["us_per_step"].values()), rep
  assert e.comm_form == rep["selected"]
  after = e.snapshot()
  assert after["step"] == before["step"]
  for part in ("params", "m", "v"):
    for k in before[part]:
      assert np.array_equal(before[part][k], after[part][k]), (part, k)
  for i in before["bn"]:
    for k in before["bn"][i]:
      assert np.array_equal(before["bn"][i][k], after["bn"][i][k])
  for slow, expect in (("1:9000,2:9000", 3), ("1:9000,3:9000", 2), ("2:9000,3:9000", 1)):
    monkeypatch.setenv("SMX_DP_FAKE_SLOW", slow)
    rep = parallel.calibrate_forms(e, Solo(), "auto", order, 128, steps=4, warmup=1)
    assert rep["selected"] == expect and e.comm_form == expect, (slow, rep)
  monkeypatch.setenv("SMX_DP_CALIBRATE", "0")
  assert parallel.calibrate_forms(e, Solo(), "auto", order, 128)["us_per_step"] == {}
  e.close()


_BAD_COMM = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import numpy as np
from sisua_amd import SmxError
from sisua_amd.config import ModelConfig
from sisua_amd.engine import Engine
rank = int(sys.argv[1]); uid_file = sys.argv[2]
e = Engine(ModelConfig(n_genes=40, enc_units=(16,), dec_units=(16,), latent_dim=4), max_batch=16)
if rank == 0:
  uid = Engine.comm_unique_id()
  open(uid_file + ".tmp", "wb").write(uid); os.replace(uid_file + ".tmp", uid_file)
else:
  t0 = time.time()
  while not os.path.exists(uid_file):
    assert time.time() - t0 < 60
    time.sleep(0.05)
  uid = open(uid_file, "rb").read()
try:
  e.comm_init(rank, 2, uid)     # two ranks on ONE device: RCCL refuses (duplicate GPU)
  print("JOINED", flush=True)
except SmxError as err:
  assert "-4" in str(err) or "ncclCommInitRank" in str(err), err
  assert e.world == 1           # left without a communicator
  x = np.ones((32, 40), np.float32)
  e.upload(x)
  m = e.train_step(np.arange(16, dtype=np.int32))   # and still usable as a single-GPU model
  assert m["step"] == 1 and m["nan_flag"] == 0
  print("REFUSED", flush=True)
e.close()
print("CLEAN", flush=True)
"""


def test_comm_init_failure_leaves_a_clean_model(Engine, tmp_path):
  """a failed ncclCommInitRank used to leave a half-built handle that smx_model_destroy then
  freed again (double free).  Two processes on the one GPU of the box: RCCL refuses the duplicate device; both
  processes must report SMX_ERR_COMM, keep working with world 1 and exit cleanly."""
  script = tmp_path / "bad_comm.py"
  script.write_text(_BAD_COMM.format(root=ROOT))
  uid_file = str(tmp_path / "uid.bin")
  env = dict(os.environ, NCCL_DEBUG="WARN")
  ps = [subprocess.Popen([sys.executable, str(script), str(r), uid_file], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         env=env, text=True) for r in range(2)]
  outs = []
  for p in ps:
    try:
      out, _ = p.communicate(timeout=120)
    except subprocess.TimeoutExpired:
      for q in ps:
        q.kill()
      pytest.fail("comm_init on a duplicate device hangs")
    outs.append((p.returncode, out))
  for rc, out in outs:
    assert rc == 0 and "CLEAN" in out, out[-2000:]
  if not all("REFUSED" in out for _, out in outs):   # an RCCL build that accepts two ranks on one device
    assert all("JOINED" in out for _, out in outs), outs


# ---- world 8 at the BASELINE.json splits -------------------------------------------------------
def _baseline_split(workload, world):
  """bench.py's workload cut into `world` contiguous shards exactly as `bench.py --gpus world` / fit(distributed) cut it."""
  import bench
  full = bench.build_workload(0, 1, workload)
  return full, [bench.build_workload(r, world, workload) for r in range(world)]


@pytest.mark.parametrize("sync_bn", [False, True])
def test_world8_c4_split_matches_oracle(Engine, sync_bn):
  """BASELINE configs[3] as the 8-GPU job runs it: eccly-shaped 2116 x 2000 training cells in 8 contiguous shards of 264,
  SISUA zinb + 38 ADT nb labels at 10 %, alpha = 10, global batch 256 = 8 ranks x 32 cells; two optimiser steps on the
  loopback communicator against oracle.dp_train_step on the UNSHARDED matrix (the noise of a cell is keyed by its global id)."""
  world, B, base = 8, 32, 0
  (cfg, xt, batch, extra), shards = _baseline_split("eccly-sisua", world)
  assert batch == 256 and cfg.labels == ((38, "nb"),) and cfg.alpha == 10.0 and xt.shape[1] == 2000
  spec = so.Spec(**cfg.to_dict())
  params = so.init_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  engines, los = [], []
  for r, (c, xs, _, ex) in enumerate(shards):
    lo = ex["cell_id_base"]
    assert xs.shape[0] == xt.shape[0] // world and lo == r * xs.shape[0] and np.array_equal(xs, xt[lo:lo + xs.shape[0]])
    e = Engine(c, max_batch=B, init=False)
    e.set_params(params)
    e.upload(xs, ex["labels"], None, ex["label_mask"], cell_id_base=lo)
    engines.append(e); los.append(lo)
  Engine.comm_init_local(engines)
  for e in engines:
    e.set_sync_bn(sync_bn)
  rng = np.random.default_rng(8)
  n_lab = 0
  for step in range(2):
    local = [rng.permutation(shards[r][1].shape[0])[:B].astype(np.int32) for r in range(world)]
    glob = [local[r] + los[r] for r in range(world)]
    n_lab += int(sum(extra["label_mask"][g].sum() for g in glob))
    ref = so.dp_train_step(spec, params, bn, opt, xt, glob, step, cell_base=base, y=extra["labels"], mask=extra["label_mask"], sync_bn=sync_bn)
    ms = run_ranks([lambda r=r: engines[r].train_step(local[r]) for r in range(world)])
    for r, m in enumerate(ms):
      assert m["nan_flag"] == 0 and m["step"] == step + 1
      for key in ("loss", "nllk_x", "nllk_y", "kl"):
        assert np.isclose(m[key], ref["metrics"][key], rtol=RTOL, atol=1e-5), (r, step, key, m[key], ref["metrics"][key])
      assert m["loss"] == ms[0]["loss"]
    if step == 0:
      for r in (0, 7):
        worst = grad_errors(engines[r].get_params(which=1), ref["grads"])
        assert max(worst.values()) < RTOL, (r, sorted(worst.items(), key=lambda kv: -kv[1])[:3])
    em, ev, where = adam_state_errors(engines[0], opt)
    assert em < (2e-4 if step == 0 else 1e-3) and ev < (4e-4 if step == 0 else 2e-3), (step, em, ev, where)
  assert n_lab > 0   # labelled cells took part: the alpha = 10 term is in the gradients
  a, b = engines[0].get_params(), engines[7].get_params()
  for k in a:
    assert np.array_equal(a[k], b[k]), k
  names = [p for p, _ in so.bn_manifest(spec)]
  for i, st in engines[3].get_bn().items():
    assert np.allclose(st["moving_mean"], bn[f"{names[i]}/moving_mean"], rtol=1e-4, atol=1e-6)
    assert np.allclose(st["moving_var"], bn[f"{names[i]}/moving_var"], rtol=1e-4, atol=1e-6)
  for e in engines:
    e.close()


@pytest.mark.parametrize("sync_bn,shard", [(False, False), (False, True)])   # (bench.py's mode; the C4 split above runs both BatchNorm forms -- at this width the oracle's eight ranks cost 5 s each way)
def test_world8_c5_split_matches_oracle(Engine, sync_bn, shard):
  """BASELINE configs[4]'s per-step shape: global batch 1024 = 8 ranks x 128 cells x 20 000 genes (a 256-cell slice resident
  per rank, uint16 store, each rank's cells generated from (seed, rank) as bench.py's c5-shard does); one optimiser step against
  oracle.dp_train_step: ELBO scalars, every reduced gradient, gradient norms, Adam moments.  `shard`: the heads' optimiser state
  sharded over the eight ranks (flag opt_shard: 1 / 8 of the 7.7 M head parameters updated per rank, then gathered)."""
  import bench
  world, B, per = 8, 128, 256
  shards = [bench.build_workload(r, world, "c5-shard", n_cells=per) for r in range(world)]   # (only the resident slice is drawn)
  cfg = shards[0][0]
  assert cfg.n_genes == 20000 and shards[0][2] == B
  spec = so.Spec(**cfg.to_dict())
  params = so.init_params(spec)
  bn, opt = so.init_bn_state(spec), so.init_opt_state(params)
  xs = [s[1][:per] for s in shards]
  assert not np.array_equal(xs[0], xs[1])
  xt = np.concatenate(xs, 0)
  engines = []
  for r in range(world):
    e = Engine(cfg, max_batch=B, init=False)
    e.set_params(params)
    e.upload(xs[r], cell_id_base=r * per, storage="u16")
    engines.append(e)
  Engine.comm_init_local(engines)
  for e in engines:
    e.set_sync_bn(sync_bn)
    e.set_flag("opt_shard", shard)
    assert e.comm_form == 2   # (30 MB of head gradients: the two-bucket chain, started behind the one-launch output head)
  rng = np.random.default_rng(5)
  local = [rng.permutation(per)[:B].astype(np.int32) for r in range(world)]
  glob = [local[r] + r * per for r in range(world)]
  ref = so.dp_train_step(spec, params, bn, opt, xt, glob, 0, cell_base=0, sync_bn=sync_bn)
  ms = run_ranks([lambda r=r: engines[r].train_step(local[r]) for r in range(world)], timeout=300)
  for r, m in enumerate(ms):
    assert m["nan_flag"] == 0
    for key in ("loss", "nllk_x", "kl"):
      assert np.isclose(m[key], ref["metrics"][key], rtol=RTOL, atol=1e-5), (r, key, m[key], ref["metrics"][key])
    assert np.isclose(m["grad_norm_max"], max(ref["norms"].values()), rtol=1e-3), r
  for r in (0, 5):
    worst = grad_errors(engines[r].get_params(which=1), ref["grads"])
    if shard:   # (the heads' REDUCED gradient exists slice by slice over the ranks: a rank's buffer holds its own slice of it)
      worst = {k: v for k, v in worst.items() if not k.startswith(("out", "lab"))}
    assert max(worst.values()) < RTOL, (r, sorted(worst.items(), key=lambda kv: -kv[1])[:3])
  if shard:
    run_ranks([lambda r=r: engines[r].opt_gather() for r in range(world)])
    p0, p7 = engines[0].get_params(), engines[7].get_params()
    assert all(np.array_equal(p0[k], p7[k]) for k in p0)
  em, ev, where = adam_state_errors(engines[2], opt)
  assert em < 2e-4 and ev < 4e-4, (em, ev, where)
  for e in engines:
    e.close()


def test_rccl_banner_does_not_reach_stdout(tmp_path):
  """Creating an RCCL communicator prints RCCL's banner with printf to STDOUT ("RCCL version : ..."); the job's stdout is a contract
  (bench.py: rank 0 prints ONE JSON line).  parallel.stdout_to_stderr around the calls keeps stdout clean -- checked with a real
  1-rank communicator in a fresh process (the banner appears once per process)."""
  import subprocess, sys
  code = ("import sys\nsys.path.insert(0, %r)\nfrom sisua_amd.engine import Engine\nfrom sisua_amd.parallel import stdout_to_stderr\n"
          "from tests.util import make_pair\nspec, cfg = make_pair(model='vae', n_genes=40, likelihood='nb', enc_units=(16,), dec_units=(16,), latent_dim=4)\n"
          "e = Engine(cfg, max_batch=32)\nwith stdout_to_stderr():\n  e.comm_init(0, 1, Engine.comm_unique_id())\nprint('{\"ok\": 1}')\ne.close()\n" % ROOT)
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=180)
  assert r.returncode == 0, r.stderr[-2000:]
  assert r.stdout.strip() == '{"ok": 1}', r.stdout


def test_bench_line_at_two_ranks_carries_every_scaling_mode(tmp_path):
  """The driver's launch line for N > 1, rehearsed with 2 ranks sharing the box's one GPU (SMX_SHARE_GPU=1, the hand-written
  exchange: RCCL refuses two ranks on one device): rank 0 prints ONE JSON line whose `scaling_modes` holds weak, strong +
  SyncBatchNorm (the reference's global batch and arithmetic) and the C5 share, each with the collective alone, the
  no-communicator step and their difference, beside DESIGN.md section 5's predictions for N = 8.  (The NUMBERS mean nothing here --
  the ranks time-slice one device; the fields are what is asserted.)  torch.distributed.run is a fresh child process."""
  import json
  import socket
  s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
  env = dict(os.environ, SMX_SHARE_GPU="1", SMX_ALLREDUCE="p2p-only", PYTHONPATH=ROOT)
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--c5-cells", "4096"]
  r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
  assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
  lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
  assert len(lines) == 1, r.stdout[-2000:]
  out = json.loads(lines[0])
  assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5 and out["unit"] == "cells/s" and out["value"] > 0
  assert out["scaling"] == "weak" and out["config"]["global_batch"] == 256 and out["dp"]["collective"] == "p2p-only"
  # the exchange form is a measured choice (here the hand-written exchange is the only form two ranks on ONE device have)
  dp = out["dp"]
  assert dp["selected"] == dp["exchange"] == "hand-written exchange" and dp["forms_us_per_step"]["hand-written exchange"] > 0 and dp["value_is"]
  assert out["roofline"]["frac"] > 0 and "cpu_baseline" not in out     # (the CPU baseline is an N = 1 item)
  # the longer run beside the contract's K steps: 300 more steps of the same engine, never `value`
  assert out["value_300"] > 0 and np.isclose(out["value_300"], 256 / (out["ms_per_step_300"] * 1e-3), rtol=1e-3)
  sm = out["scaling_modes"]
  assert set(sm) == {"weak", "strong_syncbn", "c5"}
  expect = {"weak": (128, 256, False), "strong_syncbn": (64, 128, True), "c5": (128, 256, False)}
  for k, (bpg, gb, sbn) in expect.items():
    v = sm[k]
    assert (v["batch_per_gpu"], v["global_batch"], v["sync_bn"]) == (bpg, gb, sbn), (k, v)
    assert v["collective"] == "p2p-only" and v["allreduce_us"] > 0 and v["allreduce_bytes"] > 0
    assert v["exchange"] == "hand-written exchange" and v["exchange_forms_us_per_step"]["hand-written exchange"] > 0
    assert v["ms_per_step"] > 0 and v["nocomm_ms_per_step"] > 0 and v["cells_per_s"] > 0 and np.isfinite(v["final_loss"])
    assert np.isclose(v["dp_overhead_us"], 1e3 * (v["ms_per_step"] - v["nocomm_ms_per_step"]), atol=0.2)
    assert np.isclose(v["cells_per_s"], gb / (v["ms_per_step"] * 1e-3), rtol=1e-3)
    p = v["predicted_n8"]
    assert p["ms_per_step"][0] < p["ms_per_step"][1] and p["x_one_gpu"][0] < p["x_one_gpu"][1] and p["reading"]
  assert sm["c5"]["cells_resident_per_gpu"] == 2048 and sm["c5"]["allreduce_bytes"] > 9 * sm["weak"]["allreduce_bytes"]


def test_bench_line_survives_scaling_modes_that_do_not_finish(tmp_path):
  """The scaling modes run LAST behind a watchdog on rank 0 (bench.py): with a budget they cannot meet (SMX_BENCH_MODES_BUDGET_S) the
  contract's line -- value, ms_per_step, roofline, dp -- is still printed, once, with the modes marked as unfinished."""
  import json
  import socket
  s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
  env = dict(os.environ, SMX_SHARE_GPU="1", SMX_ALLREDUCE="p2p-only", PYTHONPATH=ROOT, SMX_BENCH_MODES_BUDGET_S="0.05")
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--c5-cells", "4096"]
  r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
  lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
  assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
  out = json.loads(lines[0])
  assert out["n_gpus"] == 2 and out["value"] > 0 and out["roofline"]["frac"] > 0 and out["dp"]["collective"] == "p2p-only"
  assert "error" in out["scaling_modes"]

