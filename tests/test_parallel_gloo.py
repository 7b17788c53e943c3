"""Data-parallel path on CPU: world_size 2 over gloo.  Covers the control plane
(sisua_amd/parallel.py) and the DP contract the HIP library implements with one
all-reduce: sum of the per-rank flat buffers [grads | BN batch stats | metrics] / world
== the single-process result on the concatenated minibatch with per-replica BN."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

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
  np.save(os.path.join(out_dir, f"rank{rank}.npy"), tot)
  np.save(os.path.join(out_dir, f"rows{rank}.npy"), rows)
  cp.close()


def test_world2_gloo(tmp_path):
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
