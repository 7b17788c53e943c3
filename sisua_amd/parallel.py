"""Data-parallel plumbing: one process per GPU, cells sharded over ranks, ONE RCCL
all-reduce of the flat gradient buffer per step (inside libsisua_hip.so).

torch.distributed (gloo, CPU) is used only as the rendezvous / control plane:
exchanging the 128-byte RCCL unique id, barriers and max-over-ranks timing.  The
data plane never touches torch.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import numpy as np


def env_rank_world() -> Tuple[int, int, int]:
  return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


class ControlPlane:
  """gloo process group wrapper; a no-op for world == 1."""

  def __init__(self, rank: int, world: int, init_method: Optional[str] = None):
    self.rank, self.world, self.dist = rank, world, None
    if world > 1:
      import torch.distributed as dist
      if not dist.is_initialized():
        kw = dict(backend="gloo", rank=rank, world_size=world)
        if init_method:
          kw["init_method"] = init_method
        dist.init_process_group(**kw)
      self.dist = dist

  def barrier(self):
    if self.dist is not None:
      self.dist.barrier()

  def broadcast_bytes(self, make: Callable[[], bytes], src: int = 0) -> bytes:
    """Rank `src` calls make(); every rank returns the same bytes."""
    if self.dist is None:
      return make()
    box = [make() if self.rank == src else None]
    self.dist.broadcast_object_list(box, src=src)
    return box[0]

  def max(self, value: float) -> float:
    if self.dist is None:
      return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
    return float(t.item())

  def sum_array(self, a: np.ndarray) -> np.ndarray:
    if self.dist is None:
      return a
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
    self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
    return t.numpy()

  def close(self):
    if self.dist is not None and self.dist.is_initialized():
      self.dist.barrier()
      self.dist.destroy_process_group()


class LocalControlPlane:
  """Same interface for `world` replicas living in ONE process as threads on ONE GPU (tests: the data-parallel path
  of SingleCellModel.fit on a one-GPU box).  `LocalControlPlane.group(world)` returns one plane per rank; the engines
  join the library's loopback communicator (smx_comm_init_local) instead of RCCL."""

  class _Shared:

    def __init__(self, world):
      import threading
      self.world = world
      self.barrier = threading.Barrier(world, timeout=120)
      self.box = [None] * world
      self.engines = [None] * world

  def __init__(self, rank: int, shared: "LocalControlPlane._Shared"):
    self.rank, self.world, self._s = rank, shared.world, shared

  @classmethod
  def group(cls, world: int):
    sh = cls._Shared(world)
    return [cls(r, sh) for r in range(world)]

  def barrier(self):
    self._s.barrier.wait()

  def _gather(self, value):
    self._s.box[self.rank] = value
    self._s.barrier.wait()
    vals = list(self._s.box)
    self._s.barrier.wait()
    return vals

  def broadcast_bytes(self, make: Callable[[], bytes], src: int = 0) -> bytes:
    return self._gather(make() if self.rank == src else None)[src]

  def max(self, value: float) -> float:
    return float(max(self._gather(float(value))))

  def sum_array(self, a: np.ndarray) -> np.ndarray:
    return np.sum(self._gather(np.asarray(a, dtype=np.float64)), axis=0)

  def attach(self, engine):
    self._s.engines[self.rank] = engine
    self._s.barrier.wait()
    if self.rank == 0:
      from sisua_amd.engine import Engine
      Engine.comm_init_local(self._s.engines)
    self._s.barrier.wait()

  def close(self):
    pass


def attach_engine(engine, cp):
  """Join the engine to the communicator of the job: RCCL (unique id from rank 0 over the control plane), or the
  library's loopback communicator for an in-process LocalControlPlane."""
  if cp.world <= 1:
    return
  if isinstance(cp, LocalControlPlane):
    cp.attach(engine)
    return
  from sisua_amd.engine import Engine
  uid = cp.broadcast_bytes(Engine.comm_unique_id)
  engine.comm_init(cp.rank, cp.world, uid)
