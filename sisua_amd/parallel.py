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


def attach_engine(engine, cp: ControlPlane):
  """Join the engine to the RCCL communicator of the job (unique id from rank 0)."""
  if cp.world <= 1:
    return
  from sisua_amd.engine import Engine
  uid = cp.broadcast_bytes(Engine.comm_unique_id)
  engine.comm_init(cp.rank, cp.world, uid)
