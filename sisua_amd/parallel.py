"""Data-parallel plumbing: one process per GPU, cells sharded over ranks, ONE all-reduce of the flat gradient
buffer per step (inside libsisua_hip.so: RCCL over xGMI).

The control plane -- exchanging the 128-byte communicator id, barriers, max-over-ranks timing, the validation loss
averaged over the ranks -- is a ~100-line TCP star in this file (rank 0 serves, every collective is gather -> reduce ->
reply): no torch, no MPI.  The launcher's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, as
`python -m torch.distributed.run` or any other launcher sets them) is all it reads.  The data plane never touches it.
"""
from __future__ import annotations

import contextlib
import hashlib
import os
import socket
import struct
import sys
import tempfile
import time
from typing import Callable, List, Optional, Tuple

import numpy as np


def env_rank_world() -> Tuple[int, int, int]:
  return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


_MAGIC = b"SMXCP1"
_OP_BARRIER, _OP_BCAST, _OP_MAX, _OP_SUM, _OP_CLOSE, _OP_GATHER = range(6)


def _send(sock: socket.socket, op: int, payload: bytes = b""):
  sock.sendall(struct.pack("<BQ", op, len(payload)) + payload)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
  buf = bytearray()
  while len(buf) < n:
    chunk = sock.recv(n - len(buf))
    if not chunk:
      raise ConnectionError("control plane: peer closed the connection")
    buf += chunk
  return bytes(buf)


_MAX_PAYLOAD = 256 << 20   # the plane moves unique ids, IPC handles and a few scalars: anything larger is not a peer of this job


def _recv(sock: socket.socket) -> Tuple[int, bytes]:
  op, n = struct.unpack("<BQ", _recv_exact(sock, 9))
  if n > _MAX_PAYLOAD:
    raise ConnectionError(f"control plane: a payload of {n} bytes was announced (limit {_MAX_PAYLOAD})")
  return op, _recv_exact(sock, n) if n else b""


def _is_local(addr: str) -> bool:
  try:
    ip = socket.gethostbyname(addr)
  except OSError:
    return False
  if ip.startswith("127."):
    return True
  # an address of this host is one a socket can be BOUND to (comparing with getaddrinfo(gethostname()) misjudges hosts whose
  # name maps to 127.0.1.1 -- the Debian / Ubuntu default -- when MASTER_ADDR is the NIC's address or the FQDN)
  fam = socket.AF_INET6 if ":" in ip else socket.AF_INET
  try:
    with socket.socket(fam, socket.SOCK_STREAM) as probe:
      probe.bind((ip, 0))
    return True
  except OSError:
    pass
  try:
    return ip in {ai[4][0] for ai in socket.getaddrinfo(socket.gethostname(), None)}
  except OSError:
    return False


class ControlPlane:
  """TCP star over the ranks of one job; a no-op for world == 1.

  Rendezvous: rank 0 listens on SMX_CP_PORT at MASTER_ADDR when that variable is set (any topology); otherwise on an
  ephemeral loopback port which it publishes in a file keyed by (MASTER_ADDR, MASTER_PORT, run id, uid) under the
  temporary directory -- one node, which is what `bench.py --gpus N` and fit(distributed='auto') run on.  (MASTER_PORT
  itself belongs to the launcher: torch.distributed.run keeps its own store there.)  Every rank proves the job's token
  in its hello, so a stale file or a foreign listener is retried, not trusted."""

  def __init__(self, rank: int, world: int, addr: Optional[str] = None, port: Optional[int] = None, timeout: float = 120.0):
    self.rank, self.world = int(rank), int(world)
    self._peers: List[socket.socket] = []   # rank 0: socket of rank r at index r - 1
    self._root: Optional[socket.socket] = None
    self._file = None
    if self.world <= 1:
      return
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    mport = os.environ.get("MASTER_PORT", "0")
    run = os.environ.get("TORCHELASTIC_RUN_ID", os.environ.get("SMX_RUN_ID", "none"))
    # the job's token: public values + (fixed port) a launcher-provided SMX_CP_SECRET or (rendezvous file) 16 random bytes rank 0
    # writes into a file only this user can read -- so another local user can neither join as a rank nor learn the token
    secret = os.environ.get("SMX_CP_SECRET", "")
    base = f"{addr}:{mport}:{run}:{self.world}"
    token = hashlib.sha256(f"{base}:{secret}".encode()).digest()[:16]
    fixed = port if port is not None else (int(os.environ["SMX_CP_PORT"]) if os.environ.get("SMX_CP_PORT") else None)
    if not fixed and not _is_local(addr):
      raise RuntimeError(f"control plane: MASTER_ADDR={addr} is not this host and SMX_CP_PORT is unset -- the rendezvous file only serves "
                         "one node; set SMX_CP_PORT (and SMX_CP_SECRET) on every rank for a multi-node job")
    rdir = os.path.join(tempfile.gettempdir(), f"smx_cp_{os.getuid()}")
    os.makedirs(rdir, mode=0o700, exist_ok=True)
    st = os.stat(rdir)
    if st.st_uid != os.getuid() or (st.st_mode & 0o077):
      raise RuntimeError(f"control plane: {rdir} is not a private directory of this user")
    rfile = os.path.join(rdir, f"{hashlib.sha256(base.encode()).hexdigest()[:16]}.port")
    deadline = time.time() + timeout
    if self.rank == 0:
      srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
      srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
      srv.bind((addr if fixed else "127.0.0.1", fixed or 0))
      srv.listen(self.world)
      if not fixed:
        rnd = os.urandom(16).hex()
        token = hashlib.sha256(f"{base}:{secret}:{rnd}".encode()).digest()[:16]
        tmp = f"{rfile}.{os.getpid()}"
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "w") as f:
          f.write(f"{srv.getsockname()[1]} {rnd}")
        os.replace(tmp, rfile)   # atomic: a reader sees the old file or the new one
        self._file = rfile
      got = {}
      srv.settimeout(1.0)
      while len(got) < self.world - 1:
        if time.time() > deadline:
          raise TimeoutError(f"control plane: {self.world - 1 - len(got)} rank(s) did not join within {timeout:.0f} s")
        try:
          c, _ = srv.accept()
        except socket.timeout:
          continue
        try:
          c.settimeout(10.0)
          hello = _recv_exact(c, len(_MAGIC) + 16 + 4)
          r = struct.unpack("<I", hello[-4:])[0]
          if hello[:len(_MAGIC)] != _MAGIC or hello[len(_MAGIC):-4] != token or not 0 < r < self.world or r in got:
            c.close()
            continue
          c.sendall(_MAGIC)
          c.settimeout(None)
          c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
          got[r] = c
        except (OSError, ConnectionError, struct.error):
          c.close()
      srv.close()
      self._peers = [got[r] for r in range(1, self.world)]
    else:
      while True:
        if time.time() > deadline:
          raise TimeoutError(f"control plane: rank {self.rank} could not reach rank 0 within {timeout:.0f} s")
        try:
          if fixed:
            p, tok = fixed, token
          else:
            fields = open(rfile).read().split()
            p, tok = int(fields[0]), hashlib.sha256(f"{base}:{secret}:{fields[1]}".encode()).digest()[:16]
          c = socket.create_connection((addr if fixed else "127.0.0.1", p), timeout=5.0)
          c.sendall(_MAGIC + tok + struct.pack("<I", self.rank))
          if _recv_exact(c, len(_MAGIC)) != _MAGIC:
            raise ConnectionError("bad reply")
          c.settimeout(None)
          c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
          self._root = c
          break
        except (OSError, ValueError, IndexError, ConnectionError):
          time.sleep(0.05)   # no file yet, a stale file of an earlier job, or rank 0 not listening yet

  # every collective: the ranks send (op, payload) to rank 0, which reduces and replies to each
  def _collective(self, op: int, payload: bytes, reduce: Callable[[List[bytes]], bytes]) -> bytes:
    if self.world <= 1:
      return reduce([payload])
    if self.rank == 0:
      parts = [payload]
      for c in self._peers:
        o, data = _recv(c)
        if o != op:
          raise RuntimeError(f"control plane: ranks disagree on the collective ({o} vs {op})")
        parts.append(data)
      out = reduce(parts)
      for c in self._peers:
        _send(c, op, out)
      return out
    _send(self._root, op, payload)
    o, data = _recv(self._root)
    if o != op:
      raise RuntimeError(f"control plane: ranks disagree on the collective ({o} vs {op})")
    return data

  def barrier(self):
    self._collective(_OP_BARRIER, b"", lambda parts: b"")

  def broadcast_bytes(self, make: Callable[[], bytes], src: int = 0) -> bytes:
    """Rank `src` calls make(); every rank returns the same bytes."""
    mine = bytes(make()) if self.rank == src else b""
    return self._collective(_OP_BCAST, mine, lambda parts: parts[src])

  def allgather_bytes(self, mine: bytes) -> List[bytes]:
    """Every rank's equal-length byte string, in rank order, on every rank."""
    mine = bytes(mine)
    out = self._collective(_OP_GATHER, mine, lambda parts: b"".join(parts))
    n = len(mine)
    return [out[i * n:(i + 1) * n] for i in range(max(self.world, 1))]

  def max(self, value: float) -> float:
    out = self._collective(_OP_MAX, struct.pack("<d", float(value)), lambda parts: struct.pack("<d", max(struct.unpack("<d", p)[0] for p in parts)))
    return struct.unpack("<d", out)[0]

  def sum_array(self, a: np.ndarray) -> np.ndarray:
    """Element-wise float64 sum over the ranks, in rank order on rank 0 (every rank receives the same bits)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    if self.world <= 1:
      return a

    def reduce(parts):
      acc = np.frombuffer(parts[0], np.float64).copy()
      for p in parts[1:]:
        acc += np.frombuffer(p, np.float64)
      return acc.tobytes()
    return np.frombuffer(self._collective(_OP_SUM, a.tobytes(), reduce), np.float64).reshape(a.shape).copy()

  def close(self):
    if self.world > 1 and (self._peers or self._root):
      try:
        self._collective(_OP_CLOSE, b"", lambda parts: b"")
      except (OSError, ConnectionError, RuntimeError):
        pass
    for c in self._peers + ([self._root] if self._root else []):
      try:
        c.close()
      except OSError:
        pass
    self._peers, self._root = [], None
    if self._file:
      try:
        os.unlink(self._file)
      except OSError:
        pass
      self._file = None


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

  def allgather_bytes(self, mine: bytes) -> List[bytes]:
    return [bytes(b) for b in self._gather(bytes(mine))]

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


@contextlib.contextmanager
def stdout_to_stderr():
  """File descriptor 1 points at stderr inside the block (C stdio flushed on both sides): RCCL prints a five-line banner
  ("RCCL version : ...") with printf to STDOUT when a communicator is created, and a job's stdout may be a contract of its own
  (bench.py: rank 0 prints ONE JSON line).  SMX_KEEP_RCCL_BANNER=1 leaves stdout alone."""
  if os.environ.get("SMX_KEEP_RCCL_BANNER"):
    yield
    return
  import ctypes
  libc = ctypes.CDLL(None)
  sys.stdout.flush()
  libc.fflush(None)
  saved = os.dup(1)
  os.dup2(2, 1)
  try:
    yield
  finally:
    sys.stdout.flush()
    libc.fflush(None)
    os.dup2(saved, 1)
    os.close(saved)


def attach_engine(engine, cp):
  """Join the engine to the communicator of the job: RCCL (unique id from rank 0 over the control plane) and, beside it where HIP IPC allows,
  the library's hand-written exchange -- or the library's loopback communicator for an in-process LocalControlPlane.  Which of the attached
  collectives the steps take is the exchange FORM (Engine.comm_form): SMX_DP_FORM=1|2|3 if set, else ONE all-reduce of the flat buffer (the
  north star's wording, and the only form a multi-GPU node has ever run) until `calibrate_forms` below has measured the alternatives on the
  job's own steps."""
  if cp.world <= 1:
    return "none"
  if isinstance(cp, LocalControlPlane):
    cp.attach(engine)
    return "loopback"
  from sisua_amd.engine import Engine
  # 'auto' (default): RCCL, with the hand-written two-shot exchange over IPC-mapped peer buffers attached beside it when every rank can map
  # its peers | 'rccl': RCCL alone | 'p2p': both, the exchange taking the steps (the library's own rule) | 'p2p-only': no RCCL communicator
  mode = os.environ.get("SMX_ALLREDUCE", "auto").lower()
  if mode not in ("auto", "rccl", "p2p", "p2p-only"):
    raise ValueError("SMX_ALLREDUCE must be 'auto', 'rccl', 'p2p' or 'p2p-only'")
  if mode != "p2p-only":
    def unique_id():
      try:
        with stdout_to_stderr():
          return Engine.comm_unique_id()
      except Exception as e:   # (librccl did not load: the peers must still get an answer)
        return b"!" + str(e).encode()
    uid = cp.broadcast_bytes(unique_id)
    err = None
    try:
      if uid[:1] == b"!" and len(uid) != 128:
        raise RuntimeError(uid[1:].decode(errors="replace"))
      with stdout_to_stderr():
        engine.comm_init(cp.rank, cp.world, uid)
    except Exception as e:   # (a failed ncclCommInitRank leaves the model without a communicator: smx_comm.hip)
      err = e
    failed = [b == b"1" for b in cp.allgather_bytes(b"1" if err is not None else b"0")]
    if any(failed):
      # every rank failed the same way (library mismatch, a refused device set): the job goes on over the hand-written exchange,
      # which needs nothing but HIP IPC; a communicator that came up on SOME ranks only cannot be repaired from here
      if not all(failed) or os.environ.get("SMX_NO_COMM_FALLBACK"):
        raise err if err is not None else RuntimeError("RCCL communicator: ranks " + str([i for i, f in enumerate(failed) if f]) + " failed to join")
      if cp.rank == 0:
        import warnings
        warnings.warn(f"RCCL communicator unavailable ({err}); falling back to SMX_ALLREDUCE=p2p-only")
      mode = "p2p-only"
  if mode != "rccl":
    # (every rank takes part in both gathers whatever happens to it: a rank whose export or mapping fails says so instead of leaving)
    try:
      mine, perr = engine.comm_p2p_export(cp.world), None
    except Exception as e:
      mine, perr = b"", e
    handles = cp.allgather_bytes(mine)
    if perr is None and all(len(h) == 128 for h in handles):
      try:
        engine.comm_p2p_init(cp.rank, cp.world, b"".join(handles))
      except Exception as e:
        perr = e
    elif perr is None:
      perr = RuntimeError("a peer could not export its buffers")
    bad = [b == b"1" for b in cp.allgather_bytes(b"1" if perr is not None else b"0")]
    if any(bad):
      if mode != "auto":   # asked for by name: its absence is an error
        raise perr if perr is not None else RuntimeError("hand-written exchange: ranks " + str([i for i, f in enumerate(bad) if f]) + " could not map their peers")
      mode = "rccl"        # (ranks that did map their peers keep the mapping and never use it: no form below selects it)
  if os.environ.get("SMX_OPT_SHARD", "0") not in ("", "0"):   # opt-in: the heads' optimiser state sharded over the ranks (RCCL only; smx_opt_gather)
    engine.set_flag("opt_shard", True)
  forced = os.environ.get("SMX_DP_FORM", "").strip()
  if forced:
    if forced not in ("1", "2", "3") or (forced == "3" and mode == "rccl") or (forced != "3" and mode == "p2p-only"):
      raise ValueError(f"SMX_DP_FORM={forced} is not available with SMX_ALLREDUCE={mode}")
    engine.comm_set_form(int(forced))
  elif mode in ("auto", "rccl"):
    engine.comm_set_form(1)
  # ('p2p' / 'p2p-only': the library's own rule, as in rounds 3-5: the exchange takes the steps)
  return mode   # what is attached: 'rccl', 'auto' (RCCL + the exchange), 'p2p' (both, the exchange first) or 'p2p-only'


def dp_forms_available(engine, mode: str):
  """The exchange forms `calibrate_forms` may try for what attach_engine attached (Engine.comm_form's numbering)."""
  return {"loopback": [1, 2], "rccl": [1, 2], "auto": [1, 2, 3], "p2p": [1, 2, 3], "p2p-only": [3]}.get(mode, [])


def calibrate_forms(engine, cp, mode, order, batch, steps: int = 30, warmup: int = 5, forms=None):
  """Measure every available exchange form on the job's OWN training step and keep the fastest (the form used to be
  picked by a guessed byte count that BASELINE configs[1] sits right on).  Collective: every rank calls it with the same arguments, between
  training calls, data uploaded.  Per form: `warmup` + `steps` steps over `order` (row ids of this rank's shard, >= (warmup + steps) * batch
  of them) from the SAME parameter / optimiser / BatchNorm state, bracketed like bench.py's timed region, MAX over ranks; a form that
  raises on any rank, or whose bounded waits time out, is dropped on every rank.  The state the call found is restored at the end, so
  calibration leaves no trace in the training run.  Within 2 % the one all-reduce wins (the north star's form), then the lower number.
  Returns {"selected": form, "us_per_step": {form: us or None}, "steps": steps}.  SMX_DP_CALIBRATE=0 or SMX_DP_FORM turn it off
  (`selected` is then whatever is in force)."""
  forms = list(forms if forms is not None else dp_forms_available(engine, mode))
  report = {"selected": int(engine.comm_form), "us_per_step": {}, "steps": int(steps)}
  if cp.world <= 1 and not os.environ.get("SMX_FORCE_ALLREDUCE"):
    return report
  if os.environ.get("SMX_DP_CALIBRATE", "1") in ("0", "") or os.environ.get("SMX_DP_FORM", "").strip() or not forms:
    return report
  need = (warmup + steps) * batch
  order = np.ascontiguousarray(np.asarray(order, np.int32)[:need])
  if order.size < need:
    raise ValueError(f"calibrate_forms needs {need} row ids, got {order.size}")
  fake = {}   # test hook: SMX_DP_FAKE_SLOW="2:500,3:80" adds that many us per step to a form's measurement
  for item in filter(None, os.environ.get("SMX_DP_FAKE_SLOW", "").split(",")):
    k, v = item.split(":")
    fake[int(k)] = float(v)
  state = engine.snapshot()
  agreed = lambda ok: cp.max(0.0 if ok else 1.0) == 0.0   # noqa: E731  (True on every rank or on none)
  try:
    for f in forms:
      ok = True
      try:
        engine.comm_set_form(f)
      except Exception:
        ok = False
      if not agreed(ok and engine.comm_form == f):
        report["us_per_step"][f] = None
        continue
      us = None
      try:
        engine.restore(state)
        if warmup:
          engine.train_steps(order[: warmup * batch], warmup, batch)
        engine.stage_steps(order[warmup * batch:], steps, batch)
        engine.synchronize()
        cp.barrier()
        engine.comm_time_allreduce(1)   # device-side line-up of the ranks
        t0 = time.perf_counter()
        engine.train_steps(None, steps, batch)
        engine.synchronize()
        us = 1e6 * (time.perf_counter() - t0) / steps + fake.get(f, 0.0)
        if f == 3 and engine.comm_p2p_error():
          us = None
      except Exception:
        us = None
      worst = cp.max(us if us is not None else float("inf"))
      report["us_per_step"][f] = None if not np.isfinite(worst) else round(worst, 2)
  finally:
    timed = {f: u for f, u in report["us_per_step"].items() if u is not None}
    best = min(timed.values()) if timed else None
    pick = report["selected"] if best is None else (1 if 1 in timed and timed[1] <= 1.02 * best else min(f for f, u in timed.items() if u == best))
    engine.comm_set_form(pick)
    engine.restore(state)
  report["selected"] = int(pick)
  return report
