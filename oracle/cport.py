"""ctypes wrapper of the C / OpenMP fp32 port of the step (oracle/sisua_step.c): the timed CPU baseline.
TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from oracle import build_c
from oracle import sisua_oracle as so

_LIK = {"nb": 0, "zinb": 1, "nbd": 2, "zinbd": 3}


class _Cfg(C.Structure):
  _fields_ = [("G", C.c_int32), ("D", C.c_int32), ("n_enc", C.c_int32), ("enc", C.c_int32 * 8), ("n_dec", C.c_int32),
              ("dec", C.c_int32 * 8), ("likelihood", C.c_int32), ("batchnorm", C.c_int32), ("log_norm", C.c_int32),
              ("dropout_enc", C.c_float), ("dropout_dec", C.c_float), ("input_dropout", C.c_float), ("beta", C.c_float),
              ("bn_momentum", C.c_float), ("bn_eps", C.c_float), ("lr", C.c_float), ("b1", C.c_float), ("b2", C.c_float),
              ("adam_eps", C.c_float), ("clipnorm", C.c_float), ("seed", C.c_uint64)]


class CStep:
  """VAE-family training step on the CPU (fp32, OpenMP)."""

  def __init__(self, spec: so.Spec, params, use_blas: bool = True):
    assert spec.model == "vae", "the C port covers the benchmark's VAE family"
    self.lib = C.CDLL(build_c.build(verbose=False))
    self.blas = None
    if use_blas:   # the dense products through the OpenBLAS NumPy links (ILP64 cblas_sgemm in numpy.libs)
      import glob, os
      import numpy
      cands = glob.glob(os.path.join(os.path.dirname(numpy.__file__), "..", "numpy.libs", "libscipy_openblas64_*.so")) + \
          glob.glob(os.path.join(os.path.dirname(numpy.__file__), "..", "numpy.libs", "libopenblas64_*.so"))
      self.lib.ost_use_blas.argtypes = [C.c_char_p]
      for c in cands:
        if self.lib.ost_use_blas(os.path.realpath(c).encode()):
          self.blas = os.path.basename(c)
          break
    else:
      self.lib.ost_use_blas.argtypes = [C.c_char_p]
      self.lib.ost_use_blas(None)
    self.lib.ost_create.restype = C.c_void_p
    self.lib.ost_train_step.restype = C.c_float
    self.lib.ost_tensor_size.restype = C.c_long
    c = _Cfg(G=spec.n_genes, D=spec.latent_dim, n_enc=len(spec.enc_units), n_dec=len(spec.dec_units),
             likelihood=_LIK[spec.likelihood], batchnorm=int(spec.batchnorm), log_norm=int(spec.log_norm),
             dropout_enc=spec.dropout_enc, dropout_dec=spec.dropout_dec, input_dropout=spec.input_dropout, beta=spec.beta,
             bn_momentum=spec.bn_momentum, bn_eps=spec.bn_eps, lr=spec.lr, b1=spec.adam_beta1, b2=spec.adam_beta2,
             adam_eps=spec.adam_eps, clipnorm=spec.clipnorm, seed=spec.seed)
    for i, u in enumerate(spec.enc_units):
      c.enc[i] = u
    for i, u in enumerate(spec.dec_units):
      c.dec[i] = u
    self.names = [n for n, _ in so.manifest(spec)]
    self._keep = [np.ascontiguousarray(params[n], dtype=np.float32) for n in self.names]
    arr = (C.POINTER(C.c_float) * len(self._keep))(*[a.ctypes.data_as(C.POINTER(C.c_float)) for a in self._keep])
    self.h = C.c_void_p(self.lib.ost_create(C.byref(c), arr))
    self.shapes = {n: params[n].shape for n in self.names}

  def close(self):
    if getattr(self, "h", None):
      self.lib.ost_destroy.argtypes = [C.c_void_p]
      self.lib.ost_destroy(self.h)
      self.h = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  @property
  def threads(self):
    return int(self.lib.ost_threads())

  def set_threads(self, n: int):
    self.lib.ost_set_threads(int(n))

  def train_step(self, x, cell_ids, step):
    x = np.ascontiguousarray(x, dtype=np.float32)
    ids = np.ascontiguousarray(cell_ids, dtype=np.int64)
    return float(self.lib.ost_train_step(self.h, x.ctypes.data_as(C.POINTER(C.c_float)), ids.ctypes.data_as(C.POINTER(C.c_int64)),
                                         x.shape[0], int(step)))

  def _get(self, fn):
    out = {}
    for i, n in enumerate(self.names):
      a = np.empty(self.shapes[n], dtype=np.float32)
      fn(self.h, i, a.ctypes.data_as(C.POINTER(C.c_float)))
      out[n] = a
    return out

  def params(self):
    return self._get(self.lib.ost_get_param)

  def grads(self):
    return self._get(self.lib.ost_get_grad)
