"""ctypes binding of include/sisua_hip.h (the C-ABI of libsisua_hip.so).

This is the thin layer the north star asks for: Python host code -> C-ABI ->
hand-written HIP kernels; no torch / tensorflow on this path.  Importing the
module never needs a GPU; the library itself is loaded lazily and loading
FAILS LOUDLY (no CPU fallback exists in this package).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsisua_hip.so")

SMX_ABI_VERSION = 4
SMX_MAX_LAYERS = 8
SMX_MAX_LABELS = 4

MODEL_KINDS = {"vae": 0, "dca": 1, "scvi": 2, "sisua": 3, "scale": 4, "fvae": 5, "scale_tril": 6, "scale_post": 7}
LIKELIHOODS = {"nb": 0, "zinb": 1, "nbd": 2, "zinbd": 3, "mse": 4}
LABEL_LIKELIHOODS = {"nb": 0, "onehot": 1, "mixnb": 2, "mixgauss": 3, "mixtril": 4, "mixzinb": 5, "nbd": 6, "zinb": 7, "zinbd": 8}
SCVI_PLANE_OPTIONS = {"full": 0, "share": 1, "single": 2}
ACTIVATIONS = {"relu": 0, "linear": 1}


class SmxError(RuntimeError):
  code = 0   # the library's status code (include/sisua_hip.h: SMX_ERR_*)


class smx_config(C.Structure):
  _fields_ = [
      ("abi_version", C.c_int32), ("model", C.c_int32), ("likelihood", C.c_int32), ("n_genes", C.c_int32),
      ("latent_dim", C.c_int32),
      ("n_enc", C.c_int32), ("enc_units", C.c_int32 * SMX_MAX_LAYERS),
      ("n_dec", C.c_int32), ("dec_units", C.c_int32 * SMX_MAX_LAYERS),
      ("n_encl", C.c_int32), ("encl_units", C.c_int32 * SMX_MAX_LAYERS),
      ("n_labels", C.c_int32), ("label_dim", C.c_int32 * SMX_MAX_LABELS), ("label_llk", C.c_int32 * SMX_MAX_LABELS),
      ("label_components", C.c_int32 * SMX_MAX_LABELS), ("label_observed", C.c_int32 * SMX_MAX_LABELS),
      ("scvi_dispersion", C.c_int32), ("scvi_inflation", C.c_int32), ("n_components", C.c_int32),
      ("disc_units", C.c_int32), ("disc_layers", C.c_int32), ("gamma", C.c_float), ("disc_leak", C.c_float),
      ("batchnorm", C.c_int32), ("log_norm", C.c_int32), ("latent_activation", C.c_int32),
      ("dropout_enc", C.c_float), ("dropout_dec", C.c_float), ("input_dropout", C.c_float),
      ("beta", C.c_float), ("alpha", C.c_float), ("clip_library", C.c_float),
      ("bn_momentum", C.c_float), ("bn_eps", C.c_float),
      ("lr", C.c_float), ("adam_beta1", C.c_float), ("adam_beta2", C.c_float), ("adam_eps", C.c_float),
      ("clipnorm", C.c_float),
      ("max_batch", C.c_int32), ("seed", C.c_uint64),
  ]


class smx_metrics(C.Structure):
  _fields_ = [("loss", C.c_float), ("nllk_x", C.c_float), ("nllk_y", C.c_float), ("kl", C.c_float),
              ("kl_l", C.c_float), ("grad_norm_max", C.c_float), ("nan_flag", C.c_int32), ("step", C.c_int32),
              ("tc", C.c_float), ("dtc_loss", C.c_float), ("nllk_o", C.c_float)]

  def as_dict(self):
    return {k: getattr(self, k) for k, _ in self._fields_}


_FP = C.POINTER(C.c_float)
_IP = C.POINTER(C.c_int32)
_VP = C.c_void_p

# name -> (restype, argtypes); every symbol include/sisua_hip.h declares
SIGNATURES = {
    "smx_last_error": (C.c_char_p, []),
    "smx_abi_version": (C.c_int, []),
    "smx_device_count": (C.c_int, []),
    "smx_init": (C.c_int, [C.c_int]),
    "smx_synchronize": (C.c_int, []),
    "smx_model_create": (C.c_int, [C.POINTER(smx_config), C.POINTER(_VP)]),
    "smx_model_destroy": (C.c_int, [_VP]),
    "smx_num_tensors": (C.c_int, [_VP]),
    "smx_tensor_info": (C.c_int, [_VP, C.c_int, C.c_char_p, C.c_int, _IP, _IP]),
    "smx_get_tensor": (C.c_int, [_VP, C.c_int, C.c_int, _FP]),
    "smx_set_tensor": (C.c_int, [_VP, C.c_int, C.c_int, _FP]),
    "smx_num_bn_layers": (C.c_int, [_VP]),
    "smx_get_bn": (C.c_int, [_VP, C.c_int, C.c_int, _FP, _IP]),
    "smx_set_bn": (C.c_int, [_VP, C.c_int, C.c_int, _FP]),
    "smx_get_step": (C.c_int, [_VP, _IP]),
    "smx_set_step": (C.c_int, [_VP, C.c_int32]),
    "smx_dataset_upload": (C.c_int, [_VP, _FP, C.c_int64, C.POINTER(_FP), _FP, C.POINTER(C.c_uint8), C.c_int64]),
    "smx_dataset_upload_u16": (C.c_int, [_VP, C.POINTER(C.c_uint16), C.c_int64, C.POINTER(_FP), _FP, C.POINTER(C.c_uint8), C.c_int64]),
    "smx_dataset_upload_csr": (C.c_int, [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int32), _FP, C.c_int64, C.POINTER(_FP), _FP,
                                         C.POINTER(C.c_uint8), C.c_int64]),
    "smx_dataset_size": (C.c_int64, [_VP]),
    "smx_dataset_generate_lognormal": (C.c_int, [_VP, C.c_uint64, C.c_int32, C.c_int64, C.c_int32, C.c_double]),
    "smx_train_step": (C.c_int, [_VP, _IP, C.c_int32, C.POINTER(smx_metrics)]),
    "smx_train_step_graph": (C.c_int, [_VP, _IP, C.c_int32, C.POINTER(smx_metrics)]),
    "smx_train_steps": (C.c_int, [_VP, _IP, C.c_int32, C.c_int32, C.c_int, C.POINTER(smx_metrics)]),
    "smx_train_stage": (C.c_int, [_VP, _IP, C.c_int32, C.c_int32]),
    "smx_metrics_history": (C.c_int, [_VP, C.c_int32, _FP]),
    "smx_eval_step": (C.c_int, [_VP, _IP, C.c_int32, C.POINTER(smx_metrics)]),
    "smx_forward": (C.c_int, [_VP, _IP, _FP, _FP, C.c_int32, C.c_int32, C.c_int32, _FP, _FP, _FP, _FP, _FP, _FP, _FP,
                              C.POINTER(_FP)]),
    "smx_forward_samples": (C.c_int, [_VP, _IP, _FP, _FP, C.c_int32, C.c_int32, _FP, _FP, _FP, _FP, _FP, _FP, _FP,
                                      C.POINTER(_FP)]),
    "smx_shuffle_order": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "smx_predict": (C.c_int, [_VP, _FP, _FP, C.c_int64, C.c_int32, C.c_int32, _FP, _FP, _FP, _FP, _FP, _FP, _FP,
                              C.POINTER(_FP)]),
    "smx_predict_stat": (C.c_int, [_VP, _FP, _FP, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _FP, _FP]),
    "smx_decode": (C.c_int, [_VP, _FP, _FP, C.c_int32, _FP, C.POINTER(_FP)]),
    "smx_dataset_library": (C.c_int, [_VP, _FP]),
    "smx_dataset_corrupt": (C.c_int, [_VP, C.c_double, C.c_double, C.c_uint64, C.POINTER(C.c_int64)]),
    "smx_dataset_read": (C.c_int, [_VP, C.c_int64, C.c_int64, _FP, _FP, _FP]),
    "smx_marginal_llk": (C.c_int, [_VP, _IP, _FP, _FP, C.c_int32, C.c_int32, _FP, _FP]),
    "smx_score_llk": (C.c_int, [_VP, _IP, _FP, _FP, C.POINTER(_FP), C.c_int32, C.c_int32, C.c_int32, _FP]),
    "smx_set_noise": (C.c_int, [_VP, C.c_int32, _FP, C.c_int32, C.c_int32]),
    "smx_clear_noise": (C.c_int, [_VP]),
    "smx_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "smx_comm_init": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(C.c_uint8)]),
    "smx_comm_p2p_export": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_uint8)]),
    "smx_comm_p2p_init": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(C.c_uint8)]),
    "smx_comm_p2p_error": (C.c_int, [_VP, C.POINTER(C.c_int32)]),
    "smx_comm_time_allreduce": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int64)]),
    "smx_comm_world": (C.c_int, [_VP]),
    "smx_comm_rank": (C.c_int, [_VP]),
    "smx_comm_form": (C.c_int, [_VP]),
    "smx_comm_set_form": (C.c_int, [_VP, C.c_int]),
    "smx_opt_gather": (C.c_int, [_VP]),
    "smx_comm_library": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_int, _IP]),
    "smx_comm_set_sync_bn": (C.c_int, [_VP, C.c_int]),
    "smx_comm_init_local": (C.c_int, [C.POINTER(_VP), C.c_int]),
    "smx_k_adam": (C.c_int, [C.c_int32, _IP, _FP, _FP, _FP, _FP, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                             C.c_float, _FP]),
    "smx_set_flag": (C.c_int, [_VP, C.c_char_p, C.c_int]),
    "smx_set_tuning": (C.c_int, [C.c_char_p, C.c_double]),
    "smx_clear_tuning": (C.c_int, [C.c_char_p]),
    "smx_timing_enable": (C.c_int, [_VP, C.c_char_p]),
    "smx_timing_read": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "smx_loss_bytes_per_cell": (C.c_int64, [_VP]),
    "smx_head_fused_bytes": (C.c_int64, [_VP, C.c_int32]),
    "smx_k_count_llk": (C.c_int, [C.c_int, C.c_int, _FP, _FP, C.c_int32, C.c_int32, _FP, _FP]),
    "smx_k_head_fused": (C.c_int, [C.c_int, C.c_int, _FP, _FP, _FP, _FP, C.c_int32, C.c_int32, C.c_float, C.c_int32, _FP, _FP, _FP, _FP, _FP,
                                   _FP]),
    "smx_k_head_fused_stress": (C.c_int, [C.c_int, C.c_int, _FP, _FP, _FP, _FP, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.POINTER(C.c_int32),
                                          C.POINTER(C.c_int64)]),
    "smx_k_gemm": (C.c_int, [C.c_int, C.c_int, _FP, _FP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _FP]),
    "smx_k_hiprand": (C.c_int, [C.c_uint64, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "smx_k_noise": (C.c_int, [C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.c_int32, C.c_int32,
                              C.c_float, _FP, _FP]),
}

_lib = None


def load():
  """Load libsisua_hip.so and bind every declared symbol.  Raises SmxError when
  the library has not been built (python -m sisua_amd.build) -- there is no
  fallback implementation."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise SmxError(f"{LIB_PATH} is missing: build it with `python -m sisua_amd.build` "
                   "(hipcc --offload-arch=gfx950). sisua_amd has no CPU fallback.")
  try:
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
  except OSError as e:
    raise SmxError(f"cannot load {LIB_PATH}: {e}") from e
  for name, (res, args) in SIGNATURES.items():
    try:
      fn = getattr(lib, name)
    except AttributeError as e:
      raise SmxError(f"{LIB_PATH} does not export {name}") from e
    fn.restype = res
    fn.argtypes = args
  if lib.smx_abi_version() != SMX_ABI_VERSION:
    raise SmxError("libsisua_hip.so ABI version mismatch; rebuild")
  # Data-parallel jobs: bind ROCm's own RCCL now.  The library dlopen()s "librccl.so.1" lazily; if a caller's torch
  # (which bundles its own RCCL) were imported first, that older build would be
  # picked up by soname instead of the one matching /opt/rocm's HIP runtime.
  if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    for cand in ("/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"):
      if os.path.exists(cand):
        try:
          C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
          pass
        break
  _lib = lib
  return lib


def set_tuning(name: str, value: float = 1.0):
  """A developer knob of the library (include/sisua_hip.h: smx_set_tuning; docs/LAB_NOTES.md lists them)."""
  check(load().smx_set_tuning(name.encode(), float(value)))


def clear_tuning(name: str = ""):
  check(load().smx_clear_tuning(name.encode()))


def check(rc: int):
  if rc != 0:
    msg = load().smx_last_error()
    err = SmxError(f"libsisua_hip error {rc}: {msg.decode() if msg else '?'}")
    err.code = int(rc)
    raise err


def require_gpu(device: int = 0):
  lib = load()
  n = lib.smx_device_count()
  if n <= 0:
    raise SmxError("no HIP device visible: sisua_amd runs on MI355X only (no CPU fallback)")
  check(lib.smx_init(device))
  return lib
