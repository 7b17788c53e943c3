"""CPU-side checks of the drop-in boundary: the library builds, loads, exports every
symbol include/sisua_hip.h declares, and the product fails loudly without a GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
  from sisua_amd import build
  build.build(verbose=False)
  from sisua_amd import _hip
  return _hip.load()


def test_header_and_binding_agree(lib):
  from sisua_amd import _hip
  hdr = open(os.path.join(ROOT, "include", "sisua_hip.h")).read()
  hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
  declared = set(re.findall(r"\b(smx_[a-z_0-9]+)\s*\(", hdr))
  assert declared == set(_hip.SIGNATURES), declared ^ set(_hip.SIGNATURES)
  for name in declared:
    assert hasattr(lib, name)
  assert lib.smx_abi_version() == _hip.SMX_ABI_VERSION


def test_config_struct_layout_matches_header():
  import ctypes as C
  from sisua_amd import _hip
  # 5 + 3*(1+8) + (1+4+4+4+1) + 3 ints, 13 floats, 1 int, (pad), 1 u64
  n_int = 5 + 3 * 9 + 14 + 3
  assert C.sizeof(_hip.smx_config) == (n_int + 13 + 1) * 4 + (4 if (n_int + 14) % 2 else 0) + 8
  assert C.sizeof(_hip.smx_metrics) == 32


def test_no_cpu_fallback(lib):
  from sisua_amd import _hip
  from sisua_amd.config import ModelConfig
  from sisua_amd.engine import Engine
  if lib.smx_device_count() > 0:
    pytest.skip("a GPU is visible")
  with pytest.raises(_hip.SmxError):
    Engine(ModelConfig(n_genes=8, enc_units=(4,), dec_units=(4,), latent_dim=2))
  assert lib.smx_init(0) != 0
  assert b"no such HIP device" in lib.smx_last_error()


def test_manifest_and_init_match_oracle():
  from oracle import sisua_oracle as so
  from sisua_amd import config
  from tests.util import make_pair
  for kw in (dict(model="vae", n_genes=50, likelihood="zinb", enc_units=(16,), dec_units=(16, 8), latent_dim=5),
             dict(model="scvi", n_genes=33, likelihood="zinbd", enc_units=(8,), dec_units=(8,), latent_dim=3, encl_units=(4,)),
             dict(model="sisua", n_genes=20, likelihood="nb", enc_units=(8,), dec_units=(8,), latent_dim=3,
                  labels=((5, "nb"), (3, "onehot")), batchnorm=False),
             dict(model="dca", n_genes=20, likelihood="nbd", enc_units=(8,), dec_units=(8,), latent_dim=3),
             dict(model="scale", n_genes=20, likelihood="zinb", enc_units=(8,), dec_units=(8,), latent_dim=3, n_components=5),
             dict(model="sisua", n_genes=20, likelihood="zinb", enc_units=(8,), dec_units=(8,), latent_dim=3, labels=((5, "mixnb3"),))):
    spec, cfg = make_pair(**kw)
    assert config.manifest(cfg) == so.manifest(spec)
    a, b = config.init_params(cfg), so.init_params(spec)
    assert list(a) == list(b)
    for k in a:
      assert np.array_equal(a[k].astype(np.float64), b[k])
