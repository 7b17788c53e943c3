"""The `.metamodel` sidecar (sisua/models/single_cell_model.py:283-306, models/__init__.py:30-38): the
pickle [class_name, dataset, metadata, init_kwargs] holds plain records only, and a sidecar written by the REFERENCE
(odin RVmeta / NetConf instances inside) is readable without odin.  No GPU needed."""
import pickle
import sys
import types

import numpy as np


def test_sidecar_holds_plain_records_only():
  from sisua_amd import models as M
  from sisua_amd.config import NetConf, RVmeta
  kw = dict(outputs=RVmeta(100, "zinb", True, "transcriptomic"), labels=[RVmeta(5, "nb", True, "proteomic")],
            encoder=NetConf([64, 32], dropout=0.1), decoder=NetConf([32]), beta=2.0, name="x")
  blob = pickle.dumps(["SISUA", "8kly", M._to_plain({"transcriptomic": np.array(["g0", "g1"])}), M._to_plain(kw)])
  refs = []

  class Spy(pickle.Unpickler):   # every class / function the pickle asks for
    def find_class(self, module, name):
      refs.append(module)
      return super().find_class(module, name)

  import io
  Spy(io.BytesIO(blob)).load()
  assert refs and all(r.split(".")[0] == "numpy" for r in refs), refs     # numpy arrays only, no package classes
  back = M._from_plain(pickle.loads(blob)[3])
  assert back["outputs"] == kw["outputs"] and back["labels"] == kw["labels"] and back["encoder"] == kw["encoder"]
  assert back["beta"] == 2.0 and back["name"] == "x"


def test_reference_written_sidecar_is_readable_without_odin(tmp_path):
  """A sidecar as the reference writes it: instances of odin's config classes.  The module is created here only to
  WRITE the file and removed again before reading, as on a machine without odin."""
  mod = types.ModuleType("odin_fake_bay")

  class RVmeta:   # attribute names of odin's record as the reference passes them (train.py:75-89)
    def __init__(self, event_shape, posterior, projection, name, kwargs=None):
      self.event_shape, self.posterior, self.projection, self.name, self.kwargs = event_shape, posterior, projection, name, kwargs or {}

  class NetConf:
    def __init__(self, units, batchnorm=True, dropout=0.0, input_dropout=0.0):
      self.units, self.batchnorm, self.dropout, self.input_dropout = units, batchnorm, dropout, input_dropout
      self.activation, self.flatten_inputs = "relu", True     # attributes this build does not know

  RVmeta.__module__ = NetConf.__module__ = "odin_fake_bay"
  RVmeta.__qualname__, NetConf.__qualname__ = "RVmeta", "NetConf"
  mod.RVmeta, mod.NetConf = RVmeta, NetConf
  sys.modules["odin_fake_bay"] = mod
  try:
    kwargs = dict(outputs=RVmeta((50,), "zinbd", True, "transcriptomic"), latents=RVmeta(12, "diag", True, "latents"),
                  encoder=NetConf([64, 64], True, 0.1, 0.3), decoder=NetConf([64, 64]), log_norm=True, beta=1.0)
    path = tmp_path / "model.metamodel"
    with open(path, "wb") as f:
      pickle.dump(["VAE", "cortex", {"transcriptomic": np.array(["a", "b"])}, kwargs], f)
  finally:
    del sys.modules["odin_fake_bay"]
  from sisua_amd import models as M
  from sisua_amd.config import NetConf as NC, RVmeta as RV
  cls, ds, meta, kw = M.read_metamodel(str(path))
  assert cls == "VAE" and ds == "cortex" and list(meta["transcriptomic"]) == ["a", "b"]
  assert kw["outputs"] == RV(50, "zinbd", True, "transcriptomic") and kw["latents"] == RV(12, "diag", True, "latents")
  assert kw["encoder"] == NC((64, 64), True, 0.1, 0.3) and kw["decoder"] == NC((64, 64))
  assert kw["log_norm"] is True and kw["beta"] == 1.0
