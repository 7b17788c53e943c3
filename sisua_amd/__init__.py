"""sisua_amd -- MI355X-native (gfx950) implementation of the SISUA VAE training
hot path behind the reference's SingleCellModel.fit/predict/encode surface.

Python host code -> ctypes -> libsisua_hip.so (hand-written HIP kernels).  There
is no CPU fallback: without the built library and a visible MI355X every
compute entry point raises `SmxError`.
"""
from sisua_amd._hip import SmxError  # noqa: F401
from sisua_amd.config import ModelConfig, NetConf, RVmeta  # noqa: F401

__version__ = "0.1.0"
