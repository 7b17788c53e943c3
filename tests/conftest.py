import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(autouse=True)
def _developer_knobs_do_not_leak():
  """A test may set a developer knob of the library (sisua_amd._hip.set_tuning); every test starts from the defaults."""
  yield
  from sisua_amd import _hip
  if _hip._lib is not None:
    _hip.clear_tuning("")
