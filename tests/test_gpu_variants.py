"""Opt-in code paths kept behind environment flags (measured slower than the default path)
stay correct: the whole-step parity tests are re-run in a subprocess with the flag set.
The library reads these flags once per process, hence the subprocess."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flag", ["SMX_FUSED", "SMX_SMALL_FUSION", "SMX_SIDE_STREAM", "SMX_OUT_FUSED"])
def test_step_parity_under_flag(flag):
  env = dict(os.environ, **{flag: "1"})
  r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_step.py", "-m", "gpu", "-q", "-x", "-k",
                      "one_step or trajectory or injected or golden or committed"],
                     cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
