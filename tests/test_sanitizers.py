"""CPU-side sanitizer builds (SURVEY.md section 5 "Race detection / sanitizers": `-fsanitize=address` host build).
tools/asan_build.sh compiles, with AddressSanitizer + UBSan,
  * the C / OpenMP port of the step (oracle/sisua_step.c, the timed CPU baseline) with a driver that trains every likelihood with and without
    BatchNorm and reads every tensor back (its first run found that ost_destroy freed nothing: fixed), and
  * the HOST side of every translation unit of libsisua_hip.so (`hipcc --cuda-host-only`: no device code) with a driver that walks the entry
    points that need no device: the ABI queries, smx_shuffle_order, every argument / configuration check of smx_model_create up to its first
    device call, the accessors on a null model.
Never run on the GPU box (no GPU sanitizers on this pool): both programs run here, on the CPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None or not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs gcc and hipcc")
def test_host_code_is_clean_under_address_and_ub_sanitizers():
  r = subprocess.run([os.path.join(ROOT, "tools", "asan_build.sh"), "--run"], capture_output=True, text=True, timeout=900, cwd=ROOT)
  out = r.stdout + r.stderr
  assert r.returncode == 0, out[-4000:]
  assert "CSTEP DRIVER OK" in out and "HOST DRIVER OK" in out, out[-4000:]
  assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out and "LeakSanitizer" not in out, out[-4000:]
