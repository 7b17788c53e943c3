"""Build the C port of the step (oracle/sisua_step.c) -> oracle/libsisua_step.so (gcc + OpenMP)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "sisua_step.c")
LIB = os.path.join(HERE, "libsisua_step.so")


def build(force: bool = False, verbose: bool = True) -> str:
  if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
    return LIB
  cmd = ["gcc", "-O3", "-march=x86-64-v3", "-ffp-contract=fast", "-fopenmp", "-shared", "-fPIC", "-std=gnu99", SRC, "-o", LIB, "-lm", "-ldl"]
  if verbose:
    print(" ".join(cmd), flush=True)
  r = subprocess.run(cmd, capture_output=True, text=True)
  if r.returncode != 0:
    raise RuntimeError("C port build failed:\n" + r.stdout + r.stderr)
  return LIB


if __name__ == "__main__":
  print(build(force=True))
