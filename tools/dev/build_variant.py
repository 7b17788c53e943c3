#!/usr/bin/env python3
"""Build the library with extra compiler flags into _scratch/ab/lib_<name>.so (objects under _scratch/ab/<name>/), for same-box A/Bs
(tools/dev/lib_ab3.sh).   build_variant.py NAME [extra hipcc flags...]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sisua_amd import build as b

name, extra = sys.argv[1], sys.argv[2:]
odir = os.path.join(ROOT, "_scratch", "ab", name)
os.makedirs(odir, exist_ok=True)
objs = [os.path.join(odir, s.replace(".hip", ".o")) for s in b.SOURCES]


def cc(i):
  cmd = [b._hipcc()] + b.FLAGS + extra + ["-c", os.path.join(b.CSRC, b.SOURCES[i]), "-o", objs[i]]
  r = subprocess.run(cmd, capture_output=True, text=True)
  if r.returncode:
    raise RuntimeError(" ".join(cmd) + "\n" + r.stderr)


with ThreadPoolExecutor(max_workers=7) as ex:
  list(ex.map(cc, range(len(objs))))
lib = os.path.join(ROOT, "_scratch", "ab", "lib_%s.so" % name)
subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"], check=True)
print(lib)
