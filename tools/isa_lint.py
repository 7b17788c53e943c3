#!/usr/bin/env python3
"""ISA lint of the built gfx950 code objects: two instruction forms that return wrong results on MI355X and that neither the
compiler's hazard recogniser nor wait states in the issuing wave protect against.  Both were found in head_fused_kernel
(sisua_amd/csrc/smx_headfused.hip), reproduced in isolation (tools/dev/pk_opsel_hazard.hip, pk_opsel_forms.hip, store_hazard.hip;
results: profiles/r06_hazards.txt) and are checked here in EVERY kernel of the library, at build time and by tests/test_isa_lint.py.

R1  packed-f32 op_sel.  `v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32` with an `op_sel` that takes the HIGH dword of src1 for the LOW
    result while src0 takes its low dword (`op_sel:[0,1]`, `[0,1,0]`): in lanes 48-63 that operand reads as 0 whenever the SIMD's OTHER
    wave issues a bf16 MFMA (`v_mfma_f32_16x16x32_bf16`, `32x32x16_bf16`) in the same cycles -- the product is 0, the sum / fma is src0.
    Cross-wave: no `s_nop` helps.  Forms whose src0 also takes its high dword (`op_sel:[1,0]`, `[1,1]`, `[1,0,0]`, `[1,1,0]`) and every
    `op_sel_hi` form measured clean; the rule flags every packed-f32 `op_sel` with src0's bit clear and any other bit set.
R2  wide buffer store, register soffset.  `buffer_store_dwordx3/x4 vdata, voff, srsrc, sN offen` followed DIRECTLY by a vector
    instruction that writes one of vdata's registers stores the NEW value in part of the lanes (0.15 % of the dwords on an idle
    chip).  One wait state cures it; with an immediate soffset two are needed and the compiler inserts them, with a register soffset
    it inserts none (LLVM GCNHazardRecognizer::createsVALUHazard exempts that form).
usage: isa_lint.py [object or code-object files...]   (default: sisua_amd/csrc/*.o)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("SMX_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

_PK = re.compile(r"^\s*(v_pk_(?:mul|add|fma)_f32)\b.*\bop_sel:\[([01,]+)\]")
_STORE = re.compile(r"^\s*buffer_store_(?:dwordx[34]|format_xyzw?)\s+v\[(\d+):(\d+)\],\s*[^,]+,\s*s\[\d+:\d+\],\s*(\S+)")
_VDST = re.compile(r"^\s*(v_\w+)\s+(v\[(\d+):(\d+)\]|v(\d+)\b)")


def disassemble(path):
  """device ISA of a host object with a .hip_fatbin section, or of a bare code object; one instruction (or label) per entry"""
  with tempfile.TemporaryDirectory() as td:
    co = path
    with open(path, "rb") as f:
      is_host_object = b".hip_fatbin" in f.read()
    if is_host_object:
      fat = os.path.join(td, "fat.bin")
      subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, path], check=True)
      co = os.path.join(td, "dev.co")
      subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=" + TARGET, "--output=" + co], check=True,
                     stderr=subprocess.DEVNULL)
    out = subprocess.run([LLVM + "/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
  return out.split("\n")


def lint_lines(lines, where=""):
  """-> list of (rule, kernel, text).  `lines`: llvm-objdump -d output or `hipcc -S` text"""
  found, kernel, pending = [], "?", None
  for raw in lines:
    line = raw.split("//")[0].split(";")[0].rstrip()
    m = re.match(r"^[0-9a-f]* ?<([^>]+)>:\s*$", line) or re.match(r"^(_Z\w+):\s*$", line)
    if m:
      if not m.group(1).startswith((".L", "L")):
        kernel = m.group(1)
      continue
    if not line.strip() or line.strip().startswith(".") or line.strip().endswith(":"):
      continue
    if pending is not None:   # the instruction right behind a wide store with a register soffset
      lo, hi, text = pending
      pending = None
      d = _VDST.match(line)
      if d and not d.group(1).startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
        a, b = (int(d.group(3)), int(d.group(4))) if d.group(3) else (int(d.group(5)), int(d.group(5)))
        if a <= hi and b >= lo:
          found.append(("R2", kernel, text.strip() + "  ->  " + line.strip()))
    m = _PK.match(line)
    if m and m.group(2).startswith("0") and "1" in m.group(2):
      found.append(("R1", kernel, line.strip()))
    m = _STORE.match(line)
    if m and re.match(r"^(s\d+|m0|vcc_lo|vcc_hi|ttmp\d+)$", m.group(3).rstrip(",")):
      pending = (int(m.group(1)), int(m.group(2)), line)
  return [(r, k, (where + ": " if where else "") + t) for r, k, t in found]


def lint_files(paths):
  found = []
  for p in paths:
    found += lint_lines(disassemble(p), os.path.basename(p))
  return found


def default_objects():
  return sorted(glob.glob(os.path.join(ROOT, "sisua_amd", "csrc", "*.o")))


def main():
  paths = sys.argv[1:] or default_objects()
  if not paths:
    print("isa_lint: no objects (build first)")
    return 1
  found = lint_files(paths)
  for r, k, t in found:
    print("%s  %s\n      %s" % (r, k, t))
  print("isa_lint: %d file(s), %d finding(s)" % (len(paths), len(found)))
  return 2 if found else 0


if __name__ == "__main__":
  sys.exit(main())
