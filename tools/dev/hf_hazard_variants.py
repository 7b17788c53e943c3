#!/usr/bin/env python3
"""ISA-level variants of head_fused_kernel<NBD, u16, 0> for tools/dev/hf_hazard.hip (ROCm 7.2's code generation; the anchors are
checked, so a different compiler fails loudly instead of editing the wrong place).
  hf_hazard_variants.py OUTDIR   -> OUTDIR/{pin,nopin,<variant>...}.hsaco"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "sisua_amd", "csrc", "smx_headfused.hip")
LL = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "--cuda-device-only", "-S"]
KERNEL = "_ZN3smx17head_fused_kernelILi2ELi1ELi0EEEvNS_13HeadFusedArgsE"


def compile_s(out, *defs):
  subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + list(defs) + [SRC, "-o", out], check=True, stderr=subprocess.DEVNULL)


def assemble(s_path, hsaco):
  o = s_path[:-2] + ".o"
  subprocess.run([LL + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s_path, "-o", o], check=True)
  subprocess.run([LL + "/ld.lld", "-shared", o, "-o", hsaco], check=True)


def kernel_span(lines):
  a = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
  b = next(i for i in range(a, len(lines)) if lines[i].strip().startswith("s_endpgm"))
  return a, b


def edit(lines, pattern, fn, count=1):
  """apply fn(line) -> list of lines to the first `count` lines inside the kernel that match `pattern`"""
  a, b = kernel_span(lines)
  out, n = lines[:a], 0
  for l in lines[a:b + 1]:
    if n < count and re.search(pattern, l):
      out += fn(l); n += 1
    else:
      out.append(l)
  assert n == count, (pattern, n)
  return out + lines[b + 1:]


def main():
  outdir = sys.argv[1]
  os.makedirs(outdir, exist_ok=True)
  # pin = the kernel as the library builds it (-fno-slp-vectorize: no packed f32 at all); nopin = with the SLP vectoriser on, which forms
  # the failing v_pk_mul_f32 ... op_sel:[0,1] (rounds 4-5 built that way; round 5 hid it behind an empty-asm "pin", -DSMX_HF_NOPIN then)
  pin_s, nopin_s = os.path.join(outdir, "pin.s"), os.path.join(outdir, "nopin.s")
  compile_s(pin_s, "-fno-slp-vectorize")
  compile_s(nopin_s)
  assemble(pin_s, os.path.join(outdir, "pin.hsaco"))
  assemble(nopin_s, os.path.join(outdir, "nopin.hsaco"))
  base = open(nopin_s).read().split("\n")
  a, b = kernel_span(base)
  body = base[a:b]
  # the unpacking behind the forward product's last MFMA (what -DSMX_HF_NOPIN exposes)
  sdwa = [l for l in body if "v_cvt_f32_u32_sdwa" in l]
  assert len(sdwa) == 2 and all("src0_sel:WORD_1" in l for l in sdwa), sdwa
  NOP = ["\ts_nop 7", "\ts_nop 7"]
  variants = {}
  # wait states right behind the forward product (in front of the first vector instruction that follows its last MFMA)
  variants["nop_after_product"] = edit(base, r"v_mov_b32_e32 v164, v154", lambda l: NOP + [l])
  # the two SDWA conversions as shift + convert
  def nosdwa(l):
    m = re.search(r"v_cvt_f32_u32_sdwa (v\d+), (v\d+) ", l)
    return ["\tv_lshrrev_b32_e32 %s, 16, %s" % (m.group(1), m.group(2)), "\tv_cvt_f32_u32_e32 %s, %s" % (m.group(1), m.group(1))]
  variants["no_sdwa"] = edit(base, r"v_cvt_f32_u32_sdwa", nosdwa, 2)
  # the next unit's counts landed before the unpacked values are written beside their registers
  variants["wait_counts"] = edit(base, r"global_load_dwordx2 v\[164:165\]", lambda l: [l, "\ts_waitcnt vmcnt(0)"])
  # wait states in front of the consumer of x3 / (mu3 + eps)
  variants["nop_before_pkmul"] = edit(base, r"v_pk_mul_f32 v\[154:155\], v\[206:207\], v\[166:167\] op_sel:\[0,1\]", lambda l: NOP + [l])
  variants["scalar_mul"] = edit(base, r"v_pk_mul_f32 v\[154:155\], v\[206:207\], v\[166:167\] op_sel:\[0,1\]",
                                lambda l: ["\tv_mul_f32_e32 v154, v206, v167", "\tv_mul_f32_e32 v155, v207, v167"])
  PK = r"v_pk_mul_f32 v\[154:155\], v\[206:207\], v\[166:167\] op_sel:\[0,1\]"
  # which half of the packed product is wrong: recompute one half behind it
  variants["fix_lo"] = edit(base, PK, lambda l: [l, "\ts_nop 1", "\tv_mul_f32_e32 v154, v206, v167"])
  variants["fix_hi"] = edit(base, PK, lambda l: [l, "\ts_nop 1", "\tv_mul_f32_e32 v155, v207, v167"])
  # the same product without op_sel (x3 copied into both halves of a free register pair: v234 / v235 are beyond the kernel's 234 registers, inside its allocation of 240)
  variants["no_opsel"] = edit(base, PK, lambda l: ["\tv_mov_b32_e32 v234, v167", "\tv_mov_b32_e32 v235, v167", "\tv_pk_mul_f32 v[154:155], v[206:207], v[234:235]"])
  # op_sel kept, the operand pair copied to the free registers first
  variants["opsel_copy"] = edit(base, PK, lambda l: ["\tv_mov_b32_e32 v234, v166", "\tv_mov_b32_e32 v235, v167", "\tv_pk_mul_f32 v[154:155], v[206:207], v[234:235] op_sel:[0,1]"])
  # wait states between the packed product and its consumer
  variants["nop_after_pkmul"] = edit(base, PK, lambda l: [l] + NOP)
  for name, lines in variants.items():
    p = os.path.join(outdir, name + ".s")
    open(p, "w").write("\n".join(lines))
    assemble(p, os.path.join(outdir, name + ".hsaco"))
  print("built:", "pin nopin " + " ".join(variants))


if __name__ == "__main__":
  main()
