#!/usr/bin/env python3
"""Per-kernel instruction statistics of a hipcc --save-temps .s file: vector / MFMA / conversion / transcendental instruction
counts, registers, scratch, occupancy.   usage: python tools/isa_stats.py file.s [name-substring ...]"""
import re
import sys

s = open(sys.argv[1]).read()
pats = sys.argv[2:]
RX = dict(valu=r"\n\s+v_", mfma=r"v_mfma", cvt_bf16=r"v_cvt_pk_bf16", trans=r"v_(?:exp|log|rcp|sqrt|rsq|sin|cos)_f32", ds=r"\n\s+ds_",
          glob=r"\n\s+global_", salu=r"\n\s+s_")
META = dict(vgpr=r"; NumVgprs: (\d+)", agpr=r"; NumAgprs: (\d+)", scratch=r"; ScratchSize: (\d+)", occupancy=r"; Occupancy: (\d+)",
            lds=r"; LDSByteSize: (\d+)")
for part in re.split(r"\n\t\.globl\t", s)[1:]:
  name = part.split("\n", 1)[0].strip()
  if pats and not all(p in name for p in pats):
    continue
  code = part.split("s_endpgm")[0]
  out = [f"{k} {len(re.findall(rx, code))}" for k, rx in RX.items()]
  for k, rx in META.items():
    m = re.search(rx, part)
    out.append(f"{k} {m.group(1) if m else '?'}")
  print(name[:110] + "\n   " + "  ".join(out))
