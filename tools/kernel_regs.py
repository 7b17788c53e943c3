#!/usr/bin/env python3
"""Registers, scratch and code size per kernel of a hipcc --save-temps .s file.   usage: python tools/kernel_regs.py file.s [substring]"""
import re
import sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
  name, body = m.group(1), m.group(2)
  if pat not in name:
    continue
  f = lambda k: (re.search(k + r" (\d+)", body) or [None, "?"])[1]
  m2 = re.search(re.escape(name) + r":.*?; codeLenInByte = (\d+).*?; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; Occupancy: (\d+)", s, re.S)
  print(f"{name[:90]:90s} code {m2.group(1):>6s} vgpr {m2.group(2):>3s} scratch {m2.group(3):>4s} occ {m2.group(4)}" if m2 else name)
