#!/usr/bin/env python3
"""Requests that are waited for right where they are issued: compiles the given .hip sources of sisua_amd/csrc to gfx950 assembly (`hipcc -S`) and lists,
per kernel, the global / buffer loads followed within four instructions by `s_waitcnt vmcnt(0 | 1)` -- with the instruction behind the wait, i.e. the
"first use" that pulled the wait there.  Typical findings (round 5): the sign extension of `rows ? (long)rows[b] : b` inside its branch; `u16 ? (float)h[i] : f[i]`
per element of an unrolled loop; `x + const` of a conditionally loaded register folded into the branch against a constant in the other arm.
usage: isa_earlywait.py smx_headbwd.hip smx_gemm.hip ... [-k substring-of-kernel-name ...]"""
import os, re, subprocess, sys, tempfile

root = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "sisua_amd", "csrc")
args = sys.argv[1:]
pats = [args[i + 1] for i, a in enumerate(args) if a == "-k"]
files = [a for i, a in enumerate(args) if a.endswith(".hip")]
for f in files:
  with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", tmp.name, os.path.join(root, f)],
                   check=True, stderr=subprocess.DEVNULL)
    s = open(tmp.name).read()
  for m in re.finditer(r"^(_ZN3smx\S+):", s, re.M):
    name = m.group(1)
    if pats and not any(p in name for p in pats):
      continue
    body = s[m.start():s.find(".Lfunc_end", m.start())]
    if "s_endpgm" not in body:
      continue
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
    hits = []
    for k, l in enumerate(lines):
      if l.startswith("global_load") or l.startswith("buffer_load"):
        for j in range(k + 1, min(k + 5, len(lines))):
          if lines[j].startswith("global_load") or lines[j].startswith("buffer_load"):
            break
          w = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", lines[j])
          if w and int(w.group(1)) <= 1:
            hits.append((k, l.split()[0], lines[j], lines[j + 1] if j + 1 < len(lines) else ""))
            break
    if hits:
      print(f"== {f} {name}: {len(hits)} of {len(lines)} instructions")
      for h in hits[:8]:
        print("    ", h)
