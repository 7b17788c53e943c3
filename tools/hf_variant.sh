#!/bin/bash
# development: build the fused head with extra -D flags into the library, time it at 128 x 20 000, restore nothing (rebuild afterwards)
cd "$(dirname "$0")/.."
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c sisua_amd/csrc/smx_headfused.hip -o /tmp/hf_var.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sisua_amd/libsisua_hip.so $(ls sisua_amd/csrc/*.o | grep -v smx_headfused.o) /tmp/hf_var.o -ldl
  echo "== $flags"
  python3 tools/headfused_try.py --time-only zinb --reps 50
  python3 tools/headfused_try.py --time-only nb --reps 50
done
