#!/bin/bash
# development: rebuild the fused head with cycle stamps (SMX_HF_STAMPS), run it once at 128 x 20 000 and print the stamps' differences:
# per tile: top -> W image written -> barrier A -> forward -> likelihood -> dP image -> d d -> barrier B -> dW + stores -> (next top)
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSMX_HF_STAMPS -DSMX_HF_STAMP_WAVE=${2:-0} -c sisua_amd/csrc/smx_headfused.hip -o /tmp/hf_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sisua_amd/libsisua_hip.so $(ls sisua_amd/csrc/*.o | grep -v smx_headfused.o) /tmp/hf_stamps.o -ldl
SMX_TUNING="hf_dbg=1" python3 tools/headfused_try.py --time-only ${1:-zinb} --reps 20
