#!/bin/bash
# development: rebuild the fused head with cycle stamps (SMX_HF_STAMPS), run it once at 128 x 20 000 and print the stamps' differences of
# one wave (argument 2: 0..7) of workgroups 0 and 100.  Half-tile kernel (round 5), per interval:
#   [forward | likelihood | dP image | d d] -> dW + stores of the unit before -> W split of the unit after -> barrier
# Overwrites sisua_amd/libsisua_hip.so: run it on the GPU box's scratch copy (gpurun), rebuild afterwards when run in place.
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSMX_HF_STAMPS -DSMX_HF_STAMP_WAVE=${2:-0} ${HF_DEFS:-} -c sisua_amd/csrc/smx_headfused.hip -o /tmp/hf_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sisua_amd/libsisua_hip.so $(ls sisua_amd/csrc/*.o | grep -v smx_headfused.o) /tmp/hf_stamps.o -ldl
SMX_TUNING="hf_dbg=1${SMX_TUNING:+,$SMX_TUNING}" python3 tools/headfused_try.py --time-only ${1:-zinb} --reps 20
