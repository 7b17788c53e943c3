#!/bin/bash
# development: cycle stamps of the phases of the five 8-9 us launches of the C2 step -- both BatchNorm-forward and both
# BatchNorm-backward launches, the head's backward products -- in workgroup ${1:-0}.  Rebuilds smx_kernels / smx_headbwd with -DSMX_STAMPS
# (thread 0 of that workgroup writes clock64() at the seams of the phases) and OVERWRITES sisua_amd/libsisua_hip.so: run it on the GPU
# box's scratch copy (gpurun).  Prints the phases in shader cycles and, scaled by the launch's duration, in microseconds.
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSMX_STAMPS -DSMX_STAMP_WG=${1:-0}"
/opt/rocm/bin/hipcc $F -c sisua_amd/csrc/smx_kernels.hip -o /tmp/st_kernels.o &
/opt/rocm/bin/hipcc $F -c sisua_amd/csrc/smx_headbwd.hip -o /tmp/st_headbwd.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sisua_amd/libsisua_hip.so $(ls sisua_amd/csrc/*.o | grep -v "smx_kernels.o\|smx_headbwd.o") /tmp/st_kernels.o /tmp/st_headbwd.o -ldl
python3 tools/c2_stamps.py ${2:-8kly}
