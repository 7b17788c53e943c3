#!/bin/bash
# asan_build.sh -- CPU-side sanitizer builds (SURVEY.md section 5 "Race detection / sanitizers"; never on the GPU box):
#   1. oracle/sisua_step.c (the C / OpenMP CPU baseline) + tools/asan/cstep_driver.c with gcc -fsanitize=address,undefined
#   2. the HOST side of every translation unit of libsisua_hip.so (hipcc --cuda-host-only: no device code) + tools/asan/host_driver.cpp
# Outputs under build/asan/ (git-ignored).  usage: tools/asan_build.sh [--run]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/build/asan"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
mkdir -p "$OUT/obj"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g"
gcc -O1 $SAN -fopenmp -std=gnu99 "$ROOT/oracle/sisua_step.c" "$ROOT/tools/asan/cstep_driver.c" -o "$OUT/cstep_driver" -lm -ldl
ls "$ROOT"/sisua_amd/csrc/*.hip | xargs -P "${ASAN_JOBS:-6}" -I{} sh -c \
  "$HIPCC --cuda-host-only -O1 -std=c++17 -fPIC $SAN -w -c {} -o $OUT/obj/\$(basename {} .hip).o"
# host-only objects still name their translation unit's device blob (__hip_fatbin_<hash>, registered by the module constructor): give
# each an EMPTY stand-in -- no kernel is ever launched from these builds (the driver stops at the first device call on a CPU box)
nm -u "$OUT"/obj/*.o | awk '/__hip_fatbin_/ {print $2}' | sort -u > "$OUT/fatbins.txt"
{ echo '.section .hip_fatbin,"a",@progbits'; while read -r sym; do printf '.globl %s\n.p2align 12\n%s:\n.zero 64\n' "$sym" "$sym"; done < "$OUT/fatbins.txt"; } > "$OUT/fatbins.s"
gcc -c "$OUT/fatbins.s" -o "$OUT/obj/zz_fatbins.o"
$HIPCC -shared -fPIC $SAN -o "$OUT/libsisua_hip_asan.so" "$OUT"/obj/*.o -ldl
$HIPCC --cuda-host-only -x hip -O1 -std=c++17 $SAN -w "$ROOT/tools/asan/host_driver.cpp" -o "$OUT/host_driver" -L"$OUT" -lsisua_hip_asan -Wl,-rpath,"$OUT"
echo "built: $OUT/cstep_driver $OUT/host_driver $OUT/libsisua_hip_asan.so"
if [ "${1:-}" = "--run" ]; then
  export ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1" UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1" LSAN_OPTIONS="suppressions=$ROOT/tools/asan/lsan.supp"
  OMP_NUM_THREADS=3 "$OUT/cstep_driver"
  "$OUT/host_driver"
fi
