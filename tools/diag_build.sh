#!/bin/bash
# Builds a diagnostic variant of libdynode_hip.so next to the real one and prints its path:
#   tools/diag_build.sh ROUNDS          n_accept / n_reject return loop iterations / save rounds per wave
#   tools/diag_build.sh NOPRESCALE_ND   tangent kernels keep k = f (no step-scaled rates)
#   tools/diag_build.sh W3              every solve kernel compiled for three waves per SIMD
# Use with DYNODE_HIP_LIB=<path> (python tools/probes/ab_bench.py / ab_other.py take the variant names); docs/perf-log.md has the readings.
set -e
V=${1:?variant}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
mkdir -p "$W/dynode_amd" "$W/include"
cp -r "$ROOT/dynode_amd/csrc" "$W/dynode_amd/" && cp "$ROOT/include/dynode_hip.h" "$W/include/"
rm -rf "$W/dynode_amd/csrc/build"
make -C "$W/dynode_amd/csrc" -j8 -s ${DIAG_TARGET:-all} HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DDYN_DIAG_$V" 2>&1 | grep -v warning || true
cp "$W/dynode_amd/lib/libdynode_hip.so" "$ROOT/tools/probes/_lib_$V.so"
rm -rf "$W"
echo "$ROOT/tools/probes/_lib_$V.so"
