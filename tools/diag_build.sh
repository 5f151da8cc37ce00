#!/bin/bash
# Builds a diagnostic variant of libdynode_hip.so next to the real one and prints its path:
#   tools/diag_build.sh NOINTERP   save rows hold y + a weight instead of the dense-output polynomial (no interpolation arithmetic)
#   tools/diag_build.sh SAMEROW    every save round overwrites one of two rows (stores stay in L2: no HBM stream)
#   tools/diag_build.sh ROUNDS     n_accept / n_reject return loop iterations / save rounds per wave
# Use with DYNODE_HIP_LIB=<path> python tools/probes/probe_perf.py cfg3   (DESIGN.md section 9 has the readings)
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
