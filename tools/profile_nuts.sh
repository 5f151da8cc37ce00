#!/bin/bash
# Kernel-trace profile of cfg 4 (NUTS, 128 chains, 300 + 300 transitions) on the GPU box.
# Usage: tools/profile_nuts.sh <tag> [bench_nuts args, e.g. --fused-likelihood]   -> profiles/<tag>_nuts_kernel_stats.csv, profiles/<tag>_nuts_iteration.md
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=/tmp/prof_nuts_$TAG
rm -rf "$OUT"; mkdir -p "$OUT" "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/tools/bench_nuts.py" --chains 128 --warmup 300 --samples 300 "$@" \
  > "$ROOT/gpurun_out/prof_nuts_$TAG.log" 2>&1 || { echo "trace failed"; tail -5 "$ROOT/gpurun_out/prof_nuts_$TAG.log"; exit 1; }
python3 "$ROOT/tools/summarize_nuts_prof.py" "$OUT" "$TAG" "$ROOT/gpurun_out/prof_nuts_$TAG.log" "$@"
mkdir -p "$ROOT/gpurun_out/profiles_out"
cp "$ROOT/profiles/${TAG}_nuts_kernel_stats.csv" "$ROOT/profiles/${TAG}_nuts_iteration.md" "$ROOT/gpurun_out/profiles_out/" 2>/dev/null
