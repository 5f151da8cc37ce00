#!/bin/bash
# Writes the source revision next to the built library (dynode_amd/lib/BUILD_REV) so that profiles taken on the GPU
# box, which has no .git, can name the code they measured.  Run before a gpurun call that profiles.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT" || exit 1
REV=$(git rev-parse --short=12 HEAD)
if ! git diff --quiet HEAD -- dynode_amd include bench.py; then REV="$REV+dirty"; fi
mkdir -p dynode_amd/lib && echo "$REV" > dynode_amd/lib/BUILD_REV && echo "$REV"
