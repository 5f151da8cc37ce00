#!/bin/bash
# Profiles bench.py on the GPU box: kernel-trace stats + separate PMC passes (MI355X_MICROARCH.md
# "HBM": FETCH_SIZE and WRITE_SIZE cannot share a pass).  Usage: tools/profile.sh <tag> [bench args]
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
PARGS="--steps 4 --warmup 1 --no-cpu-baseline $*"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -- python3 "$ROOT/bench.py" $PARGS > "$OUT/pmc_$N.log" 2>&1 || { echo "pmc $C failed"; tail -5 "$OUT/pmc_$N.log"; }
done
find "$OUT" -name "*.csv" | head -50
