#!/bin/bash
# Profiles bench.py on the GPU box: kernel-trace stats + separate PMC passes (MI355X_MICROARCH.md
# "HBM": FETCH_SIZE and WRITE_SIZE cannot share a pass; counters never combined with sys/hip traces).
# Usage: tools/profile.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/, summary in profiles/
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 50 --warmup 5 --no-cpu-baseline --no-extra $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
PARGS="--steps 4 --warmup 1 --no-cpu-baseline --no-extra $*"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH SQ_VMEM_WR_TA_DATA_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$i" -- python3 "$ROOT/bench.py" $PARGS > "$OUT/pmc_$i.log" 2>&1 || { echo "pmc pass $i ($C) failed"; tail -5 "$OUT/pmc_$i.log"; }
done
python3 "$ROOT/tools/summarize_prof.py" "$OUT" "$TAG" "$@"
# the GPU box's repo copy is scratch: hand the summary, the kernel stats and the traffic table back through gpurun_out/
mkdir -p "$ROOT/gpurun_out/profiles_out"
cp "$ROOT/profiles/${TAG}_rocprof_summary.md" "$ROOT/profiles/traffic.json" "$ROOT/gpurun_out/profiles_out/" 2>/dev/null
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$ROOT/gpurun_out/profiles_out/${TAG}_kernel_stats.csv" \;
