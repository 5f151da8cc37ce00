#!/bin/bash
# static vs work-pulling dispatch of one workload under the PMC counters: tools/probes/pmc_pull.sh <workload> <outdir>
W=${1:-cfg3}; OUT=${2:-gpurun_out/pmc_pull}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p "$ROOT/$OUT"; cd /tmp && export TMPDIR=/tmp
for MODE in static pull; do
  if [ $MODE = static ]; then export DYNODE_HIP_PULL=0; else unset DYNODE_HIP_PULL; fi
  i=0
  for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $C --output-format csv -d "$ROOT/$OUT/${W}_${MODE}_$i" -- python3 "$ROOT/tools/probes/pmc_run.py" $W 3 > "$ROOT/$OUT/${W}_${MODE}_$i.log" 2>&1 || tail -3 "$ROOT/$OUT/${W}_${MODE}_$i.log"
  done
  python3 "$ROOT/tools/probes/pmc_mean.py" "$ROOT/$OUT/${W}_${MODE}_1" "$ROOT/$OUT/${W}_${MODE}_2"
done
