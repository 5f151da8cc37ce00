#!/bin/bash
# one-wave vs two-wave (producer / consumer) kernel under the PMC counters: tools/probes/pmc_pc.sh <workload> <outdir>
W=${1:-cfg5}; OUT=${2:-gpurun_out/pmc_pc}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p "$ROOT/$OUT"; cd /tmp && export TMPDIR=/tmp
for MODE in 0 1; do
  export DYNODE_HIP_PC=$MODE
  i=0
  for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $C --output-format csv -d "$ROOT/$OUT/${W}_pc${MODE}_$i" -- python3 "$ROOT/tools/probes/pmc_run.py" $W 3 > "$ROOT/$OUT/${W}_pc${MODE}_$i.log" 2>&1 || tail -3 "$ROOT/$OUT/${W}_pc${MODE}_$i.log"
  done
  python3 "$ROOT/tools/probes/pmc_mean.py" "$ROOT/$OUT/${W}_pc${MODE}_1" "$ROOT/$OUT/${W}_pc${MODE}_2"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/${W}_pc${MODE}_t" -- python3 "$ROOT/tools/probes/pmc_run.py" $W 20 > /dev/null 2>&1
  find "$ROOT/$OUT/${W}_pc${MODE}_t" -name "*kernel_stats.csv" -exec head -3 {} \;
done
