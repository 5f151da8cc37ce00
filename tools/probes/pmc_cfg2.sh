#!/bin/bash
# cfg 2 at B = 1024 (latency-bound: 128 waves) with and without replicas under the PMC counters: tools/probes/pmc_cfg2.sh <outdir>
OUT=${1:-gpurun_out/pmc_cfg2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p "$ROOT/$OUT"; cd /tmp && export TMPDIR=/tmp
for R in 0 3; do
  export DYNODE_HIP_REPLICAS_LOG2=$R
  i=0
  for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
    i=$((i+1))
    rocprofv3 --pmc $C --output-format csv -d "$ROOT/$OUT/rep${R}_$i" -- python3 "$ROOT/tools/probes/pmc_run.py" cfg2 3 1024 > "$ROOT/$OUT/rep${R}_$i.log" 2>&1 || tail -3 "$ROOT/$OUT/rep${R}_$i.log"
  done
  echo "== rep_log2 = $R"
  python3 "$ROOT/tools/probes/pmc_mean.py" "$ROOT/$OUT/rep${R}_1" "$ROOT/$OUT/rep${R}_2" "$ROOT/$OUT/rep${R}_3"
done
