#!/bin/bash
# the fused gradient-solve + sampler launch of cfg 4 under the PMC counters: tools/probes/pmc_nuts.sh <outdir>
OUT=${1:-gpurun_out/pmc_nuts}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p "$ROOT/$OUT"; cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
         "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d "$ROOT/$OUT/p_$i" -- python3 "$ROOT/tools/bench_nuts.py" --chains 128 --warmup 150 --samples 50 --fused-likelihood --adaptation per_chain > "$ROOT/$OUT/p_$i.log" 2>&1 || tail -3 "$ROOT/$OUT/p_$i.log"
done
python3 - "$ROOT/$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/p_*")):
    if not d.endswith(".log"):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "solve_kernel_fused" in r["Kernel_Name"]:
                    a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        print(d.split("/")[-1], {k: round(v[0] / max(v[1], 1), 1) for k, v in acc.items()}, "launches", max((v[1] for v in acc.values()), default=0))
PY
