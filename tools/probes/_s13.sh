set -o pipefail
mkdir -p gpurun_out/s13
for rep in 1 2 3; do
python tools/probes/probe_parity_time.py cfg3 cfg3d136 cfg2 cfg5 >> gpurun_out/s13/cur.log 2>&1
DYNODE_HIP_LIB=$PWD/tools/probes/_lib_prev.so python tools/probes/probe_parity_time.py cfg3 cfg3d136 cfg2 cfg5 >> gpurun_out/s13/prev.log 2>&1
done
for v in cur prev; do echo $v; grep -v amdgpu gpurun_out/s13/$v.log | cut -c1-100 | sort; done
