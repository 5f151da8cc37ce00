set -o pipefail
mkdir -p gpurun_out/s9
for rep in 1 2; do
python tools/probes/probe_parity_time.py cfg3 cfg3d136 cfg2 cfg5 >> gpurun_out/s9/cur.log 2>&1
DYNODE_HIP_LIB=$PWD/tools/probes/_lib_noslp.so python tools/probes/probe_parity_time.py cfg3 cfg3d136 cfg2 cfg5 >> gpurun_out/s9/noslp.log 2>&1
DYNODE_HIP_LIB=$PWD/tools/probes/_lib_prev.so python tools/probes/probe_parity_time.py cfg3 cfg3d136 cfg2 cfg5 >> gpurun_out/s9/prev.log 2>&1
done
for v in cur noslp prev; do echo $v; grep -v amdgpu gpurun_out/s9/$v.log | cut -c1-100; done
