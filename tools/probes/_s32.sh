set -o pipefail
mkdir -p gpurun_out/s32
for rep in 1 2; do
timeout -k 10 120 python tools/probes/probe_seip_const.py seip83 seip84 2>&1 | grep -v amdgpu | tee -a gpurun_out/s32/bar.log
DYNODE_HIP_LIB=$PWD/tools/probes/_lib_nob.so timeout -k 10 120 python tools/probes/probe_seip_const.py seip83 seip84 2>&1 | grep -v amdgpu | tee -a gpurun_out/s32/nob.log
done
