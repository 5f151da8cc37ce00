set -o pipefail
mkdir -p gpurun_out/s22
python tools/probes/probe_sorted.py > gpurun_out/s22/sorted.log 2>&1; grep -v amdgpu gpurun_out/s22/sorted.log
