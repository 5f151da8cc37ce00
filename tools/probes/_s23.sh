set -o pipefail
mkdir -p gpurun_out/s23
timeout -k 10 500 python tools/probes/probe_order.py > gpurun_out/s23/order.log 2>&1; grep -v amdgpu gpurun_out/s23/order.log | tail -20
