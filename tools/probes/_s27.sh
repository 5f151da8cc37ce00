set -o pipefail
mkdir -p gpurun_out/s27
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dispatch" > gpurun_out/s27/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s27/pytest.log; tail -4 gpurun_out/s27/pytest.log
timeout -k 10 500 python tools/probes/probe_order.py cfg3 cfg3d136 cfg2 > gpurun_out/s27/order.log 2>&1; grep -v amdgpu gpurun_out/s27/order.log | tail -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_order -- python3 $GRAFT_REPO_ROOT/tools/probes/probe_order.py cfg3d136 > $GRAFT_REPO_ROOT/gpurun_out/s27/prof.log 2>&1
find /tmp/prof_order -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/s27/kernel_stats.csv \;
head -8 $GRAFT_REPO_ROOT/gpurun_out/s27/kernel_stats.csv | cut -c1-200
