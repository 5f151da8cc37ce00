set -o pipefail
mkdir -p gpurun_out/s14
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/s14/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s14/pytest.log; tail -6 gpurun_out/s14/pytest.log
python tools/probes/probe_parity_time.py cfg3 cfg3d136 cfg2 cfg5 > gpurun_out/s14/cur.log 2>&1; grep -v amdgpu gpurun_out/s14/cur.log | cut -c1-110
