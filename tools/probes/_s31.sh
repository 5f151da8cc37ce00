set -o pipefail
mkdir -p gpurun_out/s31
timeout -k 10 120 python tools/probes/probe_seip_const.py seip83 seip84 2>&1 | grep -v amdgpu | tee -a gpurun_out/s31/const.log
timeout -k 10 300 python tools/probes/probe_parity_time.py seip83 seip84 2>&1 | grep -v amdgpu | cut -c1-200 | tee -a gpurun_out/s31/adaptive.log
timeout -k 10 600 python -m pytest tests/test_seip.py tests/test_gpu_jvp.py -m gpu -q -x > gpurun_out/s31/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s31/pytest.log; tail -3 gpurun_out/s31/pytest.log
