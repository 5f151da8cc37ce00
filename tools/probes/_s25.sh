set -o pipefail
mkdir -p gpurun_out/s25
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/s25/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s25/pytest.log; tail -5 gpurun_out/s25/pytest.log
