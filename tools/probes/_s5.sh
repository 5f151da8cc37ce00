set -o pipefail
mkdir -p gpurun_out/s5
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/s5/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s5/pytest.log
tail -40 gpurun_out/s5/pytest.log
for w in seip seip3 seip83 seip84; do timeout -k 10 120 python tools/probes/probe_parity_time.py $w >> gpurun_out/s5/seipw.log 2>&1; done
grep -v amdgpu.ids gpurun_out/s5/seipw.log
