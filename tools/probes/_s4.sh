set -o pipefail
mkdir -p gpurun_out/s4
timeout -k 10 240 python -m pytest tests/test_seip.py -m gpu -q -k "wave_group or A8-L3 or L4" > gpurun_out/s4/wg.log 2>&1; rc=$?; echo "wg rc=$rc" >> gpurun_out/s4/wg.log
tail -30 gpurun_out/s4/wg.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "wave-group tests timed out: stopping"; exit 1; fi
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/s4/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s4/pytest.log
tail -60 gpurun_out/s4/pytest.log
for w in seip83 seip84; do timeout -k 10 120 python tools/probes/probe_parity_time.py $w >> gpurun_out/s4/seipw.log 2>&1; done
grep -v amdgpu.ids gpurun_out/s4/seipw.log
