set -o pipefail
mkdir -p gpurun_out/s7
for rep in 1 2; do
python tools/probes/probe_parity_time.py cfg3 cfg3d136 cfg2 cfg5 >> gpurun_out/s7/new.log 2>&1
DYNODE_HIP_LIB=$PWD/tools/probes/_lib_prev.so python tools/probes/probe_parity_time.py cfg3 cfg3d136 cfg2 cfg5 >> gpurun_out/s7/prev.log 2>&1
done
echo NEW; grep -v amdgpu gpurun_out/s7/new.log | cut -c1-170; echo PREV; grep -v amdgpu gpurun_out/s7/prev.log | cut -c1-170
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/s7/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s7/pytest.log; tail -15 gpurun_out/s7/pytest.log
