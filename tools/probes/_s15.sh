set -o pipefail
mkdir -p gpurun_out/s15
for rep in 1 2; do
python tools/probes/probe_parity_time.py seip seip3 seip83 seip84 >> gpurun_out/s15/cur.log 2>&1
DYNODE_HIP_LIB=$PWD/tools/probes/_lib_seipnoslp.so python tools/probes/probe_parity_time.py seip seip3 seip83 seip84 >> gpurun_out/s15/noslp.log 2>&1
DYNODE_HIP_LIB=$PWD/tools/probes/_lib_prev.so python tools/probes/probe_parity_time.py seip seip3 seip83 seip84 >> gpurun_out/s15/prev.log 2>&1
done
for v in cur noslp prev; do echo $v; grep -v amdgpu gpurun_out/s15/$v.log | cut -c1-125 | sort; done
timeout -k 10 600 python -m pytest tests/test_seip.py tests/test_gpu_jvp.py -m gpu -q -x > gpurun_out/s15/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s15/pytest.log; tail -5 gpurun_out/s15/pytest.log
