set -o pipefail
mkdir -p gpurun_out/s20
for rep in 1 2; do
python tools/probes/probe_parity_time.py seip83 seip84 >> gpurun_out/s20/cur.log 2>&1
DYNODE_HIP_LIB=$PWD/tools/probes/_lib_prev.so python tools/probes/probe_parity_time.py seip83 seip84 >> gpurun_out/s20/prev.log 2>&1
done
for v in cur prev; do echo $v; grep -v amdgpu gpurun_out/s20/$v.log | cut -c1-125 | sort; done
timeout -k 10 600 python -m pytest tests/test_seip.py tests/test_gpu_jvp.py -m gpu -q -x > gpurun_out/s20/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s20/pytest.log; tail -5 gpurun_out/s20/pytest.log
