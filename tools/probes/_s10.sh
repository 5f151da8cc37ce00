set -o pipefail
mkdir -p gpurun_out/s10
for rep in 1 2; do
python tools/probes/probe_parity_time.py cfg3 cfg3d136 cfg2 cfg5 >> gpurun_out/s10/cur.log 2>&1
DYNODE_HIP_LIB=$PWD/tools/probes/_lib_noslp.so python tools/probes/probe_parity_time.py cfg3 cfg3d136 cfg2 cfg5 >> gpurun_out/s10/noslp.log 2>&1
done
python tools/probes/probe_parity_time.py >> gpurun_out/s10/cur_full.log 2>&1
for v in cur noslp cur_full; do echo $v; grep -v amdgpu gpurun_out/s10/$v.log | cut -c1-100; done
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/s10/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s10/pytest.log; tail -8 gpurun_out/s10/pytest.log
