set -o pipefail
mkdir -p gpurun_out/s2
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/s2/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s2/pytest.log
tail -25 gpurun_out/s2/pytest.log
for spl in 4 2 1; do DYNODE_HIP_SPL=$spl python tools/probes/probe_parity_time.py cfg5 >> gpurun_out/s2/spl.log 2>&1; done
for spl in 4 2 1; do DYNODE_HIP_SPL=$spl python tools/probes/probe_parity_time.py cfg3d136 >> gpurun_out/s2/spl.log 2>&1; done
for spl in 1 2; do DYNODE_HIP_SPL=$spl python tools/probes/probe_parity_time.py cfg3 >> gpurun_out/s2/spl.log 2>&1; done
grep -v amdgpu.ids gpurun_out/s2/spl.log
