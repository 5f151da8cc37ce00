set -o pipefail
mkdir -p gpurun_out/s19
timeout -k 10 900 python -m pytest tests/test_gpu_infer.py -m gpu -q -x -k "folded or structure" > gpurun_out/s19/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s19/pytest.log; tail -5 gpurun_out/s19/pytest.log
bash tools/profile_nuts.sh r02_folded --fused-likelihood --adaptation per_chain > gpurun_out/s19/prof_nuts.log 2>&1; tail -20 gpurun_out/s19/prof_nuts.log
bash tools/profile.sh r02_seip83 --workload seip83 > gpurun_out/s19/prof_seip83.log 2>&1; tail -30 gpurun_out/s19/prof_seip83.log
