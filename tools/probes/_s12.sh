set -o pipefail
mkdir -p gpurun_out/s12
timeout -k 10 300 python -m pytest tests/test_gpu_infer.py -m gpu -q -k "high_power or grid_quadrature" -s > gpurun_out/s12/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s12/pytest.log; grep -v "^$" gpurun_out/s12/pytest.log | tail -12
bash tools/profile.sh r02_cfg3 --workload cfg3 > gpurun_out/s12/prof_cfg3.log 2>&1; tail -2 gpurun_out/s12/prof_cfg3.log
bash tools/profile.sh r02_cfg3d136 --workload cfg3d136 > gpurun_out/s12/prof_d136.log 2>&1; tail -2 gpurun_out/s12/prof_d136.log
cd $GRAFT_REPO_ROOT && timeout -k 10 500 python bench.py > gpurun_out/s12/bench.json 2> gpurun_out/s12/bench.err; echo "bench rc=$?"; head -c 1500 gpurun_out/s12/bench.json
