set -o pipefail
mkdir -p gpurun_out/s6
bash tools/profile.sh r02_cfg3 --workload cfg3 > gpurun_out/s6/prof_cfg3.log 2>&1; tail -3 gpurun_out/s6/prof_cfg3.log
bash tools/profile.sh r02_cfg3d136 --workload cfg3d136 > gpurun_out/s6/prof_d136.log 2>&1; tail -3 gpurun_out/s6/prof_d136.log
bash tools/profile.sh r02_seip83 --workload seip83 > gpurun_out/s6/prof_seip83.log 2>&1; tail -3 gpurun_out/s6/prof_seip83.log
cd $GRAFT_REPO_ROOT && timeout -k 10 500 python bench.py > gpurun_out/s6/bench.json 2> gpurun_out/s6/bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/s6/bench.json; tail -5 gpurun_out/s6/bench.err
