set -o pipefail
mkdir -p gpurun_out/s26
bash tools/profile.sh r02_cfg3 --workload cfg3 > gpurun_out/s26/prof_cfg3.log 2>&1; tail -4 gpurun_out/s26/prof_cfg3.log
bash tools/profile.sh r02_cfg3d136 --workload cfg3d136 > gpurun_out/s26/prof_cfg3d136.log 2>&1; tail -4 gpurun_out/s26/prof_cfg3d136.log
bash tools/profile.sh r02_seip83 --workload seip83 > gpurun_out/s26/prof_seip83.log 2>&1; tail -4 gpurun_out/s26/prof_seip83.log
cp gpurun_out/profiles_out/traffic.json profiles/traffic.json
timeout -k 10 900 python bench.py > gpurun_out/s26/bench_n1.json 2> gpurun_out/s26/bench_n1.err; echo "bench rc=$?"; tail -c 600 gpurun_out/s26/bench_n1.json
