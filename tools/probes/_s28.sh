set -o pipefail
mkdir -p gpurun_out/s28
bash tools/profile.sh r02_cfg3 --workload cfg3 > gpurun_out/s28/prof_cfg3.log 2>&1; tail -3 gpurun_out/s28/prof_cfg3.log
bash tools/profile.sh r02_cfg3d136 --workload cfg3d136 > gpurun_out/s28/prof_cfg3d136.log 2>&1; tail -3 gpurun_out/s28/prof_cfg3d136.log
bash tools/profile.sh r02_seip83 --workload seip83 > gpurun_out/s28/prof_seip83.log 2>&1; tail -3 gpurun_out/s28/prof_seip83.log
cp gpurun_out/profiles_out/traffic.json profiles/traffic.json
timeout -k 10 900 python bench.py > gpurun_out/s28/bench_n1.json 2> gpurun_out/s28/bench_n1.err; echo "bench rc=$?"
DYNODE_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --batch 4096 > gpurun_out/s28/rehearsal_weak.json 2> gpurun_out/s28/rehearsal_weak.err; echo "rehearsal weak rc=$?"; tail -c 400 gpurun_out/s28/rehearsal_weak.json
DYNODE_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 3 --warmup 1 --batch 8192 --scaling strong > gpurun_out/s28/rehearsal_strong.json 2> gpurun_out/s28/rehearsal_strong.err; echo "rehearsal strong rc=$?"; tail -c 400 gpurun_out/s28/rehearsal_strong.json
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/s28/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s28/pytest.log; tail -4 gpurun_out/s28/pytest.log
