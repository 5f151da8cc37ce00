set -o pipefail
mkdir -p gpurun_out/s24
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dispatch or full_size" > gpurun_out/s24/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s24/pytest.log; tail -6 gpurun_out/s24/pytest.log
timeout -k 10 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/s24/bench_cfg3.json 2> gpurun_out/s24/bench_cfg3.err; tail -c 1800 gpurun_out/s24/bench_cfg3.json; tail -3 gpurun_out/s24/bench_cfg3.err
for w in cfg3d136 seip seip83; do timeout -k 10 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/s24/bench_$w.json 2> gpurun_out/s24/bench_$w.err; python -c "
import json,sys
d=json.loads(open('gpurun_out/s24/bench_$w.json').read().strip().splitlines()[-1])
print('$w', d['ms_per_step'], d['roofline']['frac'], d['roofline']['dispatch_order'])"; done
