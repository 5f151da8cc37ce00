set -o pipefail
mkdir -p gpurun_out/s29
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dispatch" > gpurun_out/s29/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s29/pytest.log; tail -3 gpurun_out/s29/pytest.log
timeout -k 10 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/s29/bench.json 2> gpurun_out/s29/bench.err; echo "bench rc=$?"; tail -2 gpurun_out/s29/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s29/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('cfg3', r['frac'], r['kernel_ms'], r['dispatch_order'])
r=d['roofline_d136']; print('d136', r['frac'], r['kernel_ms'], r['dispatch_order'])
for k,v in d['other_workloads'].items():
    if 'ms_per_launch' in v: print(k, round(v['ms_per_launch'],4), round(v['hbm_frac'],4), v.get('given_order_ms_per_launch'))
PY
