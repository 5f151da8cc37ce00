set -o pipefail
mkdir -p gpurun_out/s18
timeout -k 10 900 python -m pytest tests/test_gpu_infer.py -m gpu -q -x -k "folded or structure or fused_likelihood_matches or high_power" > gpurun_out/s18/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s18/pytest.log; tail -15 gpurun_out/s18/pytest.log
for a in per_chain pooled; do
python tools/bench_nuts.py --chains 128 --adaptation $a --fused-likelihood > gpurun_out/s18/fold_$a.log 2>&1; tail -2 gpurun_out/s18/fold_$a.log
python tools/bench_nuts.py --chains 128 --adaptation $a --fused-likelihood --no-fold > gpurun_out/s18/nofold_$a.log 2>&1; tail -2 gpurun_out/s18/nofold_$a.log
done
