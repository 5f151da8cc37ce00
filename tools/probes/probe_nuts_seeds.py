"""dev probe: cfg 4 (128 chains x 1000 + 1000, per-chain adaptation) over sampler seeds: seconds, gradient evaluations (= the
slowest chain's), divergences.  The wall time of per-chain adaptation is set by ONE chain's luck."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynode_amd.infer.inference import MCMCProcess
from examples import sir_infer_parameters as ex
data = ex.synthetic_incidence(100)
def run(seed, adaptation="per_chain", chains=128, n=1000):
    proc = MCMCProcess(numpyro_model=ex.model_fused, num_warmup=n, num_samples=n, num_chains=chains, nuts_max_tree_depth=10,
                       progress_bar=False, inference_prngkey=seed, mcmc_kwargs={"sampler": "kernel", "adaptation": adaptation})
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = proc.infer(config=ex.get_config(), tf=100, obs_data=data)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    return el, m.nuts.potential_evals, int(m.nuts.diverging.sum()), float(m.nuts.num_steps.double().mean())
run(1, chains=16, n=20)
for ad in ("per_chain", "pooled"):
    for seed in (8675314, 1, 2, 3, 4, 5, 6, 7):
        el, ev, dv, lf = run(seed, ad)
        print(f"{ad:9s} seed={seed:8d} {el:.3f} s evals={ev} us/iter={1e6 * el / ev:.1f} div={dv} leapfrogs/transition={lf:.2f}", flush=True)
