import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch
from dynode_amd.infer.inference import MCMCProcess
from examples import sir_infer_parameters as ex
data = ex.synthetic_incidence(100)
def run(fuse, adaptation, chains=128, n=1000):
    proc = MCMCProcess(numpyro_model=ex.model_fused, num_warmup=n, num_samples=n, num_chains=chains, nuts_max_tree_depth=10,
                       progress_bar=False, mcmc_kwargs={"sampler": "kernel", "adaptation": adaptation, "fuse": fuse})
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = proc.infer(config=ex.get_config(), tf=100, obs_data=data)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    return el, m.nuts.potential_evals, int(m.nuts.diverging.sum())
run(True, "per_chain", 16, 20)
print("unroll", os.environ.get("DYNODE_NUTS_UNROLL"))
for rep in range(2):
    for fuse in (True, False):
        for ad in ("per_chain", "pooled"):
            el, ev, dv = run(fuse, ad)
            print(f"fuse={fuse} {ad:9s} {el:.3f} s  evals={ev}  us/iter={1e6*el/ev:.1f} div={dv}", flush=True)
