"""dev probe: divergences of the short test-sized NUTS runs (48 chains x 250 + 250) over sampler seeds, per sampler / adaptation:
how much of a fixed-seed assertion on their COUNT is luck."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynode_amd.infer.inference import MCMCProcess
from examples import sir_infer_parameters as ex
data = ex.synthetic_incidence(100)
for model, name in ((ex.model, "model"), (ex.model_fused, "model_fused")):
    for adaptation in ("pooled", "per_chain"):
        out = []
        for seed in (8675314, 1, 2, 3, 4, 5):
            p = MCMCProcess(numpyro_model=model, num_warmup=250, num_samples=250, num_chains=48, nuts_max_tree_depth=10, progress_bar=False,
                            inference_prngkey=seed, mcmc_kwargs={"sampler": "kernel", "adaptation": adaptation})
            m = p.infer(config=ex.get_config(), tf=100, obs_data=data)
            out.append((int(m.nuts.diverging.sum()), round(float(m.nuts.accept_prob.mean()), 3)))
        print(name, adaptation, out, flush=True)
