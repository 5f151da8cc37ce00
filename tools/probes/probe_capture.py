"""dev probe: why is a model's potential not HIP-graph capturable?  Captures `Potential.potential_and_grad` of the nine-site
multi-strain model and prints the traceback of the first operation the capture refuses."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynode_amd.infer.inference import Potential, init_to_median
from examples import infer_multi_strain as ex_m

sites = int(sys.argv[1]) if len(sys.argv) > 1 else 9
if sites > 9:    # a model without an ODE: `sites` normal means, distributions built from python floats inside the model function
    import numpy as np
    from dynode_amd.infer import distributions as dist, handlers

    y = torch.as_tensor(np.random.default_rng(3).standard_normal((sites, 24)))

    def model(y):
        for i in range(sites):
            loc = handlers.sample(f"loc_{i}", dist.Normal(0.0, 2.0))
            handlers.sample(f"obs_{i}", dist.Normal(loc[..., None], 1.0), obs=y[i])

    pot = Potential(model, dict(y=y), 0, torch.device("cuda"))
else:
    obs = ex_m.synthetic_incidence(120)
    pot = Potential(ex_m.model, dict(config=ex_m.get_config(sites), tf=120, obs_data=obs), 0, torch.device("cuda"))
z = pot.initial(24, init_to_median, 0)
for _ in range(3):
    pot.potential_and_grad(z)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        pot.potential_and_grad(z)
    print("captured fine")
except Exception:
    traceback.print_exc()
