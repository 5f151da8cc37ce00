"""dev probe: at which iteration does each chain pass 100/250/450/950/1000/2000 transitions (cfg4, pooled)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynode_amd.infer.nuts import KernelNUTS
from dynode_amd.infer.inference import Potential, init_to_median
from examples import sir_infer_parameters as ex
dev = torch.device("cuda")
data = ex.synthetic_incidence(100)
pot = Potential(ex.model, dict(config=ex.get_config(), tf=100, obs_data=data), 0, dev)
for C in (128, 1024):
    z0 = pot.initial(C, init_to_median, 0)
    s = KernelNUTS(pot.potential_and_grad, max_tree_depth=10, seed=8675314, block=32, adaptation="pooled")
    hist = []
    s.monitor = lambda S: hist.append((S["it"].clone(), S["eps"].clone()))
    res = s.run(z0, 1000, 1000); torch.cuda.synchronize()
    its = torch.stack([h[0] for h in hist]).cpu()          # [blocks, C]
    marks = [75, 100, 150, 250, 450, 950, 1000, 2000]
    print("chains", C, "iterations", res.potential_evals)
    for m in marks:
        first = (its >= m).float().argmax(0) * 32            # iteration at which each chain reached m transitions
        q = torch.quantile(first.float(), torch.tensor([0.0, 0.5, 0.9, 0.99, 1.0]))
        print("  reach %4d transitions at iteration: min %6d median %6d p90 %6d p99 %6d max %6d" % ((m,) + tuple(int(x) for x in q)))
    slow = torch.argsort((its >= 2000).float().argmax(0))[-3:].tolist()
    for c in slow:
        print("  slow chain", c, "eps", float(res.step_size[c]), "imm", res.inverse_mass[c].flatten().tolist(),
              "reach:", [(m, int((its[:, c] >= m).float().argmax()) * 32) for m in marks])
