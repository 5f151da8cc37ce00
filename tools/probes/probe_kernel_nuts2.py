"""dev probe: where does the per-chain tail of leapfrog counts come from (cfg4, KernelNUTS)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynode_amd.infer.nuts import KernelNUTS
from dynode_amd.infer.inference import Potential, init_to_median
from examples import sir_infer_parameters as ex
dev = torch.device("cuda")
data = ex.synthetic_incidence(100)
pot = Potential(ex.model, dict(config=ex.get_config(), tf=100, obs_data=data), 0, dev)
z0 = pot.initial(1024, init_to_median, 0)
torch.set_printoptions(precision=4, linewidth=200)
for nw, ns in ((1000, 1000), (1000, 1)):
    s = KernelNUTS(pot.potential_and_grad, max_tree_depth=10, seed=1)
    res = s.run(z0, nw, ns); torch.cuda.synchronize()
    steps = res.num_steps.double().mean(1)
    print("warmup", nw, "samples", ns, "iterations", res.potential_evals, "mean total sampling leapfrogs/chain", float(res.num_steps.sum(1).double().mean()),
          "max", int(res.num_steps.sum(1).max()))
    if ns > 1:
        order = torch.argsort(steps)
        zs = res.samples
        pooled = torch.cov(zs.reshape(-1, 2).T)
        print("pooled posterior cov (z space)", pooled.flatten().tolist())
        for name, idx in (("slowest", order[-6:]), ("median", order[510:514]), ("fastest", order[:4])):
            for i in idx.tolist():
                print(name, i, "steps/trans %.2f" % float(steps[i]), "eps %.4f" % float(res.step_size[i]), "acc %.3f" % float(res.accept_prob[i].mean()),
                      "imm", [round(x, 5) for x in res.inverse_mass[i].flatten().tolist()], "chain cov", [round(x, 5) for x in torch.cov(zs[i].T).flatten().tolist()],
                      "div", int(res.diverging[i].sum()))
