"""dev probe: KernelNUTS vs GraphNUTS on a Gaussian and on cfg4: iterations, step sizes, divergences."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynode_amd.infer.nuts import KernelNUTS, GraphNUTS

dev = torch.device("cuda")
cov = torch.tensor([[4.0, 1.8, 0.0], [1.8, 1.0, 0.0], [0.0, 0.0, 0.25]], dtype=torch.float64, device=dev)
prec = torch.linalg.inv(cov)
def pg(z):
    g = z @ prec
    return 0.5 * (z * g).sum(-1), g
torch.manual_seed(0)
z0 = torch.randn(256, 3, dtype=torch.float64, device=dev)
def report(name, res, dt):
    q = lambda t: [round(float(x), 4) for x in torch.quantile(t.double(), torch.tensor([0., .05, .5, .95, 1.], dtype=torch.float64, device=t.device))]
    print(name, "evals", res.potential_evals, "sec %.2f" % dt, "div", int(res.diverging.sum()), "steps/trans %.2f" % float(res.num_steps.double().mean()),
          "acc %.3f" % float(res.accept_prob.mean()), "eps q", q(res.step_size), "acc/chain q", q(res.accept_prob.mean(1)),
          "steps/chain q", q(res.num_steps.double().mean(1)), flush=True)
for cls in (KernelNUTS, GraphNUTS):
    t = time.time(); res = cls(pg, max_tree_depth=10, seed=1).run(z0, 500, 500); torch.cuda.synchronize()
    report("gauss " + cls.__name__, res, time.time() - t)

from dynode_amd.infer.inference import Potential
from examples import sir_infer_parameters as ex
data = ex.synthetic_incidence(100)
pot = Potential(ex.model, dict(config=ex.get_config(), tf=100, obs_data=data), 0, dev)
from dynode_amd.infer.inference import init_to_median
z0 = pot.initial(128, init_to_median, 0)
for cls in (KernelNUTS, GraphNUTS):
    t = time.time(); res = cls(pot.potential_and_grad, max_tree_depth=10, seed=1).run(z0, 500, 500); torch.cuda.synchronize()
    report("cfg4 " + cls.__name__, res, time.time() - t)
