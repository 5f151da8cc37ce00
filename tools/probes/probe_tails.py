"""dev probe: tail masses of KernelNUTS draws on analytic targets (1024 chains x 1000 draws)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scipy import stats
from dynode_amd.infer.nuts import KernelNUTS, GraphNUTS
dev = torch.device("cuda")
cov = torch.tensor([[4.0, 1.8], [1.8, 1.0]], dtype=torch.float64, device=dev)
prec = torch.linalg.inv(cov)
def gauss(z):
    g = z @ prec
    return 0.5 * (z * g).sum(-1), g
a = torch.tensor([2.0, 0.7], dtype=torch.float64, device=dev)
def loggamma(z):                       # x_i = log Gamma(a_i, 1): U = -a x + e^x
    e = torch.exp(z)
    return (-(a * z) + e).sum(-1), -a + e
def report(name, x, cdf):
    x = x.reshape(-1)
    out = []
    for p in (0.001, 0.01, 0.05, 0.95, 0.99, 0.999):
        q = cdf.ppf(p)
        emp = float((x < q).mean()) if p < 0.5 else float((x > q).mean())
        tgt = p if p < 0.5 else 1 - p
        out.append("%.3g:%.3f" % (p, emp / tgt))
    print("   ", name, "tail mass ratio (draws/exact) at quantiles", " ".join(out), "| mean %.4f sd %.4f (exact %.4f %.4f)" % (x.mean(), x.std(), cdf.mean(), cdf.std()))
for adaptation in ("per_chain", "pooled"):
    for tname, pg, D in (("gauss", gauss, 2), ("loggamma", loggamma, 2)):
        torch.manual_seed(0)
        z0 = torch.randn(1024, D, dtype=torch.float64, device=dev) * 0.1
        res = KernelNUTS(pg, max_tree_depth=10, seed=3, adaptation=adaptation).run(z0, 500, 1000)
        x = res.samples.cpu().numpy()
        print(tname, adaptation, "accept %.3f steps %.2f div %d" % (float(res.accept_prob.mean()), float(res.num_steps.double().mean()), int(res.diverging.sum())))
        if tname == "gauss":
            report("x0", x[..., 0], stats.norm(0, 2.0)); report("x1", x[..., 1], stats.norm(0, 1.0))
        else:
            report("x0 (a=2)", x[..., 0], stats.loggamma(2.0)); report("x1 (a=0.7)", x[..., 1], stats.loggamma(0.7))
