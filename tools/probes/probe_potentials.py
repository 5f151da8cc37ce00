"""dev probe: potentials of model vs model_fused vs float64 over a wide grid of unconstrained points."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynode_amd.infer.inference import Potential
from dynode_amd.simulation import odes
from examples import sir_infer_parameters as ex
dev = torch.device("cuda")
data = ex.synthetic_incidence(100)
kw = dict(config=ex.get_config(), tf=100, obs_data=data)
z0 = torch.linspace(-3, 10, 131, dtype=torch.float64); z1 = torch.linspace(-2.0, 1.5, 71, dtype=torch.float64)
Z = torch.stack([m.reshape(-1) for m in torch.meshgrid(z0, z1, indexing="ij")], 1).to(dev)
pa, pb = Potential(ex.model, kw, 0, dev), Potential(ex.model_fused, kw, 0, dev)
ua, ga = pa.potential_and_grad(Z); ub, gb = pb.potential_and_grad(Z)
odes.enable_x64(True)
u64, g64 = Potential(ex.model, kw, 0, dev).potential_and_grad(Z)
odes.enable_x64(False)
rel = u64 - u64.min()
for name, u, g in (("model", ua, ga), ("fused", ub, gb)):
    d = (u - u64)
    near = rel < 30
    print(name, "finite", int(torch.isfinite(u).sum()), "of", u.numel(), "| U - U64 over points within 30 nats of the mode: mean %.2e max|.| %.2e" % (float(d[near].mean()), float(d[near].abs().max())),
          "| grad err max %.2e (rel %.2e)" % (float((g - g64)[near].abs().max()), float(((g - g64)[near].abs() / (g64[near].abs() + 1)).max())))
    worst = torch.argsort(d.abs() * near, descending=True)[:5]
    for w in worst.tolist():
        print("    z", [round(x, 3) for x in Z[w].tolist()], "U64 %.4f  U-U64 %.5f  relU %.2f" % (float(u64[w]), float(d[w]), float(rel[w])))
# along the plateau: z0 from 2 to 10 at the z1 that minimises U64
for zz in (2.0, 4.0, 6.0, 8.0, 10.0):
    m = (Z[:, 0] - zz).abs() < 1e-9
    i = torch.argmin(u64[m]); idx = torch.nonzero(m)[i, 0]
    print("plateau z0=%.0f: z1 %.2f U64 %.4f model-U64 %.5f fused-U64 %.5f" % (zz, float(Z[idx, 1]), float(u64[idx]), float(ua[idx] - u64[idx]), float(ub[idx] - u64[idx])))
