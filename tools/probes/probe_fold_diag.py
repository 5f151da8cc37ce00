"""dev probe: is cfg 4's potential folded, and do the float32 fused potentials agree with the float64 model log joint?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd.infer.inference import Potential
from dynode_amd.infer import folded
from dynode_amd.simulation import odes
from examples import sir_infer_parameters as ex

data = ex.synthetic_incidence(100)
kw = dict(config=ex.get_config(), tf=100, obs_data=data)
dev = torch.device("cuda")
pot = Potential(ex.model_fused, kw, 0, dev)
f = folded.discover(pot, seed=0, verbose=True)
print("folded:", f is not None)
g = torch.Generator().manual_seed(5)
z = (torch.randn((64, 2), generator=g, dtype=torch.float64) * torch.tensor([1.0, 0.25]) + torch.tensor([0.2, -0.35])).to(dev)
u32, g32 = pot.potential_and_grad(z)
odes.enable_x64(True)
pot64 = Potential(ex.model, kw, 0, dev)
u64, g64 = pot64.potential_and_grad(z)
odes.enable_x64(False)
d = (u32 - u64)
print("u32 - u64: mean %.3e sd %.3e max %.3e ; |u| ~ %.1f" % (float(d.mean()), float(d.std()), float(d.abs().max()), float(u64.abs().mean())))
print("g32 - g64 max rel", float(((g32 - g64).abs() / (1 + g64.abs())).max()))
if f is not None:
    uf, gf = f(z)
    print("folded - u64: mean %.3e sd %.3e max %.3e" % (float((uf - u64).mean()), float((uf - u64).std()), float((uf - u64).abs().max())))
    print("verify:", f.verify(z))
# regression of the difference on u64 (a scale error shows as a slope)
A = torch.stack([torch.ones_like(u64), u64 - u64.mean()], 1)
print("slope of (u32 - u64) against u64:", float(torch.linalg.lstsq(A, d[:, None]).solution[1]))
z2 = folded.held_out_rows(2, torch.Generator().manual_seed(1)).to(dev)
ug, gg = pot.potential_and_grad(z2)
print("held-out general u:", [round(float(v), 1) for v in ug[:12]])
