"""dev probe: history of slow chains through warm-up (cfg4, KernelNUTS)."""
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
s = KernelNUTS(pot.potential_and_grad, max_tree_depth=10, seed=1, block=16)
hist = []
s.monitor = lambda S: hist.append(tuple(S[k].clone() for k in ("it", "z", "eps", "imm", "wi", "u")))
res = s.run(z0, 1000, 50); torch.cuda.synchronize()
imm00 = res.inverse_mass[:, 0, 0]
bad = torch.argsort(imm00)[-3:].tolist()
good = torch.argsort(imm00)[510:511].tolist()
print("imm00 quantiles", torch.quantile(imm00, torch.tensor([0, .5, .9, .99, 1.0], dtype=torch.float64, device=dev)).tolist())
for c in bad + good:
    print("=== chain", c, "final imm", res.inverse_mass[c].flatten().tolist(), "eps", float(res.step_size[c]))
    last = -1
    for (it, z, eps, imm, wi, u) in hist:
        i = int(it[c])
        if i // 25 != last // 25 or abs(float(z[c, 0])) > 3:
            if i != last:
                print("  it %4d wi %d z (%.3f, %.3f) u %.2f eps %.4f imm00 %.4f imm11 %.4f" % (i, int(wi[c]), float(z[c, 0]), float(z[c, 1]), float(u[c]), float(eps[c]), float(imm[c, 0, 0]), float(imm[c, 1, 1])))
            last = i
