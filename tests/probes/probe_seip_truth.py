"""dev probe: accuracy of the fp32 / fp64 SEIP kernels against SciPy DOP853 (rtol 1e-11) on the synthetic ensemble:
max |error| / population over all compartments and save days."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from dynode_amd import synthetic
from dynode_amd.engine import solve_batch

wl = synthetic.seip(B=6, seed=17, A=8, L=2, K1=3, M1=4, n_knots=2, t1=200.0)
ts = np.arange(0.0, 201.0, 20.0)
truth = np.stack([H.ground_truth_seip(wl.model, wl.y0[b], wl.params[b], wl.contact, 200.0, ts) for b in range(6)])
for dtype in (torch.float32, torch.float64):
    for method in ("tsit5", "dopri5"):
        r = solve_batch(wl.model, wl.y0, wl.params, wl.contact, 200.0, ts, dtype=dtype, method=method)
        err = np.abs(r.ys.cpu().numpy() - truth).max() / 1000.0
        print(f"{str(dtype)[6:]:8s} {method:7s} max |error| / population = {err:.2e}   attempts mean {float((r.n_accept + r.n_reject).float().mean()):.0f}", flush=True)
