"""dev probe: tangent kernels with discontinuity points / save times in special position: the primal rows must be those of the
plain solve bit for bit, the tangents finite and equal to central differences of plain float64 solves (constant steps: the
discrete map is smooth).    python tests/probes/probe_edge_tangents.py"""
import itertools, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dynode_amd import ModelDesc
from dynode_amd.engine import solve_batch
from test_gpu_parity import random_workload
models = [ModelDesc(n_age=2), ModelDesc(n_age=8), ModelDesc(n_age=1, has_e=True, has_wane=True)]
ran = bad = 0
for m, t1 in itertools.product(models, (10.0, 40.0)):
    y0, p, C, _, _ = random_workload(m, 5, seed=3 + m.n_age, t1=t1)
    P = p.shape[1]
    d = np.zeros((5, 2, P)); d[:, 0, 0] = 1.0; d[:, 1, 1] = 1.0
    grids = {"end": np.array([t1]), "from_t0": np.linspace(0.0, t1, 11), "quarter": np.arange(0.0, t1 + 1e-9, 0.25)}
    jumpsets = {"none": [], "on_save": [t1 / 2], "t0": [0.0], "before_t1": [float(np.nextafter(t1, 0.0))], "ulp_pair": [5.0, float(np.nextafter(5.0, 9.0))],
                "grid": [2.5, 5.0, 7.25], "first_ulp": [float(np.nextafter(0.0, 1.0)), 3.0]}
    modes = {"const_.25": dict(constant_dt=0.25), "const_.7": dict(constant_dt=0.7), "adaptive": dict(rtol=1e-8, atol=1e-10)}
    for (gn, ts), (jn, js), (mn, kw) in itertools.product(grids.items(), jumpsets.items(), modes.items()):
        kk = dict(kw, **({"jump_ts": js} if js else {}))
        r = solve_batch(m, y0, p, C, t1, ts, dtype=torch.float64, dparams=d, **kk)
        r0 = solve_batch(m, y0, p, C, t1, ts, dtype=torch.float64, **kk)
        ran += 1
        ok = torch.equal(r.ys, r0.ys) and torch.equal(r.n_accept, r0.n_accept) and bool(torch.isfinite(r.dys).all())
        err = 0.0
        if ok and mn.startswith("const"):
            eps = 1e-6
            for j in range(2):
                pp, pm = p.copy(), p.copy(); pp[:, j] += eps; pm[:, j] -= eps
                fd = (solve_batch(m, y0, pp, C, t1, ts, dtype=torch.float64, **kk).ys - solve_batch(m, y0, pm, C, t1, ts, dtype=torch.float64, **kk).ys) / (2 * eps)
                err = max(err, float((r.dys[:, :, j] - fd).abs().max() / (1.0 + fd.abs().max())))
        if not ok or err > 1e-5:
            bad += 1
            print("MISMATCH", (m.n_age, m.has_e), t1, gn, jn, mn, "primal equal", torch.equal(r.ys, r0.ys), "finite", bool(torch.isfinite(r.dys).all()), "fd err", err, flush=True)
print(f"{ran} cases run, {bad} mismatches")
