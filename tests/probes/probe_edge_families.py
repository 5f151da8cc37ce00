"""dev probe: the special-position sweep of `edge_sweep` (tests/test_gpu_parity.py) on the other two kernel families -- a SEIP
model and a vaccinated s/e/i/r/c model -- with constant steps and tight tolerances (their right-hand sides have kinks: at loose
tolerances float64 step decisions may legitimately differ).    python tests/probes/probe_edge_families.py"""
import itertools, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from dynode_amd import synthetic
from dynode_amd.engine import solve_batch
from test_gpu_parity import VAX, vax_workload
O = H.O
cases = []
for t1 in (20.0, 60.0):
    wl = synthetic.seip(B=5, seed=3, A=3, L=2, K1=2, M1=3, n_knots=1, t1=t1)
    cases.append(("seip", wl.model, wl.y0, wl.params, wl.contact, t1))
    ages, m = VAX[1]
    y0, p, C, _, _, _ = vax_workload(ages, m, 5, seed=8, t1=t1)
    cases.append(("vax", m, y0, p, C, t1))
ran = bad = 0
for name, m, y0, p, C, t1 in cases:
    grids = {"end": np.array([t1]), "from_t0": np.linspace(0.0, t1, 11), "quarter": np.arange(0.0, t1 + 1e-9, 0.25)}
    jumpsets = {"none": [], "on_save": [t1 / 2], "t0": [0.0], "t1": [t1], "before_t1": [float(np.nextafter(t1, 0.0))],
                "ulp_pair": [5.0, float(np.nextafter(5.0, 9.0))], "grid": [2.5, 5.0, 7.25], "many": list(np.linspace(0.1, t1 - 0.1, 64)),
                "first_ulp": [float(np.nextafter(0.0, 1.0)), 3.0]}
    modes = {"tight": dict(rtol=1e-10, atol=1e-12), "const_.25": dict(constant_dt=0.25), "const_.7": dict(constant_dt=0.7)}
    for (gn, ts), (jn, js), (mn, kw), method in itertools.product(grids.items(), jumpsets.items(), modes.items(), ("tsit5", "dopri5")):
        kk = dict(kw, method=method, **({"jump_ts": js} if js else {}))
        try:
            r = solve_batch(m, y0, p, C, t1, ts, dtype=torch.float64, **kk)
            want, st, na, nr = O.solve(H.omodel(m), y0, p, C, t1, ts, dtype=np.float64, **kk)
        except Exception as e:  # noqa: BLE001
            print("ERROR", name, t1, gn, jn, mn, method, repr(e)[:200], flush=True)
            bad += 1
            continue
        ran += 1
        got, fin = r.ys.cpu().numpy(), np.isfinite(want)
        err = np.abs(got[fin] - want[fin]).max() / max(np.abs(want[fin]).max(), 1.0) if fin.any() else 0.0
        same = (np.array_equal(np.isfinite(got), fin) and np.array_equal(r.status.cpu().numpy(), st)
                and np.array_equal(r.n_accept.cpu().numpy(), na) and np.array_equal(r.n_reject.cpu().numpy(), nr))
        if not same or not err <= 1e-9:
            bad += 1
            print("MISMATCH", name, t1, gn, jn, mn, method, "err", err, "d_acc", int(np.abs(r.n_accept.cpu().numpy() - na).max()), flush=True)
print(f"{ran} cases run, {bad} mismatches / errors")
