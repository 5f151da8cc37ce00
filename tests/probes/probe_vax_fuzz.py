"""dev probe: randomized parity sweep for the vaccinated members of the s/e/i/r/c family (float64, HIP vs oracle):
shapes of tests/test_gpu_parity.py:VAX, random dose scales (tiers running empty or not), methods, constant / adaptive
steps, discontinuity points, save grids and masks.
    python tests/probes/probe_vax_fuzz.py [n_cases] [first_seed]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from dynode_amd.engine import solve_batch
from test_gpu_parity import VAX, vax_workload
O = H.O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad, worst = 0, 0.0
for seed in range(seed0, seed0 + N):
    rng = np.random.default_rng(seed)
    ages, m = VAX[rng.integers(len(VAX) - 1)]           # the 32-group shape is float32 only
    t1 = float(rng.uniform(20, 250))
    y0, p, C, _, _, pop = vax_workload(ages, m, int(rng.integers(1, 24)), seed=int(rng.integers(1 << 30)), t1=t1)
    at = m.param_dim - m.n_age * (4 + 2 * m.n_vax_knots)
    p[:, at:].reshape(p.shape[0], m.n_age, -1)[:, :, :2] *= float(rng.choice([0.3, 1.0, 4.0, 10.0]))
    ts = np.sort(rng.uniform(0, t1, int(rng.integers(1, 80)))) if rng.integers(2) else np.linspace(0, t1, int(rng.integers(2, 150)))
    kw = dict(method=str(rng.choice(["tsit5", "dopri5"])))
    if rng.integers(2):
        kw["constant_dt"] = float(rng.choice([0.1, 0.25, 0.5]))
    else:
        kw["rtol"], kw["atol"] = float(10 ** rng.uniform(-9, -4)), float(10 ** rng.uniform(-9, -5))
    if rng.integers(3) == 0:
        kw["jump_ts"] = sorted(float(v) for v in rng.uniform(0, t1, int(rng.integers(1, 5))))
    if rng.integers(2):
        n_comp = len(m.compartment_names)
        mask = rng.integers(0, 2, n_comp).astype(np.uint8)
        kw["save_mask"] = mask if mask.any() else np.eye(n_comp, dtype=np.uint8)[0]
    try:
        r = solve_batch(m, y0, p, C, t1, ts, dtype=torch.float64, **kw)
        want, st, na, nr = O.solve(H.omodel(m), y0, p, C, t1, ts, dtype=np.float64, n_threads=8, **kw)
        got = r.ys.cpu().numpy()
        fin = np.isfinite(want)
        err = float(np.abs(np.where(fin, got - want, 0)).max() / 1000.0) if fin.any() else 0.0
        same = np.array_equal(np.isfinite(got), fin) and np.array_equal(r.status.cpu().numpy(), st)
    except Exception as e:  # noqa: BLE001
        err, same = repr(e)[:200], False
    bound = 1e-10 if "constant_dt" in kw else 2e-4
    if not same or not isinstance(err, float) or err >= bound:
        bad += 1
        print("MISMATCH seed", seed, m, kw, "err", err, flush=True)
    elif "constant_dt" in kw:
        worst = max(worst, err)
print(f"{N} cases, {bad} mismatches, worst constant-step error {worst:.2e}")
