"""dev probe: randomized parity sweep for the SEIP family (float64, HIP vs oracle): shapes from a pool (both lane
mappings), seasonal forcing / seasonal vaccination / introductions on or off, Tsit5 / Dopri5, constant or adaptive
steps, discontinuity points, irregular save grids, sub-save masks.
    python tests/probes/probe_seip_fuzz.py [n_cases] [first_seed]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from dynode_amd import synthetic
from dynode_amd.engine import solve_batch
O = H.O

POOL = [dict(A=1, L=1, K1=1, M1=2, n_knots=0), dict(A=2, L=2, K1=2, M1=2, n_knots=1), dict(A=2, L=2, K1=3, M1=2, n_knots=2),
        dict(A=4, L=3, K1=2, M1=3, n_knots=3), dict(A=3, L=1, K1=2, M1=5, n_knots=1), dict(A=3, L=3, K1=3, M1=3, n_knots=1)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad, worst = 0, 0.0
for seed in range(seed0, seed0 + N):
    rng = np.random.default_rng(seed)
    shape = dict(POOL[rng.integers(len(POOL))])
    shape.update(seasonal=bool(rng.integers(2)), seasonal_vax=bool(rng.integers(2)), intro=bool(rng.integers(2)))
    t1 = float(rng.uniform(10, 250))
    wl = synthetic.seip(B=int(rng.integers(1, 20)), seed=int(rng.integers(1 << 30)), t1=t1, **shape)
    kind = rng.integers(3)
    ts = (np.sort(rng.uniform(0, t1, int(rng.integers(1, 60)))) if kind == 0 else np.linspace(0, t1, int(rng.integers(2, 120)))
          if kind == 1 else np.array([t1]))
    kw = dict(method=str(rng.choice(["tsit5", "dopri5"])))
    if rng.integers(2):
        kw["constant_dt"] = float(rng.choice([0.1, 0.25, 0.5, 1.0]))
    else:
        kw["rtol"], kw["atol"] = float(10 ** rng.uniform(-9, -4)), float(10 ** rng.uniform(-9, -5))
    if rng.integers(3) == 0:
        kw["jump_ts"] = sorted(float(v) for v in rng.uniform(0, t1, int(rng.integers(1, 5))))
    if rng.integers(2):
        mask = rng.integers(0, 2, 4).astype(np.uint8)
        kw["save_mask"] = mask if mask.any() else np.array([0, 0, 0, 1], np.uint8)
    from dynode_amd import engine
    engine.clear_dispatch_hints()
    if shape["K1"] > 1 and (wl.model.n_age <= 4) and rng.integers(2):
        engine.set_dispatch_hints(seip_tier_lanes=1 if rng.integers(2) else -1)    # (tiers dealt over two lanes: on / off)
    try:
        r = solve_batch(wl.model, wl.y0, wl.params, wl.contact, t1, ts, dtype=torch.float64, **kw)
        want, st, na, nr = O.solve(H.omodel(wl.model), wl.y0, wl.params, wl.contact, t1, ts, dtype=np.float64, n_threads=8, **kw)
        got = r.ys.cpu().numpy()
        fin = np.isfinite(want)
        err = float(np.abs(np.where(fin, got - want, 0)).max() / 1000.0) if fin.any() else 0.0
        same = (np.array_equal(np.isfinite(got), fin) and np.array_equal(r.status.cpu().numpy(), st))
        steps = np.abs((r.n_accept + r.n_reject).cpu().numpy() - (na + nr)).max()
    except Exception as e:  # noqa: BLE001
        err, same, steps = repr(e)[:200], False, -1
    bound = 1e-10 if "constant_dt" in kw else 2e-4       # adaptive: the dose cap has kinks (a few solver tolerances)
    if not same or not isinstance(err, float) or err >= bound:
        bad += 1
        print("MISMATCH seed", seed, shape, kw, "tier lanes", os.environ.get("DYNODE_HIP_SEIP_TIER_LANES"), "err", err, "steps diff", steps, flush=True)
    else:
        worst = max(worst, err if "constant_dt" in kw else 0.0)
print(f"{N} cases, {bad} mismatches, worst constant-step error {worst:.2e}")
