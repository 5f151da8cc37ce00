"""dev probe: introduction tangents, one parameter family at a time, against central differences of the oracle."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/probes/x.py -> repo root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from dynode_amd import ModelDesc
from dynode_amd.engine import solve_batch
from test_gpu_jvp import _workload, fd_oracle
m = ModelDesc(n_age=3, n_strain=2, has_e=True, has_wane=True, has_c=True, has_intro=True, intro_age_mask=(0b001, 0b110))
B = 11
y0, p, C, t1, ts = _workload(m, B, seed=9)
S = 2; at = 8
names = ["beta", "gamma", "sigma", "omega", "time", "scale", "pct"]
for k, name in enumerate(names):
    for eps in (1e-4, 1e-6):
        dp = np.zeros((B, 1, m.param_dim)); dp[:, 0, k * S:(k + 1) * S] = 1.0 if name in ("time", "scale") else 0.01
        r = solve_batch(m, y0, p, C, t1, ts, dtype=torch.float64, constant_dt=0.25, dparams=dp)
        want = fd_oracle(m, y0, p, C, t1, ts, dp[:, 0], None, eps=eps, constant_dt=0.25)
        got = r.dys.cpu().numpy()[:, :, 0]
        print(f"{name:6s} eps {eps:g}: rel err {np.abs(got - want).max() / (np.abs(want).max() + 1e-300):.3e}  scale {np.abs(want).max():.3e}")
rng = np.random.default_rng(0)
for label, use_p, use_y in (("dy0 only", False, True), ("params only (test-style)", True, False), ("both", True, True)):
    dp = rng.normal(size=(B, 1, m.param_dim)) * 0.1 * np.abs(p)[:, None, :] * (1.0 if use_p else 0.0)
    dy0 = rng.normal(size=(B, 1, m.state_dim)) * (1.0 if use_y else 0.0)
    r = solve_batch(m, y0, p, C, t1, ts, dtype=torch.float64, constant_dt=0.25, dparams=dp, dy0=dy0)
    got = r.dys.cpu().numpy()[:, :, 0]
    for eps in (1e-4, 1e-5, 1e-6, 1e-7):
        want = fd_oracle(m, y0, p, C, t1, ts, dp[:, 0], dy0[:, 0], eps=eps, constant_dt=0.25)
        err = np.abs(got - want)
        b = np.unravel_index(np.argmax(err), err.shape)
        print(f"{label:26s} eps {eps:g}: rel err {err.max() / np.abs(want).max():.3e} scale {np.abs(want).max():.3e} worst traj {b[0]} day {b[1]} comp {b[2]} | p[traj] intro {p[b[0], 8:]}")
