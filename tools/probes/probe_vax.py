"""dev probe: cost of the vaccination lanes -- the example's 3 ages x 3 tiers x 2 strains SEIRS (12 groups)
against a plain 12-age x 2-strain SEIRS of the same state size, 16384 trajectories, 300 days, daily save."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dynode_amd import ModelDesc
from dynode_amd.engine import solve_batch
from dynode_amd.rhs import seirs_multi_strain_ode
from examples import seirs_vaccination as ex

B = 16384
cfg = ex.get_config(); p = ex.get_odeparams(cfg)
pk = seirs_multi_strain_ode.pack(cfg.initializer.get_initial_state(cfg), p)
rng = np.random.default_rng(0)
params = np.repeat(pk.params, B, 0); params[:, :2] *= rng.uniform(0.8, 1.2, (B, 2))          # spread of beta
plain = ModelDesc(n_age=12, n_strain=2, has_e=True, has_wane=True, has_c=True, normalize=False)
cases = {"vaccinated (3 ages x 4 slots)": (pk.model, params), "plain 12 groups": (plain, params[:, :plain.param_dim])}
ts = np.arange(0.0, 301.0)
for name, (m, prm) in cases.items():
    a = [torch.as_tensor(x, dtype=torch.float32, device="cuda") for x in (pk.y0, prm, pk.contact, ts)]
    r = solve_batch(m, a[0], a[1], a[2], 300.0, a[3])
    out, st = r.ys, (r.status, r.n_accept, r.n_reject)
    run = lambda: solve_batch(m, a[0], a[1], a[2], 300.0, a[3], out=out, stats_out=st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    att = (r.n_accept + r.n_reject).float()
    print(f"{name:32s} ms={ms:7.3f} traj/s={B / ms * 1e3:12.0f} attempts mean={float(att.mean()):.1f} max={int(att.max())} "
          f"us/attempt(max)={ms * 1e3 / float(att.max()):.2f} ok={int(r.status.max()) == 0}", flush=True)
