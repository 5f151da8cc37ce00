"""dev probe: parity against the fp32 oracle and launch time, per workload.
Usage: python tools/probes/probe_parity_time.py [workload ...]   (DYNODE_HIP_LIB selects a diagnostic library)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic, _abi
from dynode_amd.engine import solve_batch
from oracle import oracle as O

def run(name, B=None, nchk=64, reps=20, method="tsit5"):
    wl = synthetic.WORKLOADS[name]() if B is None else synthetic.WORKLOADS[name](B)
    m = wl.model
    dev = "cuda"
    f32 = torch.float32
    y0 = torch.as_tensor(wl.y0, dtype=f32, device=dev); p = torch.as_tensor(wl.params, dtype=f32, device=dev)
    C = torch.as_tensor(wl.contact, dtype=f32, device=dev); ts = torch.as_tensor(wl.save_ts, dtype=f32, device=dev)
    r = solve_batch(m, y0, p, C, wl.t1, ts, dtype=f32, method=method)
    torch.cuda.synchronize()
    kern = _abi.lib().dyn_last_kernel_name().decode()
    out = r.ys; st = (r.status, r.n_accept, r.n_reject)
    om = O.Model(m.n_age, m.n_strain, m.has_e, m.has_wane, m.has_c, m.n_wane, m.normalize, m.seasonal, m.has_intro,
                 tuple(m.intro_age_mask), m.n_vax_tiers, m.n_vax_knots, m.family, m.seasonal_vax)
    yy = wl.y0[:nchk] if wl.y0.ndim == 2 else wl.y0
    ref, rst, _, _ = O.solve(om, yy, wl.params[:nchk], wl.contact, wl.t1, wl.save_ts, dtype=np.float32, n_threads=8, method=method)
    got = out[:nchk].cpu().numpy()
    err = np.abs(got - ref).max() / wl.population
    mixed = (np.abs(got - ref) / (1e-6 * wl.population + 1e-5 * np.abs(ref))).max()
    for _ in range(3): solve_batch(m, y0, p, C, wl.t1, ts, dtype=f32, out=out, stats_out=st, method=method)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): solve_batch(m, y0, p, C, wl.t1, ts, dtype=f32, out=out, stats_out=st, method=method)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    frac = wl.bytes_per_trajectory(4) * wl.B / (ms * 1e-3) / 8e12
    att = (r.n_accept + r.n_reject).float()
    print(f"{name:9s} {method} B={wl.B:6d} D={m.state_dim:4d} ms={ms:8.4f} traj/s={wl.B/ms*1e3:12.0f} hbm_frac={frac:.4f} attempts mean={float(att.mean()):.1f} max={float(att.max()):.0f} "
          f"status_ok={int(r.status.max())==0} err/scale={err:.3e} mixed(1e-6,1e-5)={mixed:.3f} finite={bool(np.isfinite(got).all())} | {kern}", flush=True)

if __name__ == "__main__":
    names = sys.argv[1:] or ["cfg3", "cfg3d136", "cfg2", "cfg5"]
    for n in names:
        run(n)
    if not sys.argv[1:]:
        run("cfg3d136", method="dopri5")
        run("cfg3", B=65536)
        run("cfg3d136", B=65536)
