"""dev probe: SEIP launch time under a constant step (same instruction stream for every build: diagnostic A/B)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic, _abi
from dynode_amd.engine import solve_batch
for name in sys.argv[1:] or ["seip83"]:
    wl = synthetic.WORKLOADS[name]()
    m = wl.model
    f32 = torch.float32
    a = [torch.as_tensor(x, dtype=f32, device="cuda") for x in (wl.y0, wl.params, wl.contact, wl.save_ts)]
    r = solve_batch(m, a[0], a[1], a[2], wl.t1, a[3], constant_dt=1.0, order=None)
    out, st = r.ys, (r.status, r.n_accept, r.n_reject)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            solve_batch(m, a[0], a[1], a[2], wl.t1, a[3], constant_dt=1.0, out=out, stats_out=st, order=None)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    print(f"{name} constant dt=1 (365 steps): {min(ts):.3f} ms  finite={bool(torch.isfinite(out).all())} status_ok={int((r.status == 0).sum())}/{wl.B} steps={float((r.n_accept + r.n_reject).float().mean()):.1f} | {_abi.lib().dyn_last_kernel_name().decode()}", flush=True)
