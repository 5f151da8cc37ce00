"""dev probe: cfg2/cfg3 at the four (dtype, method) combinations, ms per launch and trajectories/s."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic
from dynode_amd.engine import solve_batch

for which in ("cfg2", "cfg3d136", "cfg3"):
    wl = synthetic.WORKLOADS[which]()
    for dtype in (torch.float32, torch.float64):
        for method in ("tsit5", "dopri5"):
            a = [torch.as_tensor(x, dtype=dtype, device="cuda") for x in (wl.y0, wl.params, wl.contact, wl.save_ts)]
            r = solve_batch(wl.model, a[0], a[1], a[2], wl.t1, a[3], dtype=dtype, method=method)
            out, st = r.ys, (r.status, r.n_accept, r.n_reject)
            run = lambda: solve_batch(wl.model, a[0], a[1], a[2], wl.t1, a[3], dtype=dtype, method=method, out=out, stats_out=st)
            for _ in range(3): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            w = 4 if dtype == torch.float32 else 8
            gbs = wl.bytes_per_trajectory(w) * wl.B / ms / 1e6
            steps = float((r.n_accept + r.n_reject).float().mean())
            print(f"{which} {str(dtype)[6:]:8s} {method:7s} ms={ms:7.3f} traj/s={wl.B / ms * 1e3:12.0f} alg GB/s={gbs:7.0f} frac={gbs / 8000:.3f} attempts/traj={steps:.1f} ok={int(r.status.max()) == 0}", flush=True)
            del out, r
