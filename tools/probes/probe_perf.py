"""Perf probe (dev tool): kernel time vs batch size / save density on the current GPU."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic
from dynode_amd.engine import solve_batch

def timeit(wl, ts, B, reps=20, mask=None):
    dev = "cuda"
    m = wl.model
    y0 = torch.as_tensor(wl.y0[:B] if wl.y0.ndim == 2 else wl.y0, dtype=torch.float32, device=dev)
    p = torch.as_tensor(wl.params[:B], dtype=torch.float32, device=dev)
    C = torch.as_tensor(wl.contact, dtype=torch.float32, device=dev)
    tst = torch.as_tensor(ts, dtype=torch.float32, device=dev)
    r = solve_batch(m, y0, p, C, wl.t1, tst, save_mask=mask)
    out = r.ys
    st = (r.status, r.n_accept, r.n_reject)
    for _ in range(3): solve_batch(m, y0, p, C, wl.t1, tst, out=out, stats_out=st, save_mask=mask)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): solve_batch(m, y0, p, C, wl.t1, tst, out=out, stats_out=st, save_mask=mask)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    att = (r.n_accept + r.n_reject).float()
    return ms, float(att.mean()), float(att.max())

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    big = synthetic.WORKLOADS[which](65536)
    full = big.save_ts
    for B in (2048, 4096, 8192, 16384, 32768, 65536):
        for name, ts in (("daily", full), ("2pts", np.array([0.0, 365.0]))):
            ms, am, ax = timeit(big, ts, B)
            gb = B * len(ts) * big.model.state_dim * 4 / ms / 1e6
            print(f"{which} B={B:6d} save={name:5s} ms={ms:8.4f} traj/s={B/ms*1e3:12.0f} out GB/s={gb:8.1f} attempts mean={am:.1f} max={ax:.0f}", flush=True)
