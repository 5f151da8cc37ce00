"""dev probe: replicas per trajectory (DYNODE_HIP_REPLICAS_LOG2) on the small-state BASELINE shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic, _abi
from dynode_amd.engine import solve_batch


def run(wl, reps=40):
    dev, f32 = "cuda", torch.float32
    a = [torch.as_tensor(x, dtype=f32, device=dev) for x in (wl.y0, wl.params, wl.contact, wl.save_ts)]
    r = solve_batch(wl.model, a[0], a[1], a[2], wl.t1, a[3])
    st = (r.status, r.n_accept, r.n_reject)
    for _ in range(12):
        solve_batch(wl.model, a[0], a[1], a[2], wl.t1, a[3], out=r.ys, stats_out=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        solve_batch(wl.model, a[0], a[1], a[2], wl.t1, a[3], out=r.ys, stats_out=st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, r.ys


import sys as _s
for name, B in ([(a.split(":")[0], int(a.split(":")[1])) for a in _s.argv[1:]] or [("cfg2", 4096), ("cfg2", 16384), ("cfg2", 1024)]):
    wl = synthetic.WORKLOADS[name](B)
    ref = None
    for rnd in range(2):
        for r in ("", "0", "1", "2", "3"):
            if r:
                os.environ["DYNODE_HIP_REPLICAS_LOG2"] = r
            else:
                os.environ.pop("DYNODE_HIP_REPLICAS_LOG2", None)
            ms, ys = run(wl)
            if ref is None:
                ref = ys.clone()
            print(f"{name:6s} B={B:6d} rep_log2={r or 'auto':4s} {ms:8.4f} ms identical={bool(torch.equal(ys, ref))} {_abi.lib().dyn_last_kernel_name().decode()[-40:]}", flush=True)
