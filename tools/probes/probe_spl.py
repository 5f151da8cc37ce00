"""dev probe: strains per lane (DYNODE_HIP_SPL) for the 8 x 4 shapes at batch sizes of at most one wave per SIMD."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic, _abi
from dynode_amd.engine import solve_batch


def run(wl, reps=30):
    dev, f32 = "cuda", torch.float32
    a = [torch.as_tensor(x, dtype=f32, device=dev) for x in (wl.y0, wl.params, wl.contact, wl.save_ts)]
    r = solve_batch(wl.model, a[0], a[1], a[2], wl.t1, a[3])
    st = (r.status, r.n_accept, r.n_reject)
    for _ in range(8):
        solve_batch(wl.model, a[0], a[1], a[2], wl.t1, a[3], out=r.ys, stats_out=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        solve_batch(wl.model, a[0], a[1], a[2], wl.t1, a[3], out=r.ys, stats_out=st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, B in (("cfg5", 8192), ("cfg5", 4096), ("cfg3d136", 8192), ("cfg3d136", 16384)):
    wl = synthetic.WORKLOADS[name](B)
    for rnd in range(2):
        for spl in ("4", "2", "1"):
            os.environ["DYNODE_HIP_SPL"] = spl
            ms = run(wl)
            print(f"{name:9s} B={B:6d} SPL={spl} {ms:8.4f} ms {_abi.lib().dyn_last_kernel_name().decode()[-22:]}", flush=True)
