import os, sys, torch
sys.path.insert(0, os.getcwd())
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
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, r.ys
for name, B in (("cfg3d136", 16384), ("cfg5", 8192), ("cfg5", 65536), ("cfg3d136", 65536)):
    wl = synthetic.WORKLOADS[name](B)
    ref = None
    for rnd in range(2):
        for spl, pull in (("4", None), ("2", "0"), ("2", "1"), ("1", "0"), ("1", "1")):
            os.environ["DYNODE_HIP_SPL"] = spl
            if pull is None: os.environ.pop("DYNODE_HIP_PULL", None)
            else: os.environ["DYNODE_HIP_PULL"] = pull
            ms, ys = run(wl)
            if ref is None: ref = ys.clone()
            print(f"{name:9s} B={B:6d} SPL={spl} pull={pull} {ms:8.4f} ms maxdiff={float((ys-ref).abs().max()):.2e} {_abi.lib().dyn_last_kernel_name().decode()[-26:]}", flush=True)
