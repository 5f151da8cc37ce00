"""dev probe: producer / consumer two-wave kernel (DYNODE_HIP_PC) against the one-wave kernel, interleaved on one box."""
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
    return e0.elapsed_time(e1) / reps, r.ys


for name, B, extra in (("cfg5", 8192, {}), ("cfg5", 4096, {}), ("cfg3d136", 8192, {}), ("cfg3d136", 2048, {}),
                       ("cfg2", 4096, {"DYNODE_HIP_REPLICAS_LOG2": "0"}), ("cfg2", 8192, {"DYNODE_HIP_REPLICAS_LOG2": "0"})):
    wl = synthetic.WORKLOADS[name](B)
    bytes_ = wl.bytes_per_trajectory(4) * wl.B
    ref = None
    for rnd in range(2):
        for tag, env in (("one-wave", {"DYNODE_HIP_PC": "0"}), ("two-wave", {"DYNODE_HIP_PC": "1"})) + ((("replicas", {"DYNODE_HIP_PC": "0", "DYNODE_HIP_REPLICAS_LOG2": ""}),) if name == "cfg2" else ()):
            os.environ.update(extra)
            os.environ.update(env)
            if env.get("DYNODE_HIP_REPLICAS_LOG2") == "":
                os.environ.pop("DYNODE_HIP_REPLICAS_LOG2")
            ms, out = run(wl)
            ref = out.clone() if ref is None else ref
            print(f"{name:9s} B={B:6d} {tag:9s} {ms:8.4f} ms frac={bytes_ / ms / 1e6 / 8000:.4f} identical={bool(torch.equal(out, ref))} {_abi.lib().dyn_last_kernel_name().decode()[-14:]}", flush=True)
