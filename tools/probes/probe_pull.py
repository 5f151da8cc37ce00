"""dev probe: work pulling on / off (and forced grid sizes) on the BASELINE shapes, one box, interleaved.
Usage: python tools/probes/probe_pull.py [workload ...]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic
from dynode_amd.engine import solve_batch


def run(wl, reps=20, order=None):
    dev = "cuda"
    m = wl.model
    y0 = torch.as_tensor(wl.y0, dtype=torch.float32, device=dev)
    p = torch.as_tensor(wl.params, dtype=torch.float32, device=dev)
    C = torch.as_tensor(wl.contact, dtype=torch.float32, device=dev)
    ts = torch.as_tensor(wl.save_ts, dtype=torch.float32, device=dev)
    r = solve_batch(m, y0, p, C, wl.t1, ts)
    out, st = r.ys, (r.status, r.n_accept, r.n_reject)
    if order == "sorted":
        order = torch.argsort(r.n_accept + r.n_reject, descending=True, stable=True).to(torch.int32)
    for _ in range(3):
        solve_batch(m, y0, p, C, wl.t1, ts, out=out, stats_out=st, order=order)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        solve_batch(m, y0, p, C, wl.t1, ts, out=out, stats_out=st, order=order)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, float((r.n_accept + r.n_reject).float().mean()), out


if __name__ == "__main__":
    names = sys.argv[1:] or ["cfg3", "cfg3d136", "cfg2", "cfg5"]
    for name in names:
        for B in ((None, 65536) if name in ("cfg3d136", "cfg5", "cfg2") else (None,)):
            wl = synthetic.WORKLOADS[name]() if B is None else synthetic.WORKLOADS[name](B)
            bytes_ = wl.bytes_per_trajectory(4) * wl.B
            ref = None
            for rnd in range(2):
                for tag, env in (("static", {"DYNODE_HIP_PULL": "0"}), ("pull", {"DYNODE_HIP_PULL": "1"}),
                                 ("pull-2048", {"DYNODE_HIP_PULL": "1", "DYNODE_HIP_PULL_WAVES": "2048"}),
                                 ("pull-2560", {"DYNODE_HIP_PULL": "1", "DYNODE_HIP_PULL_WAVES": "2560"}),
                                 ("static+sorted", {"DYNODE_HIP_PULL": "0"}), ("pull+sorted", {"DYNODE_HIP_PULL": "1"})):
                    for k in ("DYNODE_HIP_PULL", "DYNODE_HIP_PULL_WAVES"):
                        os.environ.pop(k, None)
                    os.environ.update(env)
                    ms, att, out = run(wl, order="sorted" if "sorted" in tag else None)
                    if ref is None:
                        ref = out.clone()
                    same = bool(torch.equal(out, ref))
                    print(f"{name:9s} B={wl.B:6d} {tag:14s} {ms:8.4f} ms  frac={bytes_ / ms / 1e6 / 8000:.4f}  attempts={att:.1f} identical={same}", flush=True)
