"""dev probe: one workload under several dispatch hints (engine.dispatch_hints), alternated three times on ONE box: HIP-event ms
per launch behind 40 ms of untimed work.  Usage: python tools/probes/ab_hints.py cfg5 strains_per_lane=4 strains_per_lane=2 ..."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
from dynode_amd import _abi, engine, synthetic
from dynode_amd.engine import solve_batch

name, variants = sys.argv[1], sys.argv[2:] or ["none"]
wl = synthetic.WORKLOADS[name]()
m = wl.model
y0, p, C, ts = (torch.as_tensor(x, dtype=torch.float32, device="cuda") for x in (wl.y0, wl.params, wl.contact, wl.save_ts))
r = solve_batch(m, y0, p, C, wl.t1, ts)
out, st = r.ys, (r.status, r.n_accept, r.n_reject)
res = {v: [] for v in variants}
kern = {}
for rep in range(3):
    for v in variants:
        hint = {} if v == "none" else {k: int(x) for k, x in (kv.split("=") for kv in v.split(","))}
        with engine.dispatch_hints(**hint):
            solve_batch(m, y0, p, C, wl.t1, ts, out=out, stats_out=st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); solve_batch(m, y0, p, C, wl.t1, ts, out=out, stats_out=st); e1.record(); torch.cuda.synchronize()
            n_settle = max(12, int(40.0 / max(e0.elapsed_time(e1), 1e-3)))
            for _ in range(n_settle):
                solve_batch(m, y0, p, C, wl.t1, ts, out=out, stats_out=st)
            e0.record()
            for _ in range(40):
                solve_batch(m, y0, p, C, wl.t1, ts, out=out, stats_out=st)
            e1.record()
            torch.cuda.synchronize()
            res[v].append(round(e0.elapsed_time(e1) / 40, 4))
            kern[v] = _abi.lib().dyn_last_kernel_name().decode()
print(json.dumps({"workload": name, "B": wl.B, "ms": res, "kernel": kern}))
