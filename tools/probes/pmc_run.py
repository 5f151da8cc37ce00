"""dev probe: N launches of one workload, for rocprofv3 --pmc / --kernel-trace passes (env decides the dispatch mode).
Usage: python3 tools/probes/pmc_run.py <workload> [launches] [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic
from dynode_amd.engine import solve_batch

name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wl = synthetic.WORKLOADS[name](int(sys.argv[3])) if len(sys.argv) > 3 else synthetic.WORKLOADS[name]()
dev = "cuda"
f32 = torch.float32
y0 = torch.as_tensor(wl.y0, dtype=f32, device=dev)
p = torch.as_tensor(wl.params, dtype=f32, device=dev)
C = torch.as_tensor(wl.contact, dtype=f32, device=dev)
ts = torch.as_tensor(wl.save_ts, dtype=f32, device=dev)
r = solve_batch(wl.model, y0, p, C, wl.t1, ts)
for _ in range(n):
    solve_batch(wl.model, y0, p, C, wl.t1, ts, out=r.ys, stats_out=(r.status, r.n_accept, r.n_reject))
torch.cuda.synchronize()
print("ok", name, wl.B, float((r.n_accept + r.n_reject).float().mean()))
