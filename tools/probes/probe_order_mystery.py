"""dev probe: why does a caller-supplied order not help cfg3 inside bench.measure?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic
from dynode_amd.engine import solve_batch
wl = synthetic.WORKLOADS["cfg3"]()
m, f32, dev = wl.model, torch.float32, "cuda"
y0 = torch.as_tensor(wl.y0, dtype=f32, device=dev); p = torch.as_tensor(wl.params, dtype=f32, device=dev)
C = torch.as_tensor(wl.contact, dtype=f32, device=dev); ts = torch.as_tensor(wl.save_ts, dtype=f32, device=dev)
out = torch.empty((wl.B, wl.n_save, m.state_dim), dtype=f32, device=dev); stats = torch.empty((3, wl.B), dtype=torch.int32, device=dev)
def t(order, n=10):
    for _ in range(2): solve_batch(m, y0, p, C, wl.t1, ts, dtype=f32, out=out, stats_out=(stats[0], stats[1], stats[2]), order=order)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): solve_batch(m, y0, p, C, wl.t1, ts, dtype=f32, out=out, stats_out=(stats[0], stats[1], stats[2]), order=order)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("given", t(None))
att = stats[1] + stats[2]
hint = torch.argsort(att, descending=True, stable=True).to(torch.int32)
print("att stats", float(att.float().mean()), int(att.max()), int(att.min()), "hint head", hint[:6].tolist(), "att of hint head", att[hint[:6].long()].tolist())
print("sorted", t(hint))
print("given again", t(None))
for env in ("0", "1"):
    os.environ["DYNODE_HIP_PULL"] = env
    print("PULL", env, "given", t(None), "sorted", t(hint))
