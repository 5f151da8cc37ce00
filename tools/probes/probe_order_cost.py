"""dev probe: time of dyn_cost_order alone (forecast + counting sort) against the batch size."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic, schedule
from dynode_amd.engine import solve_batch

wl = synthetic.WORKLOADS["cfg3d136"](32768)
m = wl.model
a = [torch.as_tensor(x, dtype=torch.float32, device="cuda") for x in (wl.y0, wl.params, wl.contact, wl.save_ts[::61])]
schedule.reset()
solve_batch(m, a[0], a[1], a[2], wl.t1, a[3], order="forecast")
(cm,) = schedule._MODELS.values()
assert cm.ready
s = torch.cuda.current_stream()
for B in (16384, 65536, 262144, 1048576, 4194304):
    p = a[1].repeat((B + 32767) // 32768, 1)[:B].contiguous()
    p = p * (1.0 + 0.01 * torch.rand_like(p))
    for _ in range(3):
        o = cm.order(p, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        o = cm.order(p, s)
    e1.record()
    torch.cuda.synchronize()
    ok = bool(torch.equal(torch.sort(o.long()).values, torch.arange(B, device="cuda")))
    print(f"B={B:8d}: dyn_cost_order {e0.elapsed_time(e1) / 10 * 1e3:9.1f} us  permutation={ok}", flush=True)
