"""dev probe: a sub-save (only the cumulative-incidence compartment of the 8 x 4 model, the reference's `sub_save_indices`)
against the full save, ms per launch at B = 16384."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic, _abi
from dynode_amd.engine import solve_batch
wl = synthetic.WORKLOADS["cfg3d136"](16384)
a = [torch.as_tensor(x, dtype=torch.float32, device="cuda") for x in (wl.y0, wl.params, wl.contact, wl.save_ts)]
for name, mask in (("all", None), ("c only", np.array([0, 0, 0, 0, 1], dtype=np.uint8)), ("i + c", np.array([0, 0, 1, 0, 1], dtype=np.uint8))):
    r = solve_batch(wl.model, a[0], a[1], a[2], wl.t1, a[3], save_mask=mask)
    st = (r.status, r.n_accept, r.n_reject)
    for _ in range(12):
        solve_batch(wl.model, a[0], a[1], a[2], wl.t1, a[3], save_mask=mask, out=r.ys, stats_out=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        solve_batch(wl.model, a[0], a[1], a[2], wl.t1, a[3], save_mask=mask, out=r.ys, stats_out=st)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:7s} {e0.elapsed_time(e1) / 40:.4f} ms  checksum {float(r.ys.double().sum()):.6e}  {_abi.lib().dyn_last_kernel_name().decode()[-30:]}", flush=True)
