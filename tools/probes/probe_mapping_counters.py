"""dev probe: one workload under one lane mapping, for PMC runs --
    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -- python3 tools/probes/probe_mapping_counters.py cfg5 strains_per_lane=2
launches the workload 6 times with the given dispatch hints (none: the library's choice) and prints the instance and HIP-event ms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynode_amd import _abi, engine, synthetic
from dynode_amd.engine import solve_batch

wl = synthetic.WORKLOADS[sys.argv[1]]()
hints = {k: int(v) for k, v in (a.split("=") for a in sys.argv[2:])}
y0, p, C, ts = (torch.as_tensor(x, dtype=torch.float32, device="cuda") for x in (wl.y0, wl.params, wl.contact, wl.save_ts))
with engine.dispatch_hints(**hints):
    r = solve_batch(wl.model, y0, p, C, wl.t1, ts)
    out, st = r.ys, (r.status, r.n_accept, r.n_reject)
    for _ in range(3):
        solve_batch(wl.model, y0, p, C, wl.t1, ts, out=out, stats_out=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        solve_batch(wl.model, y0, p, C, wl.t1, ts, out=out, stats_out=st)
    e1.record()
    torch.cuda.synchronize()
print(sys.argv[1], hints, _abi.lib().dyn_last_kernel_name().decode(), f"{e0.elapsed_time(e1) / 6:.4f} ms")
