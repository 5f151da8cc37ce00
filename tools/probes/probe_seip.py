"""dev probe: the SEIP kernel on the synthetic ensemble (8 ages x 4 histories x 3 tiers x 4 waning states,
2 strains, D = 960): ms per launch, trajectories/s, algorithmic HBM fraction; all compartments and
cumulative infections only."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic
from dynode_amd.engine import solve_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
A = int(sys.argv[3]) if len(sys.argv) > 3 else 8
K1 = int(sys.argv[4]) if len(sys.argv) > 4 else 3
M1 = int(sys.argv[5]) if len(sys.argv) > 5 else 4
wl = synthetic.seip(B, L=L, A=A, K1=K1, M1=M1)
print(f"shape: {A} ages x {1 << L} histories x {K1} tiers x {M1} waning states, {L} strains, D = {wl.model.state_dim}; "
      f"DYNODE_HIP_SEIP_TIER_LANES={os.environ.get('DYNODE_HIP_SEIP_TIER_LANES', 'auto')}", flush=True)
m = wl.model
a = [torch.as_tensor(x, dtype=torch.float32, device="cuda") for x in (wl.y0, wl.params, wl.contact, wl.save_ts)]
for label, mask in (("all compartments", None), ("cumulative infections only", np.array([0, 0, 0, 1], np.uint8))):
    r = solve_batch(m, a[0], a[1], a[2], wl.t1, a[3], save_mask=mask)
    out, st = r.ys, (r.status, r.n_accept, r.n_reject)
    run = lambda: solve_batch(m, a[0], a[1], a[2], wl.t1, a[3], save_mask=mask, out=out, stats_out=st)
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    d_saved = out.shape[-1]
    gbs = wl.bytes_per_trajectory(4, d_saved) * B / ms / 1e6
    att = (r.n_accept + r.n_reject).float()
    print(f"seip B={B} {label:28s} ms={ms:8.3f} traj/s={B / ms * 1e3:10.0f} alg GB/s={gbs:7.0f} frac={gbs / 8000:.3f} "
          f"attempts mean={float(att.mean()):.1f} max={int(att.max())} ok={int(r.status.max()) == 0}", flush=True)
    del out, r
