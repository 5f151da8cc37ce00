"""Dev probe: parity + speed of the strain-split kernel variants (DYNODE_HIP_SPL) on cfg3."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))  # helpers.py
import helpers as H
from dynode_amd import synthetic
from dynode_amd.engine import solve_batch
from probe_perf import timeit
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1
wl = synthetic.seirs_multi_strain(37, seed=2, W=W, t1=120.0)
want, _, na_o, nr_o = H.O.solve(H.omodel(wl.model), wl.y0, wl.params, wl.contact, wl.t1, wl.save_ts, dtype=np.float32, n_threads=8)
r = solve_batch(wl.model, wl.y0, wl.params, wl.contact, wl.t1, wl.save_ts)
torch.cuda.synchronize()
err = np.abs(r.ys.cpu().numpy() - want).max() / 1000
print("SPL env", os.environ.get("DYNODE_HIP_SPL"), "W", W, "fp32 err/scale", err, "status", int(r.status.max()), "steps eq", float(((r.n_accept.cpu().numpy()==na_o)&(r.n_reject.cpu().numpy()==nr_o)).mean()))
assert err < 1e-5
big = synthetic.seirs_multi_strain(16384, seed=1, W=W)
for name, ts in (("daily", big.save_ts), ("2pts", np.array([0.0, 365.0]))):
    ms, am, ax = timeit(big, ts, 16384)
    print(f"   B=16384 save={name:5s} ms={ms:8.4f} traj/s={16384/ms*1e3:12.0f} out GB/s={16384*len(ts)*big.model.state_dim*4/ms/1e6:8.1f}")
