"""dev probe: the most negative value of the D = 2496 bench shape at B = 768 -- HIP (plain / general instance) against the float32 and
float64 oracle on the same trajectory."""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import torch
from dynode_amd import synthetic
from dynode_amd.engine import solve_batch
import helpers as H
from helpers import O
wl = synthetic.WORKLOADS["seip83"](768)
m = wl.model
for flag in ("1", "0"):
    pass  # (round 4: pin the instance with engine.dispatch_hints(general_instance=1) instead of the old environment switch)
    r = solve_batch(m, wl.y0, wl.params, wl.contact, wl.t1, wl.save_ts)
    ys = r.ys
    mn = ys.reshape(768, -1).min(1).values
    b = int(mn.argmin())
    print("plain" if flag == "1" else "general", "min", float(mn.min()), "trajectory", b, "second", float(mn.sort().values[1]), flush=True)
b = 235
truth, st, _, _ = O.solve(H.omodel(m), wl.y0[b:b + 1], wl.params[b:b + 1], wl.contact, wl.t1, wl.save_ts, dtype=np.float64, n_threads=8, rtol=1e-9, atol=1e-9)
o32, st, na, nr = O.solve(H.omodel(m), wl.y0[b:b + 1], wl.params[b:b + 1], wl.contact, wl.t1, wl.save_ts, dtype=np.float32, n_threads=8)
print("float32 oracle: min", o32.min(), "max err", np.abs(o32 - truth).max(), "steps", na, nr)
for flag in ("1", "0"):
    pass  # (round 4: pin the instance with engine.dispatch_hints(general_instance=1) instead of the old environment switch)
    r = solve_batch(m, wl.y0[b:b + 1], wl.params[b:b + 1], wl.contact, wl.t1, wl.save_ts)
    y = r.ys[0].cpu().numpy()
    k = np.unravel_index(y.argmin(), y.shape)
    err = np.abs(y - truth[0])
    ke = np.unravel_index(err.argmax(), err.shape)
    print("plain" if flag == "1" else "general", "min", y.min(), "at (row, column)", k, "truth there", truth[0][k], "rows around", y[max(k[0] - 2, 0):k[0] + 3, k[1]],
          "max err", err.max(), "at", ke, "steps", int(r.n_accept[0]), int(r.n_reject[0]), flush=True)
