"""dev probe: one fuzz seed in detail."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/probes/x.py -> repo root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from dynode_amd.engine import solve_batch
from test_gpu_parity import fuzz_case
O = H.O
m, y0, p, C, t1, ts, kw = fuzz_case(int(sys.argv[1]))
print(m, "B", p.shape[0], "t1", t1, "ts", ts, kw)
r = solve_batch(m, y0, p, C, t1, ts, dtype=torch.float64, **kw); torch.cuda.synchronize()
want, st, na, nr = O.solve(H.omodel(m), y0, p, C, t1, ts, dtype=np.float64, n_threads=8, **kw)
got = r.ys.cpu().numpy()
print("status hip", np.unique(r.status.cpu().numpy(), return_counts=True), "oracle", np.unique(st, return_counts=True))
d_acc = r.n_accept.cpu().numpy() - na; d_rej = r.n_reject.cpu().numpy() - nr
print("trajectories with different counts:", np.nonzero((d_acc != 0) | (d_rej != 0))[0])
for b in np.nonzero((d_acc != 0) | (d_rej != 0) | ~np.isfinite(got).all((1, 2)) | ~np.isfinite(want).all((1, 2)))[0][:5]:
    print("traj", b, "hip acc/rej", int(r.n_accept[b]), int(r.n_reject[b]), "oracle", na[b], nr[b], "status", int(r.status[b]), st[b])
    print("  hip finite rows", np.isfinite(got[b]).all(1).astype(int), "\n  ora finite rows", np.isfinite(want[b]).all(1).astype(int))
    print("  max |diff| per row", np.abs(np.where(np.isfinite(got[b]) & np.isfinite(want[b]), got[b] - want[b], 0)).max(1))
if len(sys.argv) > 2:
    b = int(sys.argv[2])
    yb = y0[b] if np.ndim(y0) == 2 else y0
    grid = np.unique(np.concatenate([np.linspace(0, t1, 731), kw.get("jump_ts", [])]))
    kw2 = dict(kw)
    r1 = solve_batch(m, yb, p[b:b + 1], C, t1, grid, dtype=torch.float64, **kw2); torch.cuda.synchronize()
    w1, s1, a1, j1 = O.solve(H.omodel(m), yb, p[b:b + 1], C, t1, grid, dtype=np.float64, **kw2)
    g1 = r1.ys.cpu().numpy()[0]
    fin = np.isfinite(g1).all(1)
    print("single trajectory: hip status", int(r1.status[0]), "acc/rej", int(r1.n_accept[0]), int(r1.n_reject[0]), "| oracle", s1[0], a1[0], j1[0])
    print("last finite save time (hip):", grid[fin][-1] if fin.any() else None, " first non-finite:", grid[~fin][0] if (~fin).any() else None)
    k = np.nonzero(fin)[0][-1] if fin.any() else 0
    print("state there hip", g1[k], "oracle", w1[0][k], "\nparams", p[b], "y0", yb)
    for ms in range(1, 16):
        r2 = solve_batch(m, yb, p[b:b + 1], C, t1, grid, dtype=torch.float64, **{**kw2, "max_steps": ms}); torch.cuda.synchronize()
        w2, s2, a2, j2 = O.solve(H.omodel(m), yb, p[b:b + 1], C, t1, grid, dtype=np.float64, **{**kw2, "max_steps": ms})
        gh, go = r2.ys.cpu().numpy()[0], w2[0]
        th = grid[np.isfinite(gh).all(1)][-1] if np.isfinite(gh).all(1).any() else -1
        to = grid[np.isfinite(go).all(1)][-1] if np.isfinite(go).all(1).any() else -1
        print(f"max_steps {ms:2d}: hip status {int(r2.status[0])} acc {int(r2.n_accept[0])} rej {int(r2.n_reject[0])} reached ~{th:.4f} | oracle status {s2[0]} acc {a2[0]} rej {j2[0]} reached ~{to:.4f}")
