"""dev probe (CPU, oracle): lock-step cost of a static grid -- mean over waves of the most step attempts among a wave's trajectories over the
mean attempts -- in the given order, after an exact sort, and after sorting by training-free proxies (parameter sums / maxima)."""
import sys, numpy as np
import os; root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
from dynode_amd import synthetic
import helpers as H
from helpers import O
for name, tpw in (("cfg3d136", 8), ("cfg3", 2), ("cfg5", 8)):
    wl = synthetic.WORKLOADS[name](4096) if name != "cfg5" else synthetic.WORKLOADS[name](4096)
    m = wl.model
    ys, st, na, nr = O.solve(H.omodel(m), wl.y0, wl.params, wl.contact, wl.t1, wl.save_ts, dtype=np.float32, n_threads=8)
    att = (na + nr).astype(np.float64)
    P = wl.params
    print(name, "P", P.shape, "attempts mean", att.mean(), "sd", att.std())
    def waste(order):
        a = att[order].reshape(-1, tpw)
        return a.max(1).mean() / att.mean()
    idx = np.arange(len(att))
    print("  given order: iterations/mean attempts =", round(waste(idx), 4), " exact sort:", round(waste(np.argsort(-att)), 4))
    # proxies: each single column, sum of all, and the best linear combo (for reference only)
    best = []
    for j in range(P.shape[1]):
        c = np.corrcoef(P[:, j], att)[0, 1]
        best.append((abs(c), j, c))
    best.sort(reverse=True)
    print("  top columns by |corr|:", [(j, round(c, 3)) for _, j, c in best[:6]])
    s = P.sum(1)
    print("  sum of all params: corr", round(np.corrcoef(s, att)[0, 1], 3), "waste", round(waste(np.argsort(-s)), 4))
    S = m.n_strain
    for lab, cols in (("beta", slice(0, S)), ("gamma", slice(S, 2 * S)), ("sigma", slice(2 * S, 3 * S)), ("omega", slice(3 * S, 4 * S))):
        try:
            v = P[:, cols].sum(1); print("  sum", lab, "corr", round(np.corrcoef(v, att)[0, 1], 3), "waste", round(waste(np.argsort(-v)), 4), " max:", round(waste(np.argsort(-P[:, cols].max(1))), 4))
        except Exception as e: print(lab, e)
    A = np.c_[P, np.ones(len(att))]
    coef, *_ = np.linalg.lstsq(A, att, rcond=None)
    pred = A @ coef
    print("  in-sample linear fit: R2", round(1 - ((att - pred) ** 2).sum() / ((att - att.mean()) ** 2).sum(), 3), "waste", round(waste(np.argsort(-pred)), 4))
