"""Dev probe: does sorting the batch by a cheap dynamics proxy reduce lock-step imbalance?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dynode_amd import synthetic
from probe_perf import timeit
wl = synthetic.seirs_multi_strain(16384, seed=1)
p = wl.params
S = 4
beta, gamma, sigma, omega = p[:, :S], p[:, S:2*S], p[:, 2*S:3*S], p[:, 3*S:4*S]
keys = {
    "none": None,
    "max_growth": (beta - gamma).max(1),
    "sum_beta": beta.sum(1),
    "max_r0": (beta / gamma).max(1),
    "growth_seir": np.max(0.5 * (-(sigma + gamma) + np.sqrt((sigma - gamma) ** 2 + 4 * sigma * beta)), axis=1),
    "lexi(maxstrain,growth)": None,
}
g = 0.5 * (-(sigma + gamma) + np.sqrt((sigma - gamma) ** 2 + 4 * sigma * beta))
keys["lexi(maxstrain,growth)"] = g.argmax(1) * 10.0 + g.max(1)
import copy
for name, key in keys.items():
    w2 = copy.copy(wl)
    if key is not None:
        o = np.argsort(key)
        w2.params = wl.params[o]; w2.y0 = wl.y0[o]
    ms, am, ax = timeit(w2, wl.save_ts, 16384, reps=20)
    ms2, _, _ = timeit(w2, np.array([0.0, 365.0]), 16384, reps=20)
    print(f"sort={name:24s} daily ms={ms:.4f}  2pts ms={ms2:.4f}")
