import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic
from dynode_amd.engine import solve_batch
wl = synthetic.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]()
r = solve_batch(wl.model, wl.y0, wl.params, wl.contact, wl.t1, wl.save_ts)
torch.cuda.synchronize()
it = r.n_accept.cpu().numpy(); rd = r.n_reject.cpu().numpy()
print("wave iterations: mean %.1f max %d | save rounds per wave: mean %.1f max %d (n_save=%d)" % (it.mean(), it.max(), rd.mean(), rd.max(), wl.n_save))
