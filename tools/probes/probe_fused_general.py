"""dev probe: what the one-launch sampler iteration buys beyond the inference example -- the six-site 2-age x 3-strain model
(chains padded to eight trajectory rows), fused against two launches per iteration: wall time of the same run, same draws."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynode_amd.infer import folded
from dynode_amd.infer.inference import Potential, init_to_median
from dynode_amd.infer.nuts import KernelNUTS
from examples import infer_multi_strain as ex_m

chains = int(sys.argv[1]) if len(sys.argv) > 1 else 128
warm = draws = int(sys.argv[2]) if len(sys.argv) > 2 else 500
pot = Potential(ex_m.model, dict(config=ex_m.get_config(6), tf=120, obs_data=ex_m.synthetic_incidence(120)), 0, torch.device("cuda"))
z0 = pot.initial(chains, init_to_median, 3)
out = {}
for rep in range(2):
    for fuse in (True, False):
        f = folded.discover(pot)
        sampler = KernelNUTS(f, max_tree_depth=8, target_accept=0.8, seed=11, fuse=fuse)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = sampler.run(z0, warm, draws)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        launches = int(res.num_steps.sum(1).max()) if hasattr(res, "num_steps") else 0
        out[fuse] = res.samples
        print(f"rep {rep} fuse={fuse}: {dt:.3f} s, launches/iteration {sampler.launches_per_iteration}, "
              f"mean leapfrogs/transition {float(res.num_steps.double().mean()):.1f}", flush=True)
print("same draws:", bool(torch.equal(out[True], out[False])))
# the gradient-solve alone (HIP events), per launch
from dynode_amd import _abi, engine
f = folded.discover(pot)
f.map_now(z0)
for hints in ({}, {"replicas_log2": 1}, {"replicas_log2": 2}, {"replicas_log2": 3}):   # k + 1 = 2^k lane groups per trajectory
    with engine.dispatch_hints(**hints):
        for _ in range(5):
            f.solve_current(chains)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f.solve_current(chains)
        e1.record()
        torch.cuda.synchronize()
        print(hints, f"gradient-solve {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch, kernel {_abi.lib().dyn_last_kernel_name().decode()}", flush=True)
