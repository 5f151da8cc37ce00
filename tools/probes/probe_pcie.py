"""Dev probe: PCIe-inclusive rate if a consumer wants the whole cfg-3 ensemble on the host."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic, sharding
from dynode_amd.engine import solve_batch
wl = synthetic.seirs_multi_strain(16384, seed=1)
dev = "cuda"
y0, params = (torch.as_tensor(a, dtype=torch.float32, device=dev) for a in (wl.y0, wl.params))
wl.y0, wl.params = y0, params          # inputs resident in HBM, as in bench.py
r = solve_batch(wl.model, wl.y0, wl.params, wl.contact, wl.t1, wl.save_ts)
torch.cuda.synchronize()
host = torch.empty(r.ys.shape, dtype=r.ys.dtype, pin_memory=True)
for _ in range(2):
    host.copy_(r.ys, non_blocking=True); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3):
    solve_batch(wl.model, wl.y0, wl.params, wl.contact, wl.t1, wl.save_ts, out=r.ys)
    host.copy_(r.ys, non_blocking=True)
    torch.cuda.synchronize()
el = (time.perf_counter() - t) / 3
gb = r.ys.numel() * 4 / 1e9
print(f"solve + D2H of {gb:.2f} GB to pinned host memory: {el*1e3:.1f} ms -> {16384/el:.3e} trajectories/s, D2H {gb/el:.1f} GB/s effective")
sharding.allreduce_ensemble_moments(r.ys); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3):
    solve_batch(wl.model, wl.y0, wl.params, wl.contact, wl.t1, wl.save_ts, out=r.ys)
    mean, var, n = sharding.allreduce_ensemble_moments(r.ys)
    m = mean.cpu()
torch.cuda.synchronize()
el = (time.perf_counter() - t) / 3
print(f"solve + on-device ensemble mean/variance + D2H of the 2x{mean.numel()} summary: {el*1e3:.2f} ms -> {16384/el:.3e} trajectories/s")
