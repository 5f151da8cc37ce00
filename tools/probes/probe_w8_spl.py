"""dev probe: cfg 3 with the 8-stage waning chain (D = 360) at 1, 2 and 4 strains per lane
(32 / 16 / 8 lanes per trajectory).  SPL = 4 is not in instances.def: built here through the JIT hooks."""
import ctypes, os, subprocess, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import _abi, jit, synthetic
from dynode_amd.engine import solve_batch

wl = synthetic.WORKLOADS["cfg3"]()
m = wl.model
L = _abi.lib()
name = jit._name(m, torch.float32, 0, 0, 4)
so = os.path.join(jit._OUT, name + ".so")
os.makedirs(jit._OUT, exist_ok=True)
if not os.path.exists(so):
    src = so[:-3] + ".hip"
    open(src, "w").write(jit._source(m, torch.float32, 0, 0, 4))
    subprocess.run([jit.HIPCC, "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", src, "-o", so], check=True)
extra = ctypes.CDLL(so)
extra.dyn_extra_launch.restype = ctypes.c_void_p
assert L.dyn_register_instance(0, 0, 8, 4, 1, 1, 1, 8, 0, 4, 0, ctypes.c_void_p(extra.dyn_extra_launch())) == 0
a = [torch.as_tensor(x, dtype=torch.float32, device="cuda") for x in (wl.y0, wl.params, wl.contact, wl.save_ts)]
ref = None
for spl in (1, 2, 4):
    os.environ["DYNODE_HIP_SPL"] = str(spl)
    r = solve_batch(m, a[0], a[1], a[2], wl.t1, a[3])
    out, st = r.ys, (r.status, r.n_accept, r.n_reject)
    run = lambda: solve_batch(m, a[0], a[1], a[2], wl.t1, a[3], out=out, stats_out=st)
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gbs = wl.bytes_per_trajectory(4) * wl.B / ms / 1e6
    if ref is None: ref = out.clone()
    print(f"cfg3 SPL={spl} ms={ms:7.3f} traj/s={wl.B / ms * 1e3:10.0f} frac={gbs / 8000:.3f} ok={int(r.status.max()) == 0} "
          f"max |diff| vs SPL=1 = {float((out - ref).abs().max()):.3g}", flush=True)
    del out, r
