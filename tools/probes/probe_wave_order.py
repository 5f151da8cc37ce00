"""dev probe: with trajectories grouped into waves by exact step count, does the ORDER OF THE WAVES matter?
(one residency round at D = 136, B = 16384: 2048 waves on 2048 wave slots)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic, _abi
from dynode_amd.engine import solve_batch


def timed(m, y0, p, C, t1, ts, order, reps=20):
    r = solve_batch(m, y0, p, C, t1, ts, order=order)
    out, st = r.ys, (r.status, r.n_accept, r.n_reject)
    res = []
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            solve_batch(m, y0, p, C, t1, ts, out=out, stats_out=st, order=order)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / reps)
    return min(res), r


for name in sys.argv[1:] or ["cfg3d136", "cfg3"]:
    wl = synthetic.WORKLOADS[name]()
    m = wl.model
    f32 = torch.float32
    y0, p, C, ts = [torch.as_tensor(x, dtype=f32, device="cuda") for x in (wl.y0, wl.params, wl.contact, wl.save_ts)]
    t0, r = timed(m, y0, p, C, wl.t1, ts, None)
    att = (r.n_accept + r.n_reject).double()
    tpw = _abi.lib().dyn_trajectories_per_wave(__import__("ctypes").byref(m.c()))
    desc = torch.argsort(att, descending=True, stable=True)
    waves = desc.reshape(-1, tpw)                    # [n_waves, tpw], heaviest wave first
    nw = waves.shape[0]
    variants = {"given": None, "descending": waves}
    half = nw // 2
    variants["snake (2nd half ascending)"] = torch.cat([waves[:half], waves[half:].flip(0)])
    variants["interleaved heavy/light"] = torch.stack([waves[:half], waves[half:].flip(0)], dim=1).reshape(nw, tpw)
    variants["ascending"] = waves.flip(0)
    quarter = nw // 4
    variants["snake over 4 quarters"] = torch.cat([waves[:quarter], waves[quarter:2 * quarter].flip(0), waves[2 * quarter:3 * quarter], waves[3 * quarter:].flip(0)])
    print(f"{name}: {nw} waves of {tpw} trajectories")
    for label, w in variants.items():
        o = None if w is None else w.reshape(-1).to(torch.int32).contiguous()
        t, r2 = timed(m, y0, p, C, wl.t1, ts, o)
        assert torch.equal(r2.ys, r.ys)
        print(f"   {label:28s} {t:.4f} ms ({t0 / t - 1:+.3f})", flush=True)
