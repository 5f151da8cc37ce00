"""dev probe: launch time with the learned dispatch order (schedule.py), overhead included, against the given order.
The forecast is trained on a different seed of the workload than the one that is timed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic, schedule
from dynode_amd.engine import solve_batch


def dev(wl):
    f32 = torch.float32
    return (torch.as_tensor(wl.y0, dtype=f32, device="cuda"), torch.as_tensor(wl.params, dtype=f32, device="cuda"),
            torch.as_tensor(wl.contact, dtype=f32, device="cuda"), torch.as_tensor(wl.save_ts, dtype=f32, device="cuda"))


def timed(m, y0, p, C, t1, ts, order, reps=20):
    r = solve_batch(m, y0, p, C, t1, ts, order=order)
    out, st = r.ys, (r.status, r.n_accept, r.n_reject)
    for _ in range(3):
        solve_batch(m, y0, p, C, t1, ts, out=out, stats_out=st, order=order)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        solve_batch(m, y0, p, C, t1, ts, out=out, stats_out=st, order=order)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, r


def run(name, B=None):
    make = synthetic.WORKLOADS[name]
    wl = make() if B is None else make(B)
    big = make(2 * wl.B)          # same generator and seed: same shared constants (age shares, contact matrix), other draws
    other = True
    m = wl.model
    y0, p, C, ts = dev(wl)
    schedule.reset()
    t_none, r0 = timed(m, y0, p, C, wl.t1, ts, None)
    yo, po, Co, tso = dev(big)
    assert torch.equal(Co, C) and not torch.equal(po[wl.B:], p)
    yo = yo[wl.B:].contiguous() if yo.dim() == 2 else yo
    for _ in range(3):
        solve_batch(m, yo, po[wl.B:].contiguous(), Co, big.t1, tso, order="forecast")          # order="forecast": trains on draws that are not timed
    cm = next(iter(schedule._MODELS.values()))
    if not cm.ready:
        print(f"{name}: forecast not used (R2 below {schedule.MIN_R2})", flush=True)
        return
    att = (r0.n_accept + r0.n_reject).double()
    corr = float(torch.corrcoef(torch.stack([cm.forecast(p), att]))[0, 1]) if cm.ready else float("nan")
    # interleaved A/B (clocks ramp and drift inside one process: single measurements of sub-millisecond launches mislead)
    ta, tn = [], []
    for _ in range(5):
        t, r1 = timed(m, y0, p, C, wl.t1, ts, "forecast")
        ta.append(t)
        tn.append(timed(m, y0, p, C, wl.t1, ts, None)[0])
    ta.sort(); tn.sort()
    t_auto, t_none2 = ta[2], tn[2]
    t_none = tn[0]
    same = torch.equal(r0.ys, r1.ys) and torch.equal(r0.n_accept, r1.n_accept) and torch.equal(r0.status, r1.status)
    print(f"{name:9s} B={wl.B:6d} given median {t_none2:.4f} (min {t_none:.4f}) ms   ordered median {t_auto:.4f} (min {ta[0]:.4f}) ms ({t_none2 / t_auto - 1:+.3f})  "
          f"forecast corr {corr:.3f} features {0 if cm.cols is None else int(cm.cols.numel())} trained on other draws  bit-identical {same}", flush=True)


if __name__ == "__main__":
    for n in sys.argv[1:] or ["cfg3", "cfg3d136", "cfg5", "cfg2", "seip"]:
        run(n)
    if not sys.argv[1:]:
        run("cfg3", 65536)
