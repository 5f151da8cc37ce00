"""dev probe: does the ORDER of the batch matter?  Longest-predicted-first ordering (LPT dispatch of waves, similar costs
paired in a wave) against the given order, with the previous launch's step counts and with a predictor from the rates.
Usage: python tools/probes/probe_sorted.py [workload ...]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic
from dynode_amd.engine import solve_batch


def timed(m, y0, p, C, t1, ts, reps=20):
    r = solve_batch(m, y0, p, C, t1, ts, dtype=torch.float32)
    out, st = r.ys, (r.status, r.n_accept, r.n_reject)
    for _ in range(3):
        solve_batch(m, y0, p, C, t1, ts, dtype=torch.float32, out=out, stats_out=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        solve_batch(m, y0, p, C, t1, ts, dtype=torch.float32, out=out, stats_out=st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, (r.n_accept + r.n_reject).double()


def run(name, B=None):
    wl = synthetic.WORKLOADS[name]() if B is None else synthetic.WORKLOADS[name](B)
    m, dev, f32 = wl.model, "cuda", torch.float32
    y0 = torch.as_tensor(wl.y0, dtype=f32, device=dev)
    p = torch.as_tensor(wl.params, dtype=f32, device=dev)
    C = torch.as_tensor(wl.contact, dtype=f32, device=dev)
    ts = torch.as_tensor(wl.save_ts, dtype=f32, device=dev)
    ms0, att = timed(m, y0, p, C, wl.t1, ts)
    print(f"{name} B={wl.B}: given order {ms0:.4f} ms; attempts mean {float(att.mean()):.1f} sd {float(att.std()):.1f}", flush=True)

    def with_order(order, label):
        yy = y0[order].contiguous() if y0.dim() == 2 else y0
        ms, a2 = timed(m, yy, p[order].contiguous(), C, wl.t1, ts)
        assert torch.equal(a2, att[order])
        print(f"   {label:34s} {ms:.4f} ms  ({ms0 / ms - 1:+.3f})", flush=True)

    with_order(torch.argsort(att, descending=True), "previous step counts, descending")
    with_order(torch.argsort(att, descending=False), "previous step counts, ascending")
    # predictor from the rates: least squares on sorted per-strain rates, fitted on another seed of the same generator
    S = m.n_strain
    def feats(P):
        P = P.double()
        cols = [torch.sort(P[:, i * S:(i + 1) * S], dim=1).values for i in range(min(4, P.shape[1] // S))]
        return torch.cat([torch.ones(P.shape[0], 1, dtype=torch.float64, device=P.device)] + cols + [P[:, 4 * S:]], dim=1)
    half = wl.B // 2
    X = feats(p)
    w = torch.linalg.lstsq(X[:half], att[:half, None]).solution
    pred = (X @ w)[:, 0]
    print(f"   predictor correlation (held out) {float(torch.corrcoef(torch.stack([pred[half:], att[half:]]))[0, 1]):.3f}")
    with_order(torch.argsort(pred, descending=True), "rate predictor, descending")
    lp = torch.log(p.double().clamp_min(1e-30))
    iu = torch.triu_indices(lp.shape[1], lp.shape[1], device=dev)
    Q = torch.cat([torch.ones(wl.B, 1, dtype=torch.float64, device=dev), lp, lp[:, iu[0]] * lp[:, iu[1]]], dim=1)
    mu, sd = Q[:half].mean(0), Q[:half].std(0) + 1e-12
    mu[0], sd[0] = 0.0, 1.0
    Qn = (Q - mu) / sd
    A = Qn[:half].T @ Qn[:half] + 1e-6 * half * torch.eye(Qn.shape[1], dtype=torch.float64, device=dev)
    wq = torch.linalg.solve(A, Qn[:half].T @ att[:half])
    predq = Qn @ wq
    print(f"   quadratic-in-log predictor: {Qn.shape[1]} features, correlation (held out) {float(torch.corrcoef(torch.stack([predq[half:], att[half:]]))[0, 1]):.3f}")
    with_order(torch.argsort(predq, descending=True), "quadratic predictor, descending")
    with_order(torch.argsort(torch.round(predq), descending=True, stable=True), "quadratic predictor, rounded keys")
    with_order(torch.randperm(wl.B, device=dev), "random permutation")
    with_order(torch.arange(wl.B, device=dev), "given order again")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        o = torch.argsort(predq.float(), descending=True).int()
    e1.record()
    torch.cuda.synchronize()
    print(f"   torch.argsort of {wl.B} float keys: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")


if __name__ == "__main__":
    for n in sys.argv[1:] or ["cfg3", "cfg3d136", "cfg5", "cfg2", "seip", "seip83"]:
        run(n)
    if not sys.argv[1:]:
        run("cfg3", 65536)
