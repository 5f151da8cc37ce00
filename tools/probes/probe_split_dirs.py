"""dev probe: gradient-solve of cfg 4 (128 chains, 2-age SIR, fused likelihood) with both tangent directions in one
trajectory (ND = 2, B = 128) against one direction per trajectory (ND = 1, B = 256, every chain twice)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd import synthetic
from dynode_amd.engine import solve_batch_loglik

wl = synthetic.sir_two_age_literal(100.0)
C = 128
g = torch.Generator().manual_seed(0)
r0 = 1.6 + 0.8 * torch.rand(C, generator=g, dtype=torch.float64)
ti = 6.0 + 3.0 * torch.rand(C, generator=g, dtype=torch.float64)
params = torch.stack([r0 / ti, 1.0 / ti], 1).float().cuda()
seeds = torch.randn((C, 2, 2), generator=g).float().cuda()
obs = (torch.rand((100, 2), generator=g) * 5 + 1).float().cuda()
ts = torch.arange(101.0).float().cuda()
y0 = torch.as_tensor(wl.y0).float().cuda()
Cm = torch.as_tensor(wl.contact).float().cuda()


def timed(p, s, reps=200):
    for _ in range(10):
        out = solve_batch_loglik(wl.model, y0, p, Cm, 100.0, ts, obs, 2, dparams=s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = solve_batch_loglik(wl.model, y0, p, Cm, 100.0, ts, obs, 2, dparams=s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, out


t2, o2 = timed(params, seeds)
p1 = params.repeat(2, 1).contiguous()
s1 = torch.cat([seeds[:, 0:1], seeds[:, 1:2]], 0).contiguous()
t1, o1 = timed(p1, s1)
print(f"ND=2 B={C}: {t2:.1f} us   ND=1 B={2 * C}: {t1:.1f} us")
print("ll equal", bool(torch.equal(o2[0], o1[0][:C])), "max |dll diff|", float((torch.stack([o1[1][:C, 0], o1[1][C:, 0]], 1) - o2[1]).abs().max()))
