"""The batched NUTS sampler on analytic targets (pure torch, runs on CPU)."""

import math

import numpy as np
import pytest
import torch
from scipy import stats

from dynode_amd.infer.nuts import BatchedNUTS, GraphNUTS, LockstepNUTS, _adaptation_windows


def gaussian_target(cov):
    prec = torch.linalg.inv(cov)

    def pg(z):
        g = z @ prec
        return 0.5 * (z * g).sum(-1), g
    return pg


def test_adaptation_windows_follow_stans_schedule():
    assert _adaptation_windows(1000) == [(75, 100), (100, 150), (150, 250), (250, 450), (450, 950)]
    w = _adaptation_windows(500)
    assert w[0][0] == 75 and w[-1][1] == 450 and all(a[1] == b[0] for a, b in zip(w, w[1:]))
    assert _adaptation_windows(10) == []


@pytest.mark.parametrize("sampler", [BatchedNUTS, LockstepNUTS, GraphNUTS], ids=["async", "lockstep", "graph-step"])
def test_correlated_gaussian_moments_and_marginals(sampler):
    torch.manual_seed(0)
    cov = torch.tensor([[4.0, 1.8], [1.8, 1.0]], dtype=torch.float64)    # strongly correlated, unequal scales
    nuts = sampler(gaussian_target(cov), max_tree_depth=8, seed=1)
    res = nuts.run(torch.randn(32, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(23)), num_warmup=300, num_samples=300)
    x = res.samples.reshape(-1, 2)
    assert res.samples.shape == (32, 300, 2) and int(res.diverging.sum()) == 0
    assert 0.6 < float(res.accept_prob.mean()) < 0.95                      # dual averaging towards 0.8
    assert torch.allclose(x.mean(0), torch.zeros(2, dtype=torch.float64), atol=0.08)
    assert torch.allclose(torch.cov(x.T), cov, rtol=0.12, atol=0.08)
    # adapted dense mass matrix ~ target covariance
    assert torch.allclose(res.inverse_mass.mean(0), cov, rtol=0.35, atol=0.3)
    # thinned marginals pass a KS test against the exact normal
    for d, sd in ((0, 2.0), (1, 1.0)):
        thin = res.samples[:, ::10, d].reshape(-1).numpy()
        assert stats.kstest(thin, "norm", args=(0.0, sd)).pvalue > 1e-3
    assert res.potential_evals > 300 and int(res.num_steps.max()) <= 2 ** 8


def test_chains_are_independent_and_reproducible():
    cov = torch.eye(3, dtype=torch.float64)
    z0 = torch.zeros(4, 3, dtype=torch.float64)
    a = BatchedNUTS(gaussian_target(cov), max_tree_depth=6, seed=5).run(z0, 60, 40)
    b = BatchedNUTS(gaussian_target(cov), max_tree_depth=6, seed=5).run(z0, 60, 40)
    assert torch.equal(a.samples, b.samples)
    assert not torch.equal(a.samples[0], a.samples[1])


def test_tree_depth_limit_and_divergence_flag():
    # a funnel-like potential with huge curvature far out: big initial steps must be flagged
    def pg(z):
        u = 0.25 * (z ** 4).sum(-1)
        return u, z ** 3
    res = BatchedNUTS(pg, max_tree_depth=3, seed=0).run(torch.full((8, 1), 0.5, dtype=torch.float64), 50, 50)
    assert int(res.num_steps.max()) <= 2 ** 3 - 0 and torch.isfinite(res.samples).all()
    thin = res.samples[:, ::5, 0].reshape(-1).numpy()
    # density ~ exp(-x^4/4): symmetric, |x| rarely beyond 2.5
    assert abs(np.mean(thin)) < 0.25 and np.mean(np.abs(thin) > 2.5) < 0.02


def test_async_needs_far_fewer_gradient_solves_than_lockstep():
    cov = torch.tensor([[4.0, 1.8], [1.8, 1.0]], dtype=torch.float64)
    z0 = torch.randn(64, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    a = BatchedNUTS(gaussian_target(cov), max_tree_depth=8, seed=2).run(z0, 150, 100)
    b = LockstepNUTS(gaussian_target(cov), max_tree_depth=8, seed=2).run(z0, 150, 100)
    per_chain = float(a.num_steps.double().mean())
    assert a.potential_evals < 0.6 * b.potential_evals
    assert a.potential_evals < 250 * (per_chain + 1) * 2.0          # ~ leapfrogs of the slowest chain, not the sum of maxima


def test_the_numpy_twin_of_the_sampler_kernel_has_the_published_philox_stream():
    """tests/nuts_twin.py (the restatement the GPU suite holds `dyn_nuts_advance` against, launch by launch): its
    Philox4x32-10 on the known-answer vectors of the Random123 distribution, its uniforms in (0, 1) with 52 random bits,
    and one fabricated chain through a full short run (a window end and recorded draws included) without leaving the reals."""
    import math

    import nuts_twin

    assert nuts_twin.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert nuts_twin.philox4x32_10((0xffffffff,) * 4, (0xffffffff,) * 2) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert nuts_twin.philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == (
        0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)
    s = nuts_twin.Stream(0x0123456789abcdef, 5, 3)
    u = [s.uniform() for _ in range(2000)]
    assert s.ctr == 2005 and 0.0 < min(u) and max(u) < 1.0 and abs(np.mean(u) - 0.5) < 0.02
    C, D, Dm = 2, 3, 4
    f = lambda *sh: np.zeros(sh)  # noqa: E731
    S = dict(z=f(C, D), u=f(C), g=f(C, D), eps=np.full(C, 0.1), eps_avg=np.full(C, 0.1), da_mu=f(C), da_xbar=f(C), da_gbar=f(C), da_t=f(C),
             imm=np.tile(np.eye(D), (C, 1, 1)), mm_sqrt=np.tile(np.eye(D), (C, 1, 1)), wf_n=f(C), wf_mean=f(C, D), wf_m2=f(C, D, D), e0=f(C),
             zl=f(C, D), rl=f(C, D), gl=f(C, D), zr=f(C, D), rr=f(C, D), gr=f(C, D), zp=f(C, D), up=f(C), gp=f(C, D), weight=f(C),
             r_sum=f(C, D), sum_acc=f(C), sgn=np.ones(C), zc=f(C, D), rc=f(C, D), gc=f(C, D), r_half=np.ones((C, D)), s_zp=f(C, D),
             s_up=f(C), s_gp=f(C, D), s_weight=np.full(C, -math.inf), s_rsum=f(C, D), s_acc=f(C), r_ck=f(C, Dm, D), rs_ck=f(C, Dm, D),
             z_eval=f(C, D), it=np.zeros(C, int), wi=np.zeros(C, int), n_prop=np.zeros(C, int), depth=np.zeros(C, int),
             right=np.ones(C, int), leaf=np.zeros(C, int), s_turn=np.zeros(C, int), s_div=np.zeros(C, int), s_n=np.zeros(C, int),
             rng_ctr=np.zeros(C, np.int64), out_z=f(C, 2, D), out_acc=f(C, 2), out_n=np.zeros((C, 2), int), out_div=np.zeros((C, 2), int))
    K = dict(seed=123456789012345, num_warmup=3, num_samples=2, max_depth=Dm, target_accept=0.8, max_delta_energy=1000.0, windows=[(1, 3)])
    for _ in range(400):
        nuts_twin.advance(S, K, 0.5 * (S["z_eval"] ** 2).sum(-1), S["z_eval"].copy())
    assert S["it"].tolist() == [5, 5] and S["wi"].tolist() == [1, 1] and all(np.isfinite(v).all() for k, v in S.items() if k != "s_weight")
    assert np.abs(S["out_z"]).max() > 0 and (S["out_n"] > 0).all() and not np.array_equal(S["imm"][0], np.eye(D))
