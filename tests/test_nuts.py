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
