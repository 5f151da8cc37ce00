"""The affine-invariant ensemble sampler (dynode_amd/infer/ensemble.py) on analytic targets, on the CPU."""

import numpy as np
import pytest
import torch

from dynode_amd.infer.ensemble import EnsembleSampler


def test_correlated_gaussian_moments():
    cov = torch.tensor([[1.0, 0.9 * 3.0], [0.9 * 3.0, 9.0]], dtype=torch.float64)
    prec = torch.linalg.inv(cov)
    mean = torch.tensor([2.0, -1.0], dtype=torch.float64)
    logp = lambda z: -0.5 * torch.einsum("ci,ij,cj->c", z - mean, prec, z - mean)
    s = EnsembleSampler(logp, seed=3)
    res = s.run(torch.zeros((64, 2), dtype=torch.float64), 500, 1500)          # identical starts are spread by the sampler
    x = res.samples.reshape(-1, 2)
    assert res.samples.shape == (64, 1500, 2) and s.evals == 1 + 2 * 2000
    assert torch.allclose(x.mean(0), mean, atol=0.15)
    assert torch.allclose(torch.cov(x.T), cov, rtol=0.15, atol=0.15)
    assert 0.5 < float(res.accept_prob.mean()) < 0.9                            # stretch a = 2 in two dimensions


def test_affine_invariance_and_rejection_of_failed_points():
    """A badly scaled target mixes as well as a round one (that is the point of the stretch move); points where
    the density is NaN / -inf (failed solves) are never accepted."""
    scale = torch.tensor([1e-3, 1e3], dtype=torch.float64)

    def logp(z):
        lp = -0.5 * ((z / scale) ** 2).sum(-1)
        return torch.where(z[:, 0] > 2e-3, torch.full_like(lp, float("nan")), lp)       # a forbidden half-space

    s = EnsembleSampler(logp, seed=5)
    z0 = torch.randn((40, 2), dtype=torch.float64, generator=torch.Generator().manual_seed(3)) * scale * 0.1    # (own stream: not whatever the tests before left of the global one)
    res = s.run(z0, 300, 700)
    x = res.samples.reshape(-1, 2)
    assert float(x[:, 0].max()) <= 2e-3
    assert abs(float(x[:, 1].std()) / 1e3 - 1.0) < 0.2
    # truncated normal on (-inf, 2 sd]: mean = -phi(2) / Phi(2) = -0.05525 sd
    assert abs(float(x[:, 0].mean()) / 1e-3 + 0.05525) < 0.08


def test_argument_checks():
    s = EnsembleSampler(lambda z: -(z ** 2).sum(-1))
    with pytest.raises(ValueError, match="even number of walkers"):
        s.run(torch.zeros((5, 2), dtype=torch.float64), 1, 1)
    with pytest.raises(ValueError, match="2 D \\+ 2"):
        s.run(torch.zeros((4, 3), dtype=torch.float64), 1, 1)
    with pytest.raises(RuntimeError, match="finite density"):
        EnsembleSampler(lambda z: torch.full((z.shape[0],), float("nan"), dtype=z.dtype)).run(torch.zeros((8, 2), dtype=torch.float64), 1, 1)


def test_diagnostics_on_known_chains():
    """Effective sample size and split R-hat (infer/diagnostics.py, the two extra columns of print_summary)."""
    from dynode_amd.infer.diagnostics import effective_sample_size, split_rhat

    rng = np.random.default_rng(0)
    iid = rng.normal(size=(8, 1000))
    assert abs(split_rhat(iid) - 1.0) < 0.01 and 0.85 * 8000 < effective_sample_size(iid) < 1.1 * 8000
    x, e = np.zeros((8, 4000)), rng.normal(size=(8, 4000))
    for t in range(1, 4000):                                   # AR(1), rho = 0.9: integrated autocorrelation time 19
        x[:, t] = 0.9 * x[:, t - 1] + e[:, t]
    assert abs(effective_sample_size(x) / (8 * 4000 / 19.0) - 1.0) < 0.2 and split_rhat(x) < 1.02
    shifted = iid + np.arange(8)[:, None]                      # chains that disagree
    assert split_rhat(shifted) > 2.0 and effective_sample_size(shifted) < 20


def test_finite_difference_gradient_of_the_potential_on_the_cpu():
    """Potential.potential_and_grad_fd against autograd on a model without any ODE (runs on the CPU): two latent
    sites with constrained supports, a Gaussian likelihood."""
    from dynode_amd.infer import distributions as dist
    from dynode_amd.infer import handlers
    from dynode_amd.infer.inference import Potential, _FiniteDifferenceLogJoint

    obs = torch.tensor([1.2, 0.7, 1.9, 1.4], dtype=torch.float64)

    def model(obs_data):
        loc = handlers.sample("loc", dist.Uniform(-3.0, 3.0))
        scale = handlers.sample("scale", dist.TransformedDistribution(dist.Beta(2.0, 2.0), dist.transforms.AffineTransform(0.2, 2.0)))
        handlers.sample("y", dist.Normal(loc[..., None], scale[..., None]), obs=obs_data)

    pot = Potential(model, dict(obs_data=obs), 0, torch.device("cpu"))
    z = torch.tensor([[0.3, -0.2], [-1.0, 0.8], [1.5, 0.1]], dtype=torch.float64)
    u, g = pot.potential_and_grad(z)
    u_fd, g_fd = pot.potential_and_grad_fd(z, 1e-6)
    assert torch.allclose(u, u_fd, rtol=1e-13) and torch.allclose(g, g_fd, rtol=1e-6, atol=1e-7)
    zz = z.clone().requires_grad_(True)
    lj = _FiniteDifferenceLogJoint.apply(zz, pot, 1e-6)
    (weights,) = torch.autograd.grad((lj * torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64)).sum(), zz)
    assert torch.allclose(weights, -g * torch.tensor([[1.0], [2.0], [3.0]], dtype=torch.float64), rtol=1e-6, atol=1e-7)
