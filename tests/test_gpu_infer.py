"""cfg 4 on the GPU: the numpyro-style model, its gradient, and NUTS posteriors.

Posterior parity against numpyro's NUTS is unpinned (numpyro absent); the checker is exact
quadrature of the 2-parameter posterior on a fine grid (one batched solve), against which the
NUTS marginals must pass a KS test -- the north star's "KS-test agreement on posteriors".
"""

import numpy as np
import pytest
import torch
from scipy import stats

from dynode_amd.infer import handlers
from dynode_amd.infer.inference import MCMCProcess, Potential, log_posterior_grid
from dynode_amd.simulation import odes
from examples import sir_infer_parameters as ex

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def data():
    return ex.synthetic_incidence(100)


def test_model_sites_and_synthetic_data(data):
    assert data.shape == (100, 2) and float(data.min()) > 0
    with handlers.seed(0), handlers.trace() as tr:
        ex.model(ex.get_config(), 100, data)
    assert list(tr.sites) == ["strains_0_r0", "strains_0_infectious_period", "inf_incidence"]
    assert tr.sites["inf_incidence"]["is_observed"]


def test_potential_gradient_matches_finite_differences(data):
    odes.enable_x64(True)
    try:
        pot = Potential(ex.model, dict(config=ex.get_config(), tf=100, obs_data=data), 0, torch.device("cuda"))
        assert list(pot.latent) == ["strains_0_r0", "strains_0_infectious_period"] and pot.dim == 2
        z = torch.tensor([[0.1, -0.3], [-0.8, 0.5], [1.2, 0.2]], dtype=torch.float64, device="cuda")
        u, g = pot.potential_and_grad(z)
        eps = 1e-5
        for d in range(2):
            dz = torch.zeros_like(z); dz[:, d] = eps
            up, _ = pot.potential_and_grad(z + dz)
            um, _ = pot.potential_and_grad(z - dz)
            fd = (up - um) / (2 * eps)
            assert torch.allclose(g[:, d], fd, rtol=2e-4, atol=1e-4), (g[:, d], fd)
        # batching does not change a chain's value
        u1, g1 = pot.potential_and_grad(z[1:2])
        assert torch.allclose(u1, u[1:2], rtol=1e-10) and torch.allclose(g1, g[1:2], rtol=1e-8)
    finally:
        odes.enable_x64(False)


def _grid_marginals(data):
    odes.enable_x64(True)
    try:
        pot = Potential(ex.model, dict(config=ex.get_config(), tf=100, obs_data=data), 0, torch.device("cuda"))
        r0 = torch.linspace(1.5 + 1e-4, 2.5 - 1e-4, 401, dtype=torch.float64)   # the prior's whole support
        ti = torch.linspace(4.5, 10.5, 481, dtype=torch.float64)
        lp = log_posterior_grid(pot, [r0, ti]).cpu()
    finally:
        odes.enable_x64(False)
    p = torch.exp(lp - lp.max())
    p = p / p.sum()
    # r0 spans the prior's whole support; only the infectious-period box can cut mass off
    assert float(p[:, 0].sum() + p[:, -1].sum()) < 1e-6 and float(p[0].sum() + p[-1].sum()) < 5e-3
    return (r0.numpy(), np.cumsum(p.sum(1).numpy())), (ti.numpy(), np.cumsum(p.sum(0).numpy()))


def test_nuts_posterior_matches_grid_quadrature(data):
    process = MCMCProcess(numpyro_model=ex.model, num_warmup=250, num_samples=250, num_chains=48,
                          nuts_max_tree_depth=10, progress_bar=False)
    mcmc = process.infer(config=ex.get_config(), tf=100, obs_data=data)
    post = process.get_samples(group_by_chain=True)
    assert set(post) == {"strains_0_r0", "strains_0_infectious_period"}
    assert post["strains_0_r0"].shape == (48, 250)
    assert process.get_samples()["strains_0_r0"].shape == (48 * 250,)
    assert int(mcmc.nuts.diverging.sum()) <= 5 and 0.6 < float(mcmc.nuts.accept_prob.mean()) < 0.97
    (g_r0, cdf_r0), (g_ti, cdf_ti) = _grid_marginals(data)
    for name, grid, cdf in (("strains_0_r0", g_r0, cdf_r0), ("strains_0_infectious_period", g_ti, cdf_ti)):
        thin = post[name][:, ::10].reshape(-1).cpu().numpy()          # 48 x 25 nearly independent draws
        ks = stats.kstest(thin, lambda x: np.interp(x, grid, cdf))
        assert ks.pvalue > 1e-3, (name, ks)
        # the data were generated at r0 = 2, T_inf = 7
        truth = 2.0 if name.endswith("r0") else 7.0
        assert abs(np.median(thin) - truth) < 4 * np.std(thin) / np.sqrt(thin.size) + 0.02 * truth
        print(name, "posterior mean %.4f sd %.4f KS p=%.3f" % (thin.mean(), thin.std(), ks.pvalue))
    print("mean leapfrogs/transition %.2f, gradient-solves %d" % (float(mcmc.nuts.num_steps.double().mean()), mcmc.nuts.potential_evals))


def test_get_samples_before_infer_raises():
    with pytest.raises(AssertionError):
        MCMCProcess(numpyro_model=ex.model, num_warmup=1, num_samples=1, num_chains=1, nuts_max_tree_depth=1).get_samples()


def test_predictive_is_one_batched_solve(data):
    """numpyro.infer.Predictive counterpart (inference.py:225-237; sir_infer_parameters.py:159-168)."""
    from dynode_amd.infer import Predictive, checkpoint_compartment_sizes

    def model(config, tf, obs_data):
        sol = ex.model(config, tf, obs_data)
        checkpoint_compartment_sizes(config, sol)
        return sol

    prior = Predictive(model, num_samples=64, exclude_deterministic=False)(rng_key=1, config=ex.get_config(), tf=60, obs_data=None)
    assert prior["strains_0_r0"].shape == (64,) and prior["inf_incidence"].shape == (64, 60, 2)
    assert prior["final_timestep_r"].shape == (64, 2) and float(prior["strains_0_r0"].min()) >= 1.5
    post = {"strains_0_r0": torch.full((5,), 2.0), "strains_0_infectious_period": torch.full((5,), 7.0)}
    pp = Predictive(model, posterior_samples=post)(rng_key=2, config=ex.get_config(), tf=100, obs_data=None)
    assert set(pp) == {"inf_incidence"} and pp["inf_incidence"].shape == (5, 100, 2)
    # Poisson draws around the noiseless incidence the data were generated from
    assert abs(float(pp["inf_incidence"].mean()) - float(data.mean())) < 0.2 * float(data.mean())


def test_svi_gaussian_fit_lands_on_the_posterior(data):
    """reference inference.py:244-302 (SVIProcess: AutoMultivariateNormal + Adam(0.1) + ELBO)."""
    from dynode_amd.infer.inference import SVIProcess

    proc = SVIProcess(numpyro_model=ex.model, num_iterations=400, num_samples=2000, num_particles=16, progress_bar=False)
    res = proc.infer(config=ex.get_config(), tf=100, obs_data=data)
    assert float(res.losses[-50:].mean()) < float(res.losses[:20].mean())          # the ELBO improved
    post = proc.get_samples()
    assert set(post) == {"strains_0_r0", "strains_0_infectious_period"} and post["strains_0_r0"].shape == (2000,)
    # a Gaussian in the unconstrained space cannot match the ridge exactly; its centre must
    assert abs(float(post["strains_0_r0"].median()) - 2.04) < 0.12
    assert abs(float(post["strains_0_infectious_period"].median()) - 7.2) < 0.5
    with pytest.raises(AssertionError):
        SVIProcess(numpyro_model=ex.model, num_iterations=1, num_samples=1).get_samples()
