"""dynode_amd/infer/checks.py on analytic targets (CPU): the tools the cfg 4 posterior gates of tests/test_gpu_infer.py
are built from must themselves be calibrated."""

import numpy as np
import torch
from scipy import stats

from dynode_amd.infer import checks
from dynode_amd.infer.nuts import BatchedNUTS

COV = np.array([[1.0, 0.6], [0.6, 0.5]])


def _gaussian_grid(n=161):
    """A correlated 2-D Gaussian in the unconstrained coordinates; site 0 constrained by exp (log-normal), site 1 by identity."""
    z0, z1 = np.linspace(-7.0, 7.0, n), np.linspace(-5.0, 5.0, n)
    prec = np.linalg.inv(COV)
    Z0, Z1 = np.meshgrid(z0, z1, indexing="ij")
    lj = -0.5 * (prec[0, 0] * Z0 ** 2 + 2 * prec[0, 1] * Z0 * Z1 + prec[1, 1] * Z1 ** 2)
    return checks.GridPosterior([z0, z1], np.exp(lj), [np.exp, lambda z: z], ("a", "b"))


def test_grid_posterior_moments_tails_and_core_against_closed_forms():
    g = _gaussian_grid()
    # b ~ N(0, 0.5); a = exp(z0), z0 ~ N(0, 1): log-normal moments
    assert abs(g.mean[1]) < 1e-9 and abs(g.sd[1] - np.sqrt(0.5)) < 1e-7
    assert abs(g.mean[0] - np.exp(0.5)) < 1e-5 and abs(g.sd[0] ** 2 - (np.e - 1) * np.e) < 1e-3
    assert abs(g.tail_mass(0, 1.0) - stats.norm.sf(1.0)) < 2e-5                     # a cell-centre threshold: half the cell counts
    assert abs(g.tail_mass(1, 0.33) - stats.norm.sf(0.33 / np.sqrt(0.5))) < 1e-4                  # inside a cell: second order in the spacing
    core = 0.5 * (stats.chi2.cdf(9.0, 3))                                           # E[x^2; |x| <= 3 sd] = var * P(chi2_3 <= 9)
    coarse, fine = g.core_second_moment(1, 3.0), g.refined(4).core_second_moment(1, 3.0)
    assert abs(fine - core) < 1e-5 and abs(coarse - core) < 2e-4 and abs(fine - core) < abs(coarse - core)
    # refinement keeps the full moments and sharpens what cuts the line
    r = g.refined(4)
    assert abs(r.sd[1] - g.sd[1]) < 1e-7 and abs(r.tail_mass(0, 1.01) - stats.norm.sf(1.01)) < 2e-5 < abs(g.tail_mass(0, 1.01) - stats.norm.sf(1.01))


def test_exact_draws_and_run_statistics_are_calibrated_on_iid_chains():
    g = _gaussian_grid().refined(2)
    rng = np.random.default_rng(0)
    z = g.draws(200_000, rng)
    assert stats.kstest(z[:, 1], "norm", args=(0.0, np.sqrt(0.5))).pvalue > 1e-3
    assert abs(np.corrcoef(z.T)[0, 1] - 0.6 / np.sqrt(0.5)) < 0.01
    runs = [checks.run_statistics(g, g.draws(64 * 200, rng).reshape(64, 200, 2), tails=((0, 1.0),)) for _ in range(6)]
    for r in runs:
        assert r["a"]["thin"] <= 3 and 0.0 < r["a"]["ks_p"] <= 1.0 and abs(r["tail_ratio"]["a>z1"] - 1.0) < 0.1
    pooled = checks.pool_runs(g, runs)
    for n in ("a", "b"):
        assert abs(pooled[n]["mean_z"]) < 4 and abs(pooled[n]["var_z"]) < 4 and abs(pooled[n]["core_z"]) < 4
        assert abs(pooled[n]["core_sd_ratio"] - 1.0) < 4 * pooled[n]["core_sd_ratio_se"] + 1e-3 and pooled[n]["ks_fisher_p"] > 1e-3
        assert "_chain_vars" not in runs[0][n]                                     # stripped: what is left is JSON
    ctl = checks.iid_control(g, 60, 2000, rng)
    assert ctl["b"]["ks_p_uniformity_p"] > 1e-3 and abs(ctl["b"]["sd_ratio_mean"] - 1.0) < 0.01


def test_excursions_counts_sojourns():
    z = np.array([[0, 3, 3, 0, 3], [3, 0, 0, 0, 0], [0, 0, 0, 0, 0]], float)
    e = checks.excursions(z, 2.0)
    assert e["count"] == 3 and e["max_length"] == 2 and e["draws_beyond"] == 4
    assert abs(e["share_of_draws_in_sojourns_cut_by_the_run"] - 0.5) < 1e-12       # the sojourn at the start and the one at the end
    assert checks.excursions(z, 5.0)["count"] == 0


def test_stationarity_check_passes_an_exact_kernel_and_catches_a_wrong_one():
    """The CPU torch sampler on the analytic Gaussian: from exact starts with a fixed kernel the states stay posterior draws;
    the same check against a posterior 20 % wider fails."""
    g = _gaussian_grid().refined(2)
    prec = torch.tensor(np.linalg.inv(COV))

    def pg(z):
        gr = z @ prec
        return 0.5 * (z * gr).sum(-1), gr

    rng = np.random.default_rng(3)
    eps, imm = torch.tensor([0.35, 0.5]), torch.tensor(np.stack([COV, np.eye(2)]))
    rep = checks.stationarity(g, BatchedNUTS(pg, max_tree_depth=6, seed=5), eps, imm, 4000, 12, rng, tails=((0, 1.0),), at=(1, 4, 12), device="cpu")
    assert rep["divergences"] == 0 and set(rep["after"]) == {"1", "4", "12"}
    for row in rep["after"].values():
        for n in ("a", "b"):
            assert row[n]["ks_p"] > 1e-3 and abs(row[n]["mean_z"]) < 4 and abs(row[n]["var_z"]) < 4
        assert abs(row["a>z1"]["z"]) < 4
    z0g, z1g = g.z
    wide = checks.GridPosterior([z0g, z1g], g.p ** (1 / 1.44), g._constrain, g.names)        # the same shape, sd x 1.2
    bad = checks.stationarity(wide, BatchedNUTS(pg, max_tree_depth=6, seed=5), eps, imm, 4000, 20, rng, at=(20,), device="cpu")
    assert bad["last"]["b"]["ks_p"] < 1e-4 and bad["last"]["b"]["var_z"] < -6


def test_two_sample_ks_at_effective_sizes_is_calibrated_on_correlated_chains():
    """AR(1) chains with a N(0, 1) marginal against other such chains (the same distribution): counted as independent their
    draws reject far too often; read at the effective sizes the rejection rate is the nominal one.  Independent draws: the
    effective sizes are the sample sizes and the p-values those of the plain test."""
    rng = np.random.default_rng(11)

    def ar1(chains, n, rho):
        x = np.empty((chains, n))
        x[:, 0] = rng.standard_normal(chains)
        e = rng.standard_normal((chains, n)) * np.sqrt(1.0 - rho * rho)
        for t in range(1, n):
            x[:, t] = rho * x[:, t - 1] + e[:, t]
        return x

    naive, eff, n_effs = [], [], []
    for _ in range(150):
        a, b = ar1(32, 200, 0.9), ar1(24, 300, 0.8)
        naive.append(stats.ks_2samp(a.ravel(), b.ravel()).pvalue)
        d, p, na, nb = checks.ks_two_sample_effective(list(a), list(b))
        eff.append(p)
        n_effs.append((na, nb))
    naive, eff, n_effs = np.array(naive), np.array(eff), np.array(n_effs)
    assert (naive < 0.01).mean() > 0.25                                   # (the plain test: a quarter of correct pairs and more fail at 1 %)
    assert (eff < 0.01).mean() <= 0.04 and (eff < 0.1).mean() <= 0.2      # (nominal 1 % / 10 %; 150 repetitions)
    # effective sizes: N (1 - rho) / (1 + rho) = 337 / 800, estimated from 32 / 24 replicates
    assert 250 < np.median(n_effs[:, 0]) < 450 and 600 < np.median(n_effs[:, 1]) < 1050
    a, b = rng.standard_normal((40, 50)), rng.standard_normal((40, 50))
    d, p, na, nb = checks.ks_two_sample_effective(list(a), list(b))
    assert na > 1000 and nb > 1000 and abs(p - stats.ks_2samp(a.ravel(), b.ravel(), method="asymp").pvalue) < 0.1
    d, p, _, _ = checks.ks_two_sample_effective(list(a), list(b + 0.25))   # a quarter of a standard deviation apart: rejected
    assert p < 1e-3
