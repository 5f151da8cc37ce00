"""Calibrated posterior checks for low-dimensional models: draws against tensor-grid quadrature.

The north star asks for "KS-test agreement on posteriors" (BASELINE.json) of the NUTS path that replaces numpyro's
(reference src/dynode/infer/inference.py:149-163; model: examples/sir_infer_parameters.py:21-58).  A KS test of thinned
MCMC draws is only as good as the thinning, and a moment check only as good as its effective-sample-size error bar; on
this posterior -- a bent ridge 0.04 wide with an exponential tail along z0 that holds 0.74 % of the mass and 9 % of r0's
variance -- both are anti-conservative.  What this module provides instead:

`GridPosterior`       the posterior on a tensor grid of the unconstrained coordinates: marginal CDFs and moments of the
                      constrained sites, and EXACT independent draws of the joint (cell by its mass, uniform inside).
`run_statistics`      one run's draws against it: KS with the thinning taken from the effective sample size of the squares,
                      and ACROSS-CHAIN z statistics -- chains are independent, so the mean over chains of a per-chain
                      statistic has an honest standard error whatever the autocorrelation inside a chain, PROVIDED rare
                      chains do not dominate it: hence the CORE second moment (within 3 sd of the mean) next to the variance.
`pool_runs`           several runs (sampler seeds) pooled: sd ratio with its across-chain standard error, Fisher's
                      combination and a uniformity test of the runs' KS p-values.
`ks_two_sample_effective`  two samplers' draws of one site against each other (no quadrature: any dimension): the KS statistic
                      of all draws, its p-value at the samples' effective sizes, taken from replicate groups.
`stationarity`        the exactly calibrated test of a transition kernel: C chains start at independent exact posterior
                      draws, nothing adapts, T transitions; if the kernel leaves the posterior invariant the C states
                      after any number of transitions are i.i.d. posterior draws, so KS p-values are uniform without any
                      effective-sample-size estimate and tail counts are binomial.

Findings on cfg 4 with these tools (tools/posterior_study.py, docs/perf-log.md round 4; fused-likelihood model, numpyro's
per-chain adaptation, refined 2801 x 1201 truth):
  * the transition kernel is invariant at the resolution of a MILLION chains: after 1 ... 100 transitions from exact starts
    every KS p is in 0.18-0.99 and every z statistic (mean, variance, counts beyond z0 > 1, 2, 3, 4) is inside +-2;
  * 65,536 chains x 1000 draws from exact starts with the production runs' adapted kernels: sd ratio 0.9996 +- 0.0008, core sd
    ratio 1.0001 +- 0.0002, tail occupancies 0.98-0.99;
  * 64 production runs (128 chains x (1000 + 1000), init_to_median) pooled: core sd ratio 0.9998 +- 0.0004, but the plain sd
    ratio 0.9956 +- 0.0015 and the far tail under-visited (z0 > 2 / 3 / 4 occupied 0.93 / 0.81 / 0.69 x): in the stationary
    process a third of the draws beyond z0 > 4 belong to chains that sit there for their whole run (a body-adapted kernel barely
    moves on the far ridge), and a state that is that hard to leave is as hard to reach from the bulk in 2000 transitions.
    A per-run sd ratio has sd 1.1 % (min 0.976, max 1.046, median 0.995 over the 64 runs), its core 0.34 %.
  The same holds for any sampler with this kernel family -- numpyro's included -- on this target; what a finite run is held
  to is therefore the core moments, the KS tests and the exact stationarity check, with the plain sd ratio reported beside them.
"""

from __future__ import annotations

from typing import Callable, Optional, Sequence

import numpy as np


class GridPosterior:
    """Posterior mass on a tensor grid of the unconstrained coordinates.

    ``z_grids``: one uniform 1-D grid per site (unconstrained); ``pmf``: joint cell masses (sums to 1);
    ``constrain``: list of callables z_i -> x_i (numpy in, numpy out), one per site."""

    def __init__(self, z_grids: Sequence[np.ndarray], pmf: np.ndarray, constrain: Sequence[Callable], names: Sequence[str]):
        self.z = [np.asarray(g, np.float64) for g in z_grids]
        self.p = np.asarray(pmf, np.float64)
        self.p = self.p / self.p.sum()
        self.h = [g[1] - g[0] for g in self.z]
        self.names = list(names)
        self._constrain = list(constrain)
        D = self.p.ndim
        self.x = [self._constrain[k](self.z[k]) for k in range(D)]
        self.pmf = [self.p.sum(tuple(d for d in range(D) if d != k)) for k in range(D)]
        self.cdf = [np.cumsum(m) - 0.5 * m for m in self.pmf]                       # midpoint rule
        self.mean = [float((x * m).sum()) for x, m in zip(self.x, self.pmf)]
        self.sd = [float(np.sqrt((((x - mu) ** 2) * m).sum())) for x, m, mu in zip(self.x, self.pmf, self.mean)]

    def core_second_moment(self, k: int, width: float) -> float:
        """E[(x - mean)^2 ; |x - mean| <= width sd] of site k: the part of the variance that sits within ``width`` standard
        deviations of the mean (`run_statistics` compares a run's with it: a moment whose per-chain values are bounded)."""
        d = self.x[k] - self.mean[k]
        # a cell that straddles a cut-off counts by the share of its (constrained) width inside
        lo, hi = self._constrain[k](self.z[k] - 0.5 * self.h[k]), self._constrain[k](self.z[k] + 0.5 * self.h[k])
        a, b = self.mean[k] - width * self.sd[k], self.mean[k] + width * self.sd[k]
        inside = np.clip((np.minimum(hi, b) - np.maximum(lo, a)) / (hi - lo), 0.0, 1.0)
        return float(((d ** 2) * inside * self.pmf[k]).sum())

    @classmethod
    def from_potential(cls, potential, z_grids, chunk: int = 200_000) -> "GridPosterior":
        """Quadrature with the package's own log joint (float64 solves when x64 is enabled: callers switch it on)."""
        import torch

        grids = [torch.as_tensor(g, dtype=torch.float64) for g in z_grids]
        mesh = torch.meshgrid(*grids, indexing="ij")
        z = torch.stack([m.reshape(-1) for m in mesh], dim=1).to(potential.device)
        with torch.no_grad():
            lj = torch.cat([potential.log_joint(z[i:i + chunk])[0] for i in range(0, z.shape[0], chunk)])
        p = torch.exp(lj - lj.max()).reshape(mesh[0].shape).cpu().numpy()
        bij = list(potential.bij.values())
        con = [(lambda v, b=b: b(torch.as_tensor(np.ascontiguousarray(v), dtype=torch.float64)).numpy()) for b in bij]
        return cls([g.numpy() for g in grids], p, con, list(potential.bij))

    def refined(self, factor: int = 4) -> "GridPosterior":
        """The same posterior on a grid ``factor`` times finer in every coordinate (two sites), by bicubic interpolation of the
        log masses.  The node values of a tensor-grid quadrature are spectrally accurate for full-line integrals, but
        everything that CUTS the line is second order in the spacing -- tail masses, the core moment, the linear
        interpolation of the CDF -- and `draws` spreads a cell's mass uniformly over the cell, which widens a ridge 0.04
        across by 3 % on a 0.024 grid.  On the refined grid those effects are ``factor``^2 smaller (measured on cfg 4: the
        core second moment of the infectious period read 0.17 % / 0.37 % off on the 1401 x 601 / 701 x 501 grids themselves,
        six to eleven standard errors of a 65,536-chain check)."""
        from scipy.interpolate import RectBivariateSpline

        if self.p.ndim != 2:
            raise NotImplementedError("refined(): two sites")
        logp = np.log(np.maximum(self.p, 1e-300))
        logp = np.maximum(logp, logp.max() - 250.0)          # (far corners: keep the spline from ringing around -inf)
        fine = [np.linspace(g[0], g[-1], (len(g) - 1) * factor + 1) for g in self.z]
        lf = RectBivariateSpline(self.z[0], self.z[1], logp, kx=3, ky=3)(fine[0], fine[1])
        return GridPosterior(fine, np.exp(lf - lf.max()), self._constrain, self.names)

    def cdf_of(self, k: int) -> Callable:
        return lambda q: np.interp(q, self.x[k], self.cdf[k])

    def tail_mass(self, k: int, z_threshold: float) -> float:
        """Marginal mass of site k beyond ``z_threshold`` (unconstrained coordinate); a cell that straddles the threshold
        counts by the share of its width beyond it, as `draws` spreads a cell's mass uniformly over its width."""
        share = np.clip((self.z[k] + 0.5 * self.h[k] - z_threshold) / self.h[k], 0.0, 1.0)
        return float((self.pmf[k] * share).sum())

    def draws(self, n: int, rng: np.random.Generator) -> np.ndarray:
        """``n`` independent draws of the joint in the unconstrained coordinates, [n, D]."""
        idx = rng.choice(self.p.size, size=n, p=self.p.ravel())
        sub = np.unravel_index(idx, self.p.shape)
        return np.stack([self.z[k][sub[k]] + (rng.random(n) - 0.5) * self.h[k] for k in range(self.p.ndim)], axis=1)

    def constrain(self, z: np.ndarray) -> list:
        return [self._constrain[k](np.ascontiguousarray(z[..., k])) for k in range(self.p.ndim)]


def run_statistics(post: GridPosterior, z: np.ndarray, thin: Optional[int] = None, tails: Sequence[tuple] = (), core: float = 3.0) -> dict:
    """Statistics of one run's draws ``z`` [chains, draws, D] (unconstrained).  ``tails``: (site index, z threshold) pairs whose
    occupancy is reported relative to the quadrature mass.  Per site: mean, sd ratio, ESS of the values and of the squared
    deviations, the thinning used (``thin`` or ceil(2 x chains x draws / min ESS): every kept draw stands for at most half an
    effective one), KS p of the thinned draws, and the across-chain z statistics of the mean, the variance and the CORE
    variance (the second moment within ``core`` standard deviations of the quadrature mean)."""
    from scipy import stats

    from .diagnostics import effective_sample_size

    z = np.asarray(z, np.float64)
    C, N, _ = z.shape
    x = post.constrain(z)
    out = {"chains": C, "draws": N,
           "tail_ratio": {f"{post.names[k]}>z{t:g}": float((z[..., k] > t).mean() / post.tail_mass(k, t)) for k, t in tails}}
    for k, name in enumerate(post.names):
        xs = x[k]
        d2 = (xs - post.mean[k]) ** 2
        ess, ess2 = effective_sample_size(xs), effective_sample_size(d2)
        th = int(thin) if thin else int(max(1, np.ceil(2.0 * C * N / max(min(ess, ess2), 1.0))))
        th = min(th, N)
        thinned = xs[:, th - 1::th].reshape(-1)
        ks = stats.kstest(thinned, post.cdf_of(k))
        # the variance within `core` standard deviations of the mean: per-chain values are bounded, so the across-chain
        # standard error is honest where the plain variance's is not (on cfg 4 a third of the draws beyond z0 > 4 belong to
        # chains that sit there for their whole run -- 0.1 % of the chains carrying 17 x the typical squared deviation)
        core_q = post.core_second_moment(k, core)
        w_c = (d2 * (d2 <= (core * post.sd[k]) ** 2)).mean(1)
        m_c, v_c = xs.mean(1), d2.mean(1)
        se = lambda a: a.std(ddof=1) / np.sqrt(C) if C > 1 else float("nan")   # noqa: E731
        out[name] = {"mean": float(xs.mean()), "sd": float(xs.std()), "sd_ratio": float(np.sqrt(d2.mean()) / post.sd[k]),
                     "core_sd_ratio": float(np.sqrt(w_c.mean() / core_q)),
                     "ess": float(ess), "ess_sq": float(ess2), "thin": th, "n_thinned": int(thinned.size), "ks_p": float(ks.pvalue),
                     "chain_mean_z": float((m_c.mean() - post.mean[k]) / se(m_c)), "chain_var_z": float((v_c.mean() - post.sd[k] ** 2) / se(v_c)),
                     "chain_core_z": float((w_c.mean() - core_q) / se(w_c)),
                     "_chain_means": m_c, "_chain_vars": v_c, "_chain_core": w_c}
    return out


def pool_runs(post: GridPosterior, runs: Sequence[dict], strip_runs: bool = True) -> dict:
    """Pool the `run_statistics` of several independent runs: per site the sd ratio over all chains with its across-chain
    standard error, the z statistics of mean and variance, the runs' KS p-values with Fisher's combination and a KS test of
    their uniformity; the mean tail occupancies.  Strips the per-chain arrays from ``runs`` (unless told not to)."""
    from scipy import stats

    out = {"runs": len(runs), "chains": int(sum(r["chains"] for r in runs)),
           "tail_ratio_mean": {t: float(np.mean([r["tail_ratio"][t] for r in runs])) for t in (runs[0]["tail_ratio"] if runs else {})}}
    for k, name in enumerate(post.names):
        m_c = np.concatenate([r[name]["_chain_means"] for r in runs])
        v_c = np.concatenate([r[name]["_chain_vars"] for r in runs])
        w_c = np.concatenate([r[name]["_chain_core"] for r in runs])
        ks = np.array([r[name]["ks_p"] for r in runs])
        n = m_c.size
        core_q = float(np.mean([r[name]["_chain_core"].mean() / r[name]["core_sd_ratio"] ** 2 for r in runs]))   # (the quadrature value, recovered)
        out[name] = {"sd_ratio": float(np.sqrt(v_c.mean()) / post.sd[k]),
                     "core_sd_ratio": float(np.sqrt(w_c.mean() / core_q)),
                     "core_sd_ratio_se": float(w_c.std(ddof=1) / np.sqrt(n) / (2.0 * core_q)),
                     "core_z": float((w_c.mean() - core_q) / (w_c.std(ddof=1) / np.sqrt(n))),
                     "sd_ratio_se": float(v_c.std(ddof=1) / np.sqrt(n) / (2.0 * post.sd[k] ** 2)),
                     "mean_z": float((m_c.mean() - post.mean[k]) / (m_c.std(ddof=1) / np.sqrt(n))),
                     "var_z": float((v_c.mean() - post.sd[k] ** 2) / (v_c.std(ddof=1) / np.sqrt(n))),
                     "ks_p": [float(v) for v in ks], "ks_p_min": float(ks.min()),
                     "ks_fisher_p": float(stats.combine_pvalues(ks, method="fisher")[1]) if ks.size > 1 else float(ks[0]),
                     "ks_uniformity_p": float(stats.kstest(ks, "uniform").pvalue) if ks.size > 1 else float("nan")}
    if strip_runs:
        for r in runs:
            strip(r)
    return out


def excursions(z_site: np.ndarray, z_threshold: float) -> dict:
    """Sojourns of the chains beyond ``z_threshold``: runs of consecutive draws above it in ``z_site`` [chains, draws].
    How many there were, their mean / longest length in draws, and the share of the tail's draws that sits in sojourns
    cut off by the end of the run -- the numbers that say how many independent visits a tail statistic rests on."""
    above = np.asarray(z_site) > z_threshold
    C, N = above.shape
    pad = np.zeros((C, 1), dtype=bool)
    edges = np.diff(np.concatenate([pad, above, pad], axis=1).astype(np.int8), axis=1)
    starts, ends = np.nonzero(edges == 1), np.nonzero(edges == -1)
    lengths = ends[1] - starts[1]
    cut = (ends[1] == N) | (starts[1] == 0)
    return {"count": int(lengths.size), "mean_length": float(lengths.mean()) if lengths.size else 0.0,
            "max_length": int(lengths.max()) if lengths.size else 0, "draws_beyond": int(above.sum()),
            "share_of_draws_in_sojourns_cut_by_the_run": float(lengths[cut].sum() / max(int(above.sum()), 1))}


def strip(run: dict) -> dict:
    """Drop the per-chain arrays of a `run_statistics` result (what is left is JSON)."""
    for v in run.values():
        if isinstance(v, dict):
            v.pop("_chain_means", None)
            v.pop("_chain_vars", None)
            v.pop("_chain_core", None)
    return run


def iid_control(post: GridPosterior, reps: int, n: int, rng: np.random.Generator) -> dict:
    """What the per-run statistics look like when the draws ARE the posterior: ``reps`` sets of ``n`` independent draws.
    Per site: share of KS p-values below 0.01, a uniformity test of them, mean / sd / 1 %-99 % range of the sd ratio."""
    from scipy import stats

    ks = [[] for _ in post.names]
    sr = [[] for _ in post.names]
    for _ in range(reps):
        x = post.constrain(post.draws(n, rng))
        for k in range(len(post.names)):
            ks[k].append(float(stats.kstest(x[k], post.cdf_of(k)).pvalue))
            sr[k].append(float(np.sqrt(((x[k] - post.mean[k]) ** 2).mean()) / post.sd[k]))
    out = {"reps": reps, "draws_per_rep": n}
    for k, name in enumerate(post.names):
        p, s = np.array(ks[k]), np.array(sr[k])
        out[name] = {"ks_p_below_0.01": float((p < 0.01).mean()), "ks_p_uniformity_p": float(stats.kstest(p, "uniform").pvalue),
                     "sd_ratio_mean": float(s.mean()), "sd_ratio_sd": float(s.std()), "sd_ratio_q01_q99": [float(np.quantile(s, 0.01)), float(np.quantile(s, 0.99))]}
    return out


def stationarity(post: GridPosterior, sampler, step_size, inverse_mass, chains: int, transitions: int, rng: np.random.Generator,
                 tails: Sequence[tuple] = (), at: Optional[Sequence[int]] = None, device="cuda") -> dict:
    """Start ``chains`` chains at independent exact posterior draws, give chain c the kernel (step_size[pick[c]],
    inverse_mass[pick[c]]) with ``pick`` uniform over the rows supplied (the adapted kernels of production runs), run
    ``transitions`` transitions with nothing adapting (``sampler.run(z0, 0, T, step_size=..., inverse_mass=...)``), and
    compare the states after the transition counts in ``at`` (default: a few, and the last) with the posterior: per site KS p,
    z statistics of mean and variance (i.i.d. across chains: exact standard errors from the quadrature moments), and for
    every (site index, z threshold) in ``tails`` the count beyond the threshold against its binomial law."""
    import torch
    from scipy import stats

    step_size = torch.as_tensor(step_size, dtype=torch.float64)
    inverse_mass = torch.as_tensor(inverse_mass, dtype=torch.float64)
    z_start = post.draws(chains, rng)
    pick = torch.from_numpy(rng.integers(0, step_size.shape[0], chains))
    res = sampler.run(torch.from_numpy(z_start).to(device), 0, transitions, step_size=step_size[pick].to(device),
                      inverse_mass=inverse_mass[pick].to(device))
    z = res.samples.cpu().numpy()
    at = sorted({t for t in (at or (1, 2, 5, 10, 20, 50, transitions)) if 1 <= t <= transitions})
    out = {"chains": chains, "transitions": transitions, "gradient_solves": int(res.potential_evals), "divergences": int(res.diverging.sum()),
           "after": {}}

    def compare(zt):
        x = post.constrain(zt)
        row = {}
        for k, name in enumerate(post.names):
            d2 = (x[k] - post.mean[k]) ** 2
            row[name] = {"ks_p": float(stats.kstest(x[k], post.cdf_of(k)).pvalue),
                         "mean_z": float((x[k].mean() - post.mean[k]) / (post.sd[k] / np.sqrt(chains))),
                         "var_z": float((d2.mean() - post.sd[k] ** 2) / (d2.std(ddof=1) / np.sqrt(chains))),
                         "sd_ratio": float(np.sqrt(d2.mean()) / post.sd[k])}
        for k, t in tails:
            n_t, p_t = int((zt[:, k] > t).sum()), post.tail_mass(k, t)
            row[f"{post.names[k]}>z{t:g}"] = {"count": n_t, "expected": chains * p_t,
                                               "z": float((n_t - chains * p_t) / np.sqrt(chains * p_t * (1.0 - p_t))),
                                               "binom_p": float(stats.binomtest(n_t, chains, p_t).pvalue)}
        return row

    out["start"] = compare(z_start)
    for t in at:
        out["after"][str(t)] = compare(z[:, t - 1])
    out["last"] = out["after"][str(at[-1])]
    return out


def ks_two_sample_effective(a_groups: Sequence[np.ndarray], b_groups: Sequence[np.ndarray]) -> tuple:
    """Two-sample Kolmogorov-Smirnov on correlated draws: the statistic from all draws, its p-value at the samples' EFFECTIVE
    sizes.  `*_groups`: arrays whose means are replicate estimates of the same mean -- a sampler's independent chains; the
    time blocks of an ensemble of coupled walkers --: n_eff = groups x pooled variance / variance of the group means, at most
    the number of draws.  Thinned draws counted as independent make the p-value anti-conservative by the factor they are
    not (the stretch-move ensemble in nine dimensions: about 5), and then a correct sampler fails by realization.  Returns
    (statistic, p, n_eff_a, n_eff_b); the asymptotic Kolmogorov distribution, so a few hundred effective draws at least."""
    from scipy import stats
    from scipy.special import kolmogorov

    def n_eff(groups):
        pooled = np.concatenate([np.asarray(g, dtype=np.float64).ravel() for g in groups])
        means = np.array([np.mean(g) for g in groups])
        return float(min(pooled.size, len(groups) * pooled.var() / max(means.var(ddof=1), 1e-300))), pooled

    na, a = n_eff(a_groups)
    nb, b = n_eff(b_groups)
    d = float(stats.ks_2samp(a, b).statistic)
    return d, float(kolmogorov(np.sqrt(na * nb / (na + nb)) * d)), na, nb
