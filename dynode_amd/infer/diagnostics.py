"""Convergence diagnostics for the sampler output, the two columns numpyro's ``MCMC.print_summary`` adds to the moments:
effective sample size and split R-hat (Gelman et al., BDA3 section 11.4-11.5; Geyer's initial monotone sequence
for the autocorrelation sum, as in Stan / numpyro ``diagnostics.effective_sample_size``)."""

from __future__ import annotations

import numpy as np


def _split(x: np.ndarray) -> np.ndarray:
    """[chains, draws] -> [2 chains, draws // 2] (first and second halves as separate chains)."""
    n = x.shape[1] // 2
    return np.concatenate([x[:, :n], x[:, x.shape[1] - n:]], axis=0)


def split_rhat(x) -> float:
    """Potential scale reduction on split chains; 1 at convergence.  ``x``: [chains, draws]."""
    x = _split(np.asarray(x, dtype=np.float64))
    n = x.shape[1]
    if n < 2:
        return float("nan")
    within = x.var(axis=1, ddof=1).mean()
    between = n * x.mean(axis=1).var(ddof=1)
    if within == 0.0:
        return float("nan")
    return float(np.sqrt(((n - 1) / n * within + between / n) / within))


def _autocovariance(x: np.ndarray) -> np.ndarray:
    """Biased autocovariance of every chain by FFT: [chains, draws] -> [chains, draws]."""
    n = x.shape[1]
    size = 1 << int(np.ceil(np.log2(2 * n)))
    f = np.fft.rfft(x - x.mean(axis=1, keepdims=True), size, axis=1)
    return np.fft.irfft(f * np.conj(f), size, axis=1)[:, :n] / n


def effective_sample_size(x) -> float:
    """Effective number of independent draws in ``x`` [chains, draws] (all chains together)."""
    x = np.asarray(x, dtype=np.float64)
    m, n = x.shape
    if n < 4:
        return float("nan")
    gamma = _autocovariance(x)                                  # per chain, lag 0 .. n - 1
    within = gamma[:, 0].mean() * n / (n - 1.0)
    var_plus = gamma[:, 0].mean() + (x.mean(axis=1).var(ddof=1) if m > 1 else 0.0)
    if var_plus == 0.0:
        return float("nan")
    rho = 1.0 - (within - gamma.mean(axis=0) * n / (n - 1.0)) / var_plus
    rho[0] = 1.0
    # Geyer: sums of adjacent pairs are positive and decreasing; truncate at the first negative pair
    pairs = rho[: (n // 2) * 2].reshape(-1, 2).sum(axis=1)
    neg = np.nonzero(pairs < 0)[0]
    if neg.size:
        pairs = pairs[: neg[0]]
    pairs = np.minimum.accumulate(pairs) if pairs.size else pairs
    tau = -1.0 + 2.0 * pairs.sum()
    return float(m * n / max(tau, 1.0 / np.log10(max(m * n, 10))))
