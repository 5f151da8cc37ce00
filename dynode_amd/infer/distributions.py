"""The handful of distributions the reference's priors and likelihood use, under numpyro's names.

numpyro is not available to this build; these are thin float64 wrappers over
``torch.distributions`` (host side, priors and likelihoods only -- never on the solve path).
Used by: examples/sir_infer_parameters.py:47-58 of the reference
(``TransformedDistribution(Beta(0.5, 0.5), AffineTransform(1.5, 1))``,
``TruncatedNormal(8, 2, low=2, high=15)``) and its ``Poisson`` likelihood (:34-38).
"""

from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.distributions as td

_F = torch.float64


def _t(x):
    return torch.as_tensor(x, dtype=_F)


class Distribution:
    """Minimal numpyro-like distribution: sample(rng, shape), log_prob(value), support."""

    support = (-math.inf, math.inf)

    def sample(self, rng: torch.Generator, sample_shape=()):
        raise NotImplementedError

    def log_prob(self, value):
        raise NotImplementedError

    @property
    def median(self):
        return self.icdf(_t(0.5))

    def icdf(self, q):
        raise NotImplementedError


class _Wrapped(Distribution):
    def __init__(self, base: td.Distribution, support):
        self._base = base
        self.support = support

    def sample(self, rng, sample_shape=()):
        # torch.distributions has no generator argument: draw uniforms with ours, then invert the cdf
        u = torch.rand(tuple(sample_shape) + tuple(self._base.batch_shape), generator=rng, dtype=_F)
        u = u.clamp(1e-12, 1 - 1e-12)
        return self.icdf(u)

    def log_prob(self, value):
        return self._base.log_prob(_t(value))

    def icdf(self, q):
        return self._base.icdf(_t(q))


class Normal(_Wrapped):
    def __init__(self, loc=0.0, scale=1.0):
        super().__init__(td.Normal(_t(loc), _t(scale)), (-math.inf, math.inf))


class Uniform(_Wrapped):
    def __init__(self, low=0.0, high=1.0):
        super().__init__(td.Uniform(_t(low), _t(high)), (float(low), float(high)))


class Beta(Distribution):
    support = (0.0, 1.0)

    def __init__(self, concentration1, concentration0):
        self._base = td.Beta(_t(concentration1), _t(concentration0))

    def sample(self, rng, sample_shape=()):
        a, b = self._base.concentration1, self._base.concentration0
        # ratio of gammas drawn with our generator (torch's Beta.sample takes no generator)
        shape = tuple(sample_shape) + tuple(self._base.batch_shape)
        ga = torch._standard_gamma(a.expand(shape).contiguous(), generator=rng) if hasattr(torch, "_standard_gamma") else None
        gb = torch._standard_gamma(b.expand(shape).contiguous(), generator=rng)
        return (ga / (ga + gb)).clamp(1e-12, 1 - 1e-12)

    def log_prob(self, value):
        return self._base.log_prob(_t(value))

    def icdf(self, q):
        # bisection on the regularised incomplete beta (median of Beta(.5,.5) = .5 etc.)
        from scipy.stats import beta as sbeta
        return _t(sbeta.ppf(_t(q).numpy(), self._base.concentration1.numpy(), self._base.concentration0.numpy()))


class TruncatedNormal(Distribution):
    """Normal(loc, scale) restricted to [low, high] (numpyro.distributions.TruncatedNormal)."""

    def __init__(self, loc=0.0, scale=1.0, low=None, high=None):
        self.loc, self.scale = _t(loc), _t(scale)
        self.low = -math.inf if low is None else float(low)
        self.high = math.inf if high is None else float(high)
        self.support = (self.low, self.high)
        self._n = td.Normal(_t(0.0), _t(1.0))
        self._a = self._n.cdf((_t(self.low) - self.loc) / self.scale)
        self._b = self._n.cdf((_t(self.high) - self.loc) / self.scale)
        self._logz = torch.log(self._b - self._a)

    def icdf(self, q):
        return self.loc + self.scale * self._n.icdf(self._a + _t(q) * (self._b - self._a))

    def sample(self, rng, sample_shape=()):
        u = torch.rand(tuple(sample_shape) + tuple(self.loc.shape), generator=rng, dtype=_F).clamp(1e-12, 1 - 1e-12)
        return self.icdf(u)

    def log_prob(self, value):
        v = _t(value)
        z = (v - self.loc) / self.scale
        lp = -0.5 * z * z - 0.5 * math.log(2 * math.pi) - torch.log(self.scale) - self._logz
        return torch.where((v >= self.low) & (v <= self.high), lp, _t(-math.inf))


class Poisson(Distribution):
    support = (0.0, math.inf)

    def __init__(self, rate):
        self.rate = _t(rate)

    def sample(self, rng, sample_shape=()):
        return torch.poisson(self.rate.expand(tuple(sample_shape) + tuple(self.rate.shape)), generator=rng)

    def log_prob(self, value):
        v = _t(value)
        return v * torch.log(self.rate) - self.rate - torch.lgamma(v + 1.0)


class AffineTransform:
    def __init__(self, loc, scale):
        self.loc, self.scale = float(loc), float(scale)

    def __call__(self, x):
        return self.loc + self.scale * x

    def inv(self, y):
        return (y - self.loc) / self.scale

    def log_abs_det_jacobian(self):
        return math.log(abs(self.scale))


class TransformedDistribution(Distribution):
    """y = T_n(...T_1(x)), x ~ base; only affine transforms are needed by the reference's priors."""

    def __init__(self, base_distribution: Distribution, transforms):
        self.base = base_distribution
        self.transforms = list(transforms) if isinstance(transforms, (list, tuple)) else [transforms]
        lo, hi = base_distribution.support
        for t in self.transforms:
            lo, hi = sorted((t(lo), t(hi)))
        self.support = (lo, hi)

    def _inv(self, y):
        x, ladj = _t(y), 0.0
        for t in reversed(self.transforms):
            x = t.inv(x)
            ladj += t.log_abs_det_jacobian()
        return x, ladj

    def sample(self, rng, sample_shape=()):
        x = self.base.sample(rng, sample_shape)
        for t in self.transforms:
            x = t(x)
        return x

    def log_prob(self, value):
        x, ladj = self._inv(value)
        return self.base.log_prob(x) - ladj

    def icdf(self, q):
        x = self.base.icdf(q)
        for t in self.transforms:
            x = t(x)
        return x


transforms = SimpleNamespace(AffineTransform=AffineTransform)
