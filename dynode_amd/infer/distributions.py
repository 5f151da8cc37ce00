"""The handful of distributions the reference's priors and likelihood use, under numpyro's names.

numpyro is not available to this build.  These are small torch implementations (float64,
autograd- and device-transparent: ``log_prob`` follows the device of its argument), used for
priors and likelihoods only -- never on the solve path.  Reference call sites:
examples/sir_infer_parameters.py:47-58 (``TransformedDistribution(Beta(0.5, 0.5),
AffineTransform(1.5, 1))``, ``TruncatedNormal(8, 2, low=2, high=15)``) and the ``Poisson``
likelihood (:34-38).
"""

from __future__ import annotations

import math
from types import SimpleNamespace

import torch

_F = torch.float64
_SQRT2 = math.sqrt(2.0)


_DEV_CACHE: dict = {}


def _t(x, like=None):
    """float64 tensor of ``x`` on the device of ``like``.  Device copies of (constant) distribution
    parameters are cached, so evaluating a log-density in a sampler loop issues no host-to-device
    copies (and stays HIP-graph capturable)."""
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=_F)
    dev = like.device if isinstance(like, torch.Tensor) else t.device
    if t.device == dev and t.dtype == _F:
        return t
    if t.requires_grad or t.numel() > (1 << 20):
        return t.to(device=dev, dtype=_F)
    if t.device.type == "cpu" and dev.type != "cpu" and t.numel() == 1:
        # a host scalar (``dist.Normal(mu, 1.0)`` inside a model function makes a new one in every evaluation, so the cache below
        # never hits): written by a fill kernel instead of a host-to-device copy -- which a HIP-graph capture refuses
        return torch.full(t.shape, float(t), dtype=_F, device=dev)
    # constant that needs a device copy and/or a float64 conversion: do it once
    key = (id(t), str(dev))
    hit = _DEV_CACHE.get(key)
    if hit is None or hit[0] is not t or hit[2] != t._version:
        if len(_DEV_CACHE) > 512:
            _DEV_CACHE.clear()
        hit = (t, t.to(device=dev, dtype=_F), t._version)
        _DEV_CACHE[key] = hit
    return hit[1]


def _ndtr(z):
    return 0.5 * (1.0 + torch.erf(z / _SQRT2))


def _ndtri(p):
    return _SQRT2 * torch.erfinv(2.0 * p - 1.0)


class Distribution:
    """Minimal numpyro-like distribution: sample(rng, shape), log_prob(value), support, icdf."""

    support = (-math.inf, math.inf)

    def sample(self, rng: torch.Generator, sample_shape=()):
        u = torch.rand(tuple(sample_shape) + tuple(self.batch_shape), generator=rng, dtype=_F)
        return self.icdf(u.clamp(1e-12, 1 - 1e-12))

    batch_shape = ()

    def log_prob(self, value):
        raise NotImplementedError

    def icdf(self, q):
        raise NotImplementedError

    @property
    def median(self):
        return self.icdf(torch.tensor(0.5, dtype=_F))


class Normal(Distribution):
    def __init__(self, loc=0.0, scale=1.0):
        self.loc, self.scale = _t(loc), _t(scale)
        self.batch_shape = torch.broadcast_shapes(self.loc.shape, self.scale.shape)

    def log_prob(self, value):
        v = _t(value)
        loc, scale = _t(self.loc, v), _t(self.scale, v)
        z = (v - loc) / scale
        return -0.5 * z * z - torch.log(scale) - 0.5 * math.log(2 * math.pi)

    def icdf(self, q):
        q = _t(q)
        return _t(self.loc, q) + _t(self.scale, q) * _ndtri(q)


class Uniform(Distribution):
    def __init__(self, low=0.0, high=1.0):
        self.low, self.high = _t(low), _t(high)
        self.support = (float(self.low.min()), float(self.high.max()))
        self.batch_shape = torch.broadcast_shapes(self.low.shape, self.high.shape)

    def log_prob(self, value):
        v = _t(value)
        lo, hi = _t(self.low, v), _t(self.high, v)
        inside = (v >= lo) & (v <= hi)
        return torch.where(inside, -torch.log(hi - lo), torch.full_like(v, -math.inf))

    def icdf(self, q):
        q = _t(q)
        return _t(self.low, q) + (_t(self.high, q) - _t(self.low, q)) * q


class Beta(Distribution):
    support = (0.0, 1.0)

    def __init__(self, concentration1, concentration0):
        self.a, self.b = _t(concentration1), _t(concentration0)
        self.batch_shape = torch.broadcast_shapes(self.a.shape, self.b.shape)

    def sample(self, rng, sample_shape=()):
        shape = tuple(sample_shape) + tuple(self.batch_shape)
        ga = torch._standard_gamma(self.a.expand(shape).contiguous(), generator=rng)
        gb = torch._standard_gamma(self.b.expand(shape).contiguous(), generator=rng)
        return (ga / (ga + gb)).clamp(1e-12, 1 - 1e-12)

    def log_prob(self, value):
        v = _t(value)
        a, b = _t(self.a, v), _t(self.b, v)
        lbeta = torch.lgamma(a) + torch.lgamma(b) - torch.lgamma(a + b)
        return (a - 1.0) * torch.log(v) + (b - 1.0) * torch.log1p(-v) - lbeta

    def icdf(self, q):
        from scipy.stats import beta as sbeta  # host-side, used for medians / initial values only
        q = _t(q)
        return torch.as_tensor(sbeta.ppf(q.cpu().numpy(), self.a.numpy(), self.b.numpy()), dtype=_F)


class TruncatedNormal(Distribution):
    """Normal(loc, scale) restricted to [low, high] (numpyro.distributions.TruncatedNormal)."""

    def __init__(self, loc=0.0, scale=1.0, low=None, high=None):
        self.loc, self.scale = _t(loc), _t(scale)
        self.low = -math.inf if low is None else float(low)
        self.high = math.inf if high is None else float(high)
        self.support = (self.low, self.high)
        self.batch_shape = torch.broadcast_shapes(self.loc.shape, self.scale.shape)
        self._a = _ndtr((torch.tensor(self.low, dtype=_F) - self.loc) / self.scale)
        self._b = _ndtr((torch.tensor(self.high, dtype=_F) - self.loc) / self.scale)
        self._logz = torch.log(self._b - self._a)

    def icdf(self, q):
        q = _t(q)
        return _t(self.loc, q) + _t(self.scale, q) * _ndtri(_t(self._a, q) + q * _t(self._b - self._a, q))

    def log_prob(self, value):
        v = _t(value)
        loc, scale = _t(self.loc, v), _t(self.scale, v)
        z = (v - loc) / scale
        lp = -0.5 * z * z - 0.5 * math.log(2 * math.pi) - torch.log(scale) - _t(self._logz, v)
        return torch.where((v >= self.low) & (v <= self.high), lp, torch.full_like(lp, -math.inf))


_LGAMMA_CACHE: dict = {}


def _lgamma1p(v: torch.Tensor) -> torch.Tensor:
    """lgamma(v + 1); cached for constant tensors (observed counts scored once per gradient)."""
    if v.requires_grad:
        return torch.lgamma(v + 1.0)
    key = (id(v), v.data_ptr(), v._version)
    hit = _LGAMMA_CACHE.get(key)
    if hit is None or hit[0] is not v:
        if len(_LGAMMA_CACHE) > 64:
            _LGAMMA_CACHE.clear()
        hit = (v, torch.lgamma(v + 1.0))
        _LGAMMA_CACHE[key] = hit
    return hit[1]


class Poisson(Distribution):
    support = (0.0, math.inf)

    def __init__(self, rate):
        self.rate = rate if isinstance(rate, torch.Tensor) else _t(rate)
        self.batch_shape = tuple(self.rate.shape)

    def sample(self, rng, sample_shape=()):
        rate = self.rate.detach().to("cpu", _F)
        return torch.poisson(rate.expand(tuple(sample_shape) + tuple(rate.shape)), generator=rng)

    def log_prob(self, value):
        rate = self.rate.to(_F)
        v = _t(value, rate)
        return v * torch.log(rate) - rate - _lgamma1p(v)


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
        self.batch_shape = base_distribution.batch_shape

    def sample(self, rng, sample_shape=()):
        x = self.base.sample(rng, sample_shape)
        for t in self.transforms:
            x = t(x)
        return x

    def log_prob(self, value):
        x, ladj = _t(value), 0.0
        for t in reversed(self.transforms):
            x = t.inv(x)
            ladj += t.log_abs_det_jacobian()
        return self.base.log_prob(x) - ladj

    def icdf(self, q):
        x = self.base.icdf(q)
        for t in self.transforms:
            x = t(x)
        return x


transforms = SimpleNamespace(AffineTransform=AffineTransform)


# ---------------------------------------------------------------------- unconstraining bijections
class Bijection:
    """Map an unconstrained real z to the support of a distribution (numpyro's ``biject_to``)."""

    def __init__(self, support):
        self.lo, self.hi = support
        self.kind = ("real" if math.isinf(self.lo) and math.isinf(self.hi) else
                     "interval" if not (math.isinf(self.lo) or math.isinf(self.hi)) else
                     "lower" if math.isinf(self.hi) else "upper")

    def __call__(self, z):
        if self.kind == "real":
            return z
        if self.kind == "interval":
            return self.lo + (self.hi - self.lo) * torch.sigmoid(z)
        if self.kind == "lower":
            return self.lo + torch.exp(z)
        return self.hi - torch.exp(z)

    def inv(self, x):
        x = _t(x)
        if self.kind == "real":
            return x
        if self.kind == "interval":
            u = ((x - self.lo) / (self.hi - self.lo)).clamp(1e-12, 1 - 1e-12)
            return torch.log(u) - torch.log1p(-u)
        if self.kind == "lower":
            return torch.log(x - self.lo)
        return torch.log(self.hi - x)

    def log_abs_det_jacobian(self, z):
        if self.kind == "real":
            return torch.zeros_like(z)
        if self.kind == "interval":
            return math.log(self.hi - self.lo) + torch.nn.functional.logsigmoid(z) + torch.nn.functional.logsigmoid(-z)
        return z


def biject_to(support) -> Bijection:
    return Bijection(support)
