"""A prior that must be supplied from outside (reference src/dynode/config/placeholder_sample.py).

``PlaceholderSample()`` marks a parameter whose values come from an external set of draws
(``handlers.substitute`` / ``Predictive(posterior_samples=...)``).  Drawing from it directly is a
mistake and raises `SamplePlaceholderError`.
"""

from __future__ import annotations

from ..infer.distributions import Distribution


class SamplePlaceholderError(Exception):
    """Raised when a `PlaceholderSample` is sampled instead of substituted."""


class PlaceholderSample(Distribution):
    def sample(self, rng=None, sample_shape=()):
        raise SamplePlaceholderError(
            "a PlaceholderSample parameter was sampled directly: provide its values through "
            "handlers.substitute(...) or Predictive(posterior_samples=...)")

    def log_prob(self, value):
        raise SamplePlaceholderError("a PlaceholderSample has no density; it only stands in for substituted values")
