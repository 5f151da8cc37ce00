"""A minimal effect-handler stack: the part of numpyro's primitives the DynODE path relies on.

``sample(name, dist, obs=None)`` and ``deterministic(name, value)`` record sites; the handlers
``seed`` (supplies randomness, optionally a whole batch of draws per site), ``substitute``
(replays given values, e.g. posterior samples) and ``trace`` (records sites) compose like
numpyro's.  Reference call sites: src/dynode/infer/sample.py:76,155;
examples/sir_infer_parameters.py:34-38.
"""

from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import torch

_STACK: list = []


class _Handler:
    def __enter__(self):
        _STACK.append(self)
        return self

    def __exit__(self, *exc):
        assert _STACK.pop() is self
        return False

    def process(self, msg):  # pragma: no cover - overridden
        pass

    def postprocess(self, msg):
        pass


class seed(_Handler):
    """Provide a random stream; ``batch`` > 0 draws that many values per sample site."""

    def __init__(self, rng_seed: int = 0, batch: int = 0):
        self.gen = torch.Generator().manual_seed(int(rng_seed))
        self.batch = int(batch)

    def process(self, msg):
        if msg["type"] == "sample" and msg["value"] is None and msg["gen"] is None:
            msg["gen"] = self.gen
            # one draw per batch member: prior sites are unbatched and get a sample_shape; sites whose
            # distribution already depends on batched values (e.g. the likelihood) are batched already
            shape = tuple(getattr(msg["fn"], "batch_shape", ()))
            if self.batch and not msg["sample_shape"] and not (shape and shape[0] == self.batch):
                msg["sample_shape"] = (self.batch,)


class substitute(_Handler):
    """Fix the value of the named sample sites (posterior / prior predictive replay)."""

    def __init__(self, data: dict):
        self.data = data

    def process(self, msg):
        if msg["type"] == "sample" and msg["name"] in self.data and not msg["is_observed"]:
            msg["value"] = torch.as_tensor(self.data[msg["name"]], dtype=torch.float64)


class trace(_Handler):
    """Record every site: ``with trace() as tr: model(...)`` then ``tr.sites``."""

    def __init__(self):
        self.sites: "OrderedDict[str, dict]" = OrderedDict()

    def postprocess(self, msg):
        if msg["name"] in self.sites:
            raise ValueError(f"duplicate site name {msg['name']!r}")
        self.sites[msg["name"]] = msg


def _apply(msg):
    for h in reversed(_STACK):
        h.process(msg)
    if msg["type"] == "sample" and msg["value"] is None:
        if msg["gen"] is None:
            raise RuntimeError(
                f"sample site {msg['name']!r} needs randomness: wrap the call in handlers.seed(...) or pass rng_key")
        msg["value"] = msg["fn"].sample(msg["gen"], msg["sample_shape"])
    for h in reversed(_STACK):
        h.postprocess(msg)
    return msg["value"]


def sample(name: str, fn, obs=None, rng_key: Optional[int] = None, sample_shape=()):
    gen = torch.Generator().manual_seed(int(rng_key)) if rng_key is not None else None
    # tensors are kept as given (log_prob converts, with a cache for constants): no copy per evaluation
    value = None if obs is None else (obs if isinstance(obs, torch.Tensor) else torch.as_tensor(obs, dtype=torch.float64))
    return _apply({"type": "sample", "name": name, "fn": fn, "value": value, "is_observed": obs is not None,
                   "gen": gen, "sample_shape": tuple(sample_shape)})


def factor(name: str, log_factor):
    """Add ``log_factor`` (a tensor, one value per chain) to the log joint -- ``numpyro.factor``.
    Used with likelihoods that are computed inside the solve kernel (``simulate(..., observe=...)``)."""
    return _apply({"type": "factor", "name": name, "fn": None, "value": log_factor, "is_observed": True,
                   "gen": None, "sample_shape": ()})


def deterministic(name: str, value):
    return _apply({"type": "deterministic", "name": name, "fn": None, "value": value, "is_observed": False,
                   "gen": None, "sample_shape": ()})
