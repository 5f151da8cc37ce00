"""Recursive sampling / resolution of parameter objects and the site names it defines.

Mirror of /root/reference/src/dynode/infer/sample.py:18-197.  The naming rules fix the keys of
every posterior dict: fields of pydantic models / dicts are prefixed ``<key>_``, list elements
``<i>_``, and the trailing underscore is dropped at the sample site -- e.g.
``strains_0_r0``, ``strains_0_infectious_period`` (reference tests/test_infer/test_sample.py:49-151).
"""

from __future__ import annotations

from copy import deepcopy
from typing import Any

import numpy as np
from pydantic import BaseModel

from ..config import DeterministicParameter
from . import handlers
from .distributions import Distribution


def _is_container(obj) -> bool:
    # numeric arrays are leaves (the reference's jax arrays are not np.ndarray either)
    return isinstance(obj, list) or (isinstance(obj, np.ndarray) and obj.dtype == object)


def _rebuild(obj, values: dict):
    if isinstance(obj, dict):
        return dict(values)
    return obj.__class__(**values)


def sample_distributions(obj: Any, rng_key=None, _prefix: str = ""):
    """Replace every Distribution found in ``obj`` by a draw recorded at its site name."""
    if isinstance(obj, (BaseModel, dict)):
        out = {k: sample_distributions(v, rng_key=rng_key, _prefix=_prefix + f"{k}_") for k, v in dict(obj).items()}
        return _rebuild(obj, out)
    if _is_container(obj):
        return [sample_distributions(v, rng_key=rng_key, _prefix=_prefix + f"{i}_") for i, v in enumerate(obj)]
    if isinstance(obj, Distribution):
        return handlers.sample(_prefix[:-1] if _prefix else _prefix, obj, rng_key=rng_key)
    return obj


def resolve_deterministic(obj: Any, root_params, _prefix: str = ""):
    """Replace every DeterministicParameter by the value it points to inside ``root_params``."""
    if isinstance(root_params, BaseModel):
        root_params = dict(root_params)
    if isinstance(obj, (BaseModel, dict)):
        out = {k: resolve_deterministic(v, root_params, _prefix=_prefix + f"{k}_") for k, v in dict(obj).items()}
        return _rebuild(obj, out)
    if _is_container(obj):
        return [resolve_deterministic(v, root_params, _prefix=_prefix + f"{i}_") for i, v in enumerate(obj)]
    if isinstance(obj, DeterministicParameter):
        return handlers.deterministic(_prefix[:-1] if _prefix else _prefix, obj.resolve(root_params))
    return obj


def sample_then_resolve(parameters: Any, rng_key=None, _prefix: str = ""):
    """Deep-copy, sample, resolve -- the copy keeps chains / batches from interfering (sample.py:190)."""
    parameters = deepcopy(parameters)
    parameters = sample_distributions(parameters, rng_key=rng_key, _prefix=_prefix)
    return resolve_deterministic(parameters, root_params=dict(parameters), _prefix=_prefix)
