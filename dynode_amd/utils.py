"""Host helpers of ``dynode.utils`` (reference src/dynode/utils/{utils,splines,datetime_utils}.py).

Only ``vectorize_objects`` and the simulation-day helpers sit on the simulate/infer path
(``get_odeparams`` uses the former); the spline evaluators and key helpers are
what the reference's post-processing and vaccination-rate code call.  Everything here is plain
NumPy / Python, accepts torch tensors where arrays are expected, and has no plotting or logging.
"""

from __future__ import annotations

import datetime
from typing import Any, Callable, Dict, List, Optional

import numpy as np


# ------------------------------------------------------------------------------ object helpers
def vectorize_objects(objs: List[Any], target: str, filter: Callable[[Any], bool] = lambda _: True) -> list:
    """The ``target`` attribute of every object that passes ``filter``
    (reference utils/utils.py:10-38).  An object that passes the filter but lacks the attribute
    raises ``AttributeError``, exactly as a bare ``getattr`` does."""
    assert isinstance(target, str), "target must be a string"
    return [getattr(o, target) for o in objs if filter(o)]


def drop_keys_with_substring(dct: Dict[str, Any], drop_s: str) -> Dict[str, Any]:
    """Remove, in place, every key that contains ``drop_s``; returns the same dict (utils.py:86-105)."""
    for key in [k for k in dct if drop_s in k]:
        del dct[key]
    return dct


def _is_array(v) -> bool:
    return isinstance(v, np.ndarray) or type(v).__module__.split(".")[0] in ("torch", "jax", "jaxlib")


def flatten_list_parameters(samples: Dict[str, Any]) -> Dict[str, Any]:
    """Split plated posterior sites into one key per element (utils.py:41-83).

    A value of shape ``(chains, draws, *plate)`` becomes ``prod(plate)`` entries named
    ``key_i`` / ``key_i_j`` ... of shape ``(chains, draws)``; values with at most two axes are kept.
    """
    out: Dict[str, Any] = {}
    for key, value in samples.items():
        if _is_array(value) and value.ndim > 2:
            for idx in np.ndindex(*tuple(value.shape[2:])):
                out[key + "".join(f"_{i}" for i in idx)] = value[(slice(None), slice(None)) + idx]
        else:
            out[key] = value
    return out


def identify_distribution_indexes(parameters: Dict[str, Any]) -> Dict[str, dict]:
    """Where the sampled sites of a parameter dict come from -- the inverse of the site naming of
    ``sample_distributions`` (utils.py:108-178): ``{"test": [0, Normal(), 2], "example": Normal()}``
    -> ``{"test_1": {"sample_name": "test", "sample_idx": (1,)},
    "example": {"sample_name": "example", "sample_idx": None}}``."""
    from .infer.distributions import Distribution

    found: Dict[str, dict] = {}
    for key, param in parameters.items():
        if isinstance(param, Distribution):
            found[key] = {"sample_name": key, "sample_idx": None}
        elif isinstance(param, (list, tuple, np.ndarray)):
            arr = param if isinstance(param, np.ndarray) else _object_array(param)
            for idx in np.ndindex(*arr.shape):
                if isinstance(arr[idx], Distribution):
                    found[key + "".join(f"_{i}" for i in idx)] = {"sample_name": key,
                                                                  "sample_idx": tuple(int(i) for i in idx)}
    return found


def _object_array(nested) -> np.ndarray:
    """Nested lists -> object ndarray without NumPy trying to interpret the leaves."""
    shape = []
    probe = nested
    while isinstance(probe, (list, tuple)):
        shape.append(len(probe))
        probe = probe[0] if len(probe) else None
    arr = np.empty(tuple(shape), dtype=object)
    for idx in np.ndindex(*arr.shape):
        leaf = nested
        for i in idx:
            leaf = leaf[i]
        arr[idx] = leaf
    return arr


# ------------------------------------------------------------------------------ cubic splines
# Vaccination-rate splines of the reference (utils/splines.py): per (age bin, dose count) a cubic
# a + b t + c t^2 + d t^3 plus truncated-power terms sum_i coef_i * (t - knot_i)^3 * [t > knot_i].
def _np(a):
    return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)


def base_equation(t, coefficients):
    """``a + b t + c t^2 + d t^3`` over the last axis of ``coefficients`` (shape ``(..., 4)``)."""
    c = _np(coefficients)
    powers = np.array([1.0, t, t * t, t * t * t], dtype=np.result_type(c.dtype, np.float64))
    return (c * powers).sum(axis=-1)


def conditional_knots(t, knots, coefficients):
    """``sum_i coefficients[..., i] * (t - knots[..., i])^3`` over the knots already passed (``t > knot``)."""
    k, c = _np(knots), _np(coefficients)
    past = np.where(t > k, t - k, 0)
    return (past ** 3 * c).sum(axis=-1)


def evaluate_cubic_spline(t, knot_locations, base_equations, knot_coefficients):
    """Spline value on day ``t`` for every (age bin, dose count): base cubic + active knot terms.
    Shapes: knots / knot coefficients ``(..., n_knots)``, base ``(..., 4)`` -> ``(...)``."""
    return base_equation(t, base_equations) + conditional_knots(t, knot_locations, knot_coefficients)


# ------------------------------------------------------------------------------ dates
def date_to_sim_day(date: datetime.date, init_date: datetime.date) -> int:
    """Days since the model's initialisation date (reference utils/datetime_utils.py:64-88)."""
    return (date - init_date).days


def sim_day_to_date(sim_day: int, init_date: datetime.date) -> datetime.date:
    """The calendar date of simulation day ``sim_day`` (day 0 = ``init_date``)."""
    return init_date + datetime.timedelta(days=int(sim_day))
