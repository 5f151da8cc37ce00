"""Aliases used across the drop-in surface.

Counterpart of the reference's ``dynode.typing`` (src/dynode/typing/typing.py:11-39); where the
reference says ``jax.Array`` a compartment here is a ``torch.Tensor`` in HBM (results) or a
``numpy.ndarray`` / ``torch.Tensor`` (inputs).
"""

from __future__ import annotations

import re
from typing import Annotated, Any, Callable, Tuple, Union

import numpy as np
import torch
from annotated_types import Ge, Le
from pydantic import BeforeValidator

# ---- arrays and state tuples
ArrayLike = Union[np.ndarray, torch.Tensor]
CompartmentState = Tuple[ArrayLike, ...]        # one array per compartment, in config order
CompartmentGradients = CompartmentState
CompartmentTimeseries = CompartmentState         # the same with a leading time axis
ObservedData = Union[CompartmentState, ArrayLike]
ODE_Eqns = Callable[[Any, CompartmentState, Any], CompartmentGradients]   # (t, state, params) -> d(state)/dt

# ---- constrained scalars
UnitIntervalFloat = Annotated[float, Ge(0.0), Le(1.0)]

_NAME = re.compile(r"[A-Za-z_][A-Za-z0-9_]*\Z")


def _check_identifier(name: str) -> str:
    """Names of compartments, dimensions, bins and strains become attributes of ``config.idx``, so
    they must look like identifiers: letters, digits, underscores, no leading digit, no spaces.
    Same three rejections (and messages) as the reference's validator."""
    if _NAME.match(name):
        return name
    if name[:1].isdigit():
        raise ValueError("Name can not start with a number.")
    if " " in name:
        raise ValueError("Name can not have spaces.")
    raise ValueError("Name can only contain alphanumerics or underscores.")


DynodeName = Annotated[str, BeforeValidator(_check_identifier)]


def is_array(x: Any) -> bool:
    """True for the array types ``simulate`` accepts as compartment values."""
    return isinstance(x, (np.ndarray, torch.Tensor))
