"""Type aliases of the drop-in surface (mirrors /root/reference/src/dynode/typing/typing.py:11-39).

The reference's arrays are ``jax.Array``; here a compartment array is a ``torch.Tensor``
(device memory) or a ``numpy.ndarray`` on the way in.
"""

from __future__ import annotations

from typing import Annotated, Any, Callable, Tuple, Union

import numpy as np
import torch
from annotated_types import Ge, Le
from pydantic import BeforeValidator

ArrayLike = Union[np.ndarray, torch.Tensor]
CompartmentState = Tuple[ArrayLike, ...]
CompartmentGradients = Tuple[ArrayLike, ...]
CompartmentTimeseries = CompartmentState
UnitIntervalFloat = Annotated[float, Ge(0.0), Le(1.0)]
ODE_Eqns = Callable[[Any, CompartmentState, Any], CompartmentGradients]
ObservedData = Union[Tuple[ArrayLike, ...], ArrayLike]


def _verify_name(name: str) -> str:
    """No leading digit, no spaces, alphanumerics/underscores only (typing.py:27-36)."""
    if name[0].isnumeric():
        raise ValueError("Name can not start with a number.")
    if " " in name:
        raise ValueError("Name can not have spaces.")
    if not all(ch.isalnum() or ch == "_" for ch in name):
        raise ValueError("Name can only contain alphanumerics or underscores.")
    return name


DynodeName = Annotated[str, BeforeValidator(_verify_name)]


def is_array(x: Any) -> bool:
    return isinstance(x, (np.ndarray, torch.Tensor))
