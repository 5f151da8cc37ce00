"""Strain definition (reference src/dynode/config/strains.py:20-109), fields used by the path."""

from __future__ import annotations

from typing import Any, List, Optional

from pydantic import BaseModel, ConfigDict, Field

from ..typing import DynodeName
from .bins import AgeBin


class Strain(BaseModel):
    """A pathogen strain.  ``r0`` / ``infectious_period`` may be numbers, arrays, distributions
    (sampled by ``dynode_amd.infer.sample_then_resolve``) or DeterministicParameter links."""

    model_config = ConfigDict(arbitrary_types_allowed=True)
    strain_name: DynodeName
    r0: Any
    infectious_period: Any
    exposed_to_infectious: Optional[Any] = None
    vaccine_efficacy: Optional[dict] = None
    is_introduced: bool = False
    introduction_time: Optional[Any] = None
    introduction_percentage: Optional[Any] = None
    introduction_scale: Optional[Any] = None
    introduction_ages: Optional[List[AgeBin]] = None
    introduction_ages_mask_vector: Optional[List[int]] = Field(default=None)
