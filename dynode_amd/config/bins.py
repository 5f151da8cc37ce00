"""Bins of a compartment dimension (reference src/dynode/config/bins.py)."""

from __future__ import annotations

from pydantic import BaseModel, Field, NonNegativeFloat, NonNegativeInt, PositiveFloat, model_validator

from ..typing import DynodeName


class Bin(BaseModel):
    """A named cell of a dimension."""

    name: DynodeName = Field(description="bin name, unique within its dimension")


class DiscretizedPositiveIntBin(Bin):
    """Inclusive integer range [min_value, max_value]; default name ``range_<min>_<max>``."""

    min_value: NonNegativeInt
    max_value: NonNegativeInt

    def __init__(self, min_value, max_value, name=None):
        super().__init__(name=name if name is not None else f"range_{min_value}_{max_value}",
                         min_value=min_value, max_value=max_value)

    @model_validator(mode="after")
    def _ordered(self):
        assert self.min_value <= self.max_value
        return self


class AgeBin(DiscretizedPositiveIntBin):
    """Age range; default name ``a<min>_<max>`` (bins.py AgeBin)."""

    def __init__(self, min_value, max_value, name=None):
        super().__init__(min_value, max_value, name if name is not None else f"a{min_value}_{max_value}")


class WaneBin(Bin):
    """One stage of waning immunity (reference bins.py:77-89): the mean number of days spent in
    the stage (``math.inf`` = the stage is never left; waning rate = 1 / waiting_time) and the share
    of immune protection its occupants keep."""

    waiting_time: PositiveFloat = Field(description="mean days in this stage; math.inf = absorbing")
    base_protection: NonNegativeFloat = Field(le=1.0, description="retained protection in [0, 1]")
