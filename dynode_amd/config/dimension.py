"""A named axis of a compartment (reference src/dynode/config/dimension.py:24-104)."""

from __future__ import annotations

from itertools import combinations
from math import isinf
from types import SimpleNamespace
from typing import List

from pydantic import BaseModel, Field, field_validator, model_validator

from ..typing import DynodeName
from .bins import Bin, DiscretizedPositiveIntBin, WaneBin


class Dimension(BaseModel):
    name: DynodeName = Field(description="dimension name, unique within a compartment")
    bins: List[Bin] = Field(description="bins along this axis")

    def __len__(self) -> int:
        return len(self.bins)

    @property
    def idx(self) -> SimpleNamespace:
        """bin name -> position along the axis (dimension.py:37-45)."""
        return SimpleNamespace(**{b.name: i for i, b in enumerate(self.bins)})

    @field_validator("bins", mode="after")
    @classmethod
    def _check_bins(cls, bins):
        assert len(bins) > 0, "can not have dimension with no bins"
        kind = type(bins[0])
        assert all(type(b) is kind for b in bins), "can not instantiate dimension with mixed type bins"
        names = [b.name for b in bins]
        assert len(set(names)) == len(names), "Dimension of categorical bins must have unique bin names."
        if all(isinstance(b, DiscretizedPositiveIntBin) for b in bins):
            assert bins == sorted(bins, key=lambda b: b.min_value), "DiscretizedIntBins must be sorted"
            for lo, hi in zip(bins[:-1], bins[1:]):
                assert lo.max_value < hi.min_value, "DiscretizedPositiveIntBin within a dimension can not overlap."
                assert lo.max_value + 1 == hi.min_value, "DiscretizedPositiveIntBin dimensions can not have gaps"
        return bins


class VaccinationDimension(Dimension):
    """Dose-count axis ``v0 .. vK`` (reference dimension.py:107-144): one single-valued integer bin
    per tracked dose count, zero included; a seasonal vaccine adds one more tier on top."""

    seasonal_vaccination: bool = Field(default=False, description="whether the top tier is a seasonal dose")

    def __init__(self, max_ordinal_vaccinations: int, seasonal_vaccination: bool = False, name: str = "vax"):
        tiers = max_ordinal_vaccinations + (1 if seasonal_vaccination else 0)
        super().__init__(name=name, bins=[DiscretizedPositiveIntBin(min_value=k, max_value=k, name=f"v{k}")
                                          for k in range(tiers + 1)])
        self.seasonal_vaccination = seasonal_vaccination

    @property
    def max_shots(self) -> int:
        """Highest tracked dose count (further doses do not move anyone)."""
        return len(self.bins) - 1


class ImmuneHistoryDimension(Dimension):
    """Axis recording which strains a population has recovered from."""


def _strain_names(strains) -> list:
    assert len(strains) > 0, "Must pass at least one strain to immune history dimension."
    return [s.strain_name for s in strains]


class FullStratifiedImmuneHistoryDimension(ImmuneHistoryDimension):
    """Every subset of the strains: ``none, a, b, c, a_b, a_c, b_c, a_b_c`` for three strains
    (reference dimension.py:152-171) -- 2^S bins, ordered by subset size, then by strain order."""

    def __init__(self, strains: list, name: str = "hist"):
        names = _strain_names(strains)
        subsets = [subset for size in range(1, len(names) + 1) for subset in combinations(names, size)]
        super().__init__(name=name, bins=[Bin(name="none")] + [Bin(name="_".join(sub)) for sub in subsets])


class LastStrainImmuneHistoryDimension(ImmuneHistoryDimension):
    """Only the most recent infecting strain: ``none, a, b, c`` (reference dimension.py:174-187)."""

    def __init__(self, strains: list, name: str = "hist"):
        super().__init__(name=name, bins=[Bin(name="none")] + [Bin(name=n) for n in _strain_names(strains)])


class WaneDimension(Dimension):
    """Chain of waning stages ``W0, W1, ...`` (reference dimension.py:190-244), built from parallel
    lists of waiting times and retained protections; the last stage must be absorbing."""

    def __init__(self, waiting_times: list, base_protections: list, name: str = "wane"):
        assert len(waiting_times) > 0, "Wane dimension must have at least one bin."
        assert len(waiting_times) == len(base_protections), "must pass equal length wait times and base protections"
        super().__init__(name=name, bins=[WaneBin(name=f"W{i}", waiting_time=w, base_protection=p)
                                          for i, (w, p) in enumerate(zip(waiting_times, base_protections))])

    @model_validator(mode="after")
    def _last_stage_absorbing(self):
        last = self.bins[-1]
        assert isinstance(last, WaneBin) and isinf(last.waiting_time), "last wane bin should have math.inf waiting time"
        return self
