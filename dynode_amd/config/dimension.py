"""A named axis of a compartment (reference src/dynode/config/dimension.py:24-104)."""

from __future__ import annotations

from types import SimpleNamespace
from typing import List

from pydantic import BaseModel, Field, field_validator

from ..typing import DynodeName
from .bins import Bin, DiscretizedPositiveIntBin


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
