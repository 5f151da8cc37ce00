"""Compartment / SimulationConfig and the recursive ``idx`` namespace.

Reference: src/dynode/config/simulation_config.py:28-147 -- ``config.idx.<compartment>`` is an
int (position in ``solution.ys``) carrying ``.<dimension>`` (int = axis position, WITHOUT the
leading time axis) carrying ``.<bin>`` (int = bin position); pinned by the reference's
tests/test_config/test_simulation_config.py:42-48 and test_compartment.py:21-22.
"""

from __future__ import annotations

from functools import cached_property
from types import SimpleNamespace
from typing import List

from pydantic import BaseModel, ConfigDict, model_validator

from ..typing import DynodeName
from .dimension import Dimension
from .initializer import Initializer
from .params import Params


class _IntWithAttributes(int):
    """An int that also carries named children."""

    def __new__(cls, value, **attributes):
        obj = super().__new__(cls, value)
        for key, val in attributes.items():
            setattr(obj, key, val)
        return obj

    def __str__(self) -> str:
        return str(self.__dict__)


class Compartment(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)
    name: DynodeName
    dimensions: List[Dimension]

    @model_validator(mode="after")
    def _unique_dimension_names(self):
        names = [d.name for d in self.dimensions]
        assert len(set(names)) == len(names), (
            "you can not have two identically named dimensions within a compartment")
        return self

    @property
    def shape(self) -> tuple:
        return tuple(len(d) for d in self.dimensions)

    @cached_property
    def idx(self) -> SimpleNamespace:
        ns = SimpleNamespace()
        for axis, dim in enumerate(self.dimensions):
            setattr(ns, dim.name, _IntWithAttributes(axis, **dim.idx.__dict__))
        return ns

    def __eq__(self, other) -> bool:
        return (isinstance(other, Compartment) and self.name == other.name
                and len(self.dimensions) == len(other.dimensions)
                and all(a == b for a, b in zip(self.dimensions, other.dimensions)))


class SimulationConfig(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)
    initializer: Initializer
    compartments: List[Compartment]
    parameters: Params

    @cached_property
    def idx(self) -> SimpleNamespace:
        ns = SimpleNamespace()
        for pos, comp in enumerate(self.compartments):
            setattr(ns, comp.name, _IntWithAttributes(pos, **comp.idx.__dict__))
        return ns

    @model_validator(mode="after")
    def _validate(self):
        names = [c.name for c in self.compartments]
        assert len(set(names)) == len(names), (
            f"you can not have two identically named compartments, found shared names: "
            f"{set(n for n in names if names.count(n) > 1)}")
        seen: dict = {}
        for dim in self.flatten_dims():
            if dim.name in seen:
                assert dim == seen[dim.name], (
                    f"dimension {dim.name} has different definitions across different compartments")
            else:
                seen[dim.name] = dim
        # immune-history axes must be generated from exactly the strains of transmission_params
        # (reference simulation_config.py:170-206)
        from .dimension import ImmuneHistoryDimension

        strains = self.parameters.transmission_params.strains
        for dim in self.flatten_dims():
            if isinstance(dim, ImmuneHistoryDimension):
                assert type(dim) is not ImmuneHistoryDimension and type(dim)(strains) == dim, (
                    "Found immune states that dont correlate with strains from transmission_params")
        self._encode_introduction_ages(strains)
        return self

    def _encode_introduction_ages(self, strains) -> None:
        """Externally introduced strains name the age bins they arrive in; those must be age bins of
        the model, and are turned into a 0/1 vector over the model's age axis
        (``Strain.introduction_ages_mask_vector``; reference simulation_config.py:208-264)."""
        from .bins import AgeBin

        model_ages = [b for b in self.flatten_bins() if isinstance(b, AgeBin)]
        for s in strains:
            if s.is_introduced and s.introduction_ages is not None:
                assert all(target in model_ages for target in s.introduction_ages), (
                    f"{s.strain_name} attempts to introduce itself using {s.introduction_ages} age bins, "
                    "but those are not found within the age structure of the model.")
        if not any(s.introduction_ages is not None for s in strains):
            return
        age_axis = next((d.bins for d in self.flatten_dims() if isinstance(d.bins[0], AgeBin)), [])
        assert len(age_axis) > 0, ("attempted to encode introduction_ages but could not find any age "
                                   "structure in the compartments")
        for s in strains:
            wanted = s.introduction_ages or []
            s.introduction_ages_mask_vector = [int(b in wanted) for b in age_axis]

    def get_compartment(self, compartment_name: str) -> Compartment:
        for comp in self.compartments:
            if comp.name == compartment_name:
                return comp
        raise AssertionError(
            "Compartment with name %s not found in model, found only these names: %s"
            % (compartment_name, str([c.name for c in self.compartments])))

    def flatten_bins(self) -> list:
        return [b for c in self.compartments for d in c.dimensions for b in d.bins]

    def flatten_dims(self) -> list:
        return [d for c in self.compartments for d in c.dimensions]
