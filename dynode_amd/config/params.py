"""Solver and transmission parameters (reference src/dynode/config/params.py:24-164)."""

from __future__ import annotations

from typing import Any, List

from pydantic import (BaseModel, ConfigDict, Field, NonNegativeFloat, PositiveFloat, PositiveInt,
                      field_validator, model_validator)

from .strains import Strain


class _Solver:
    """Marker for the stepper, standing in for diffrax.AbstractSolver (params.py:5,28-29)."""

    method = "tsit5"

    def __repr__(self) -> str:
        return f"{type(self).__name__}()"

    def __eq__(self, other) -> bool:
        return type(self) is type(other)

    def __hash__(self) -> int:
        return hash(type(self))


class Tsit5(_Solver):
    """Tsitouras 5(4), the reference default (params.py:28-29)."""

    method = "tsit5"


class Dopri5(_Solver):
    """Dormand-Prince 5(4)."""

    method = "dopri5"


class SolverParams(BaseModel):
    """Knobs of the ODE solve; defaults and validation as params.py:24-67."""

    model_config = ConfigDict(arbitrary_types_allowed=True)
    solver_method: _Solver = Field(default_factory=Tsit5)
    ode_solver_rel_tolerance: PositiveFloat = 1e-5
    ode_solver_abs_tolerance: PositiveFloat = 1e-6
    max_steps: PositiveInt = int(1e6)
    constant_step_size: NonNegativeFloat = 0
    discontinuity_points: list[float] = Field(default_factory=list)


class TransmissionParams(BaseModel):
    """Transmission parameters; extra fields (contact_matrix, waning_period ...) are allowed."""

    model_config = ConfigDict(arbitrary_types_allowed=True, extra="allow")
    strain_interactions: dict[str, dict[str, Any]]
    strains: List[Strain]

    @field_validator("strains", mode="before")
    @classmethod
    def _not_empty(cls, strains):
        if not strains:
            raise ValueError("strains field must contain at least one Strain.")
        return strains

    @model_validator(mode="after")
    def _interactions_cover_strains(self):
        names = {s.strain_name for s in self.strains}
        assert names == set(self.strain_interactions), (
            f"first dimension of strain_interactions must contain all strain names as keys. "
            f"Found {list(self.strain_interactions)} but expected {sorted(names)}.")
        for name, row in self.strain_interactions.items():
            assert names == set(row), f"strain_interactions[{name}] must contain all strains as keys"
        for field in ("exposed_to_infectious", "vaccine_efficacy"):
            present = [getattr(s, field) is not None for s in self.strains]
            if any(present) and not all(present):
                raise AssertionError(f"if {field} is set within one strain it must be set in all of them.")
        return self


class Params(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)
    solver_params: SolverParams
    transmission_params: TransmissionParams
