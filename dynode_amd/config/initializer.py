"""Abstract initial-state builder (reference src/dynode/config/initializer.py:12-47)."""

from __future__ import annotations

from datetime import date

from pydantic import BaseModel, Field, PositiveInt

from ..typing import CompartmentState


class Initializer(BaseModel):
    description: str = Field(description="what this initializer does / its data streams")
    initialize_date: date
    population_size: PositiveInt

    def get_initial_state(self, **kwargs) -> CompartmentState:
        raise NotImplementedError("implement functionality to get initial state")
