"""A parameter whose value is read from another parameter (reference deterministic_parameter.py:8-76)."""

from __future__ import annotations

from typing import Any, Callable, Optional


class DeterministicParameter:
    def __init__(self, depends_on: str, index: Optional[int | tuple | slice] = None,
                 transform: Callable[[Any], Any] = lambda x: x):
        self.depends_on = depends_on
        self.index = index
        self.transform = transform

    def resolve(self, parameter_state: dict) -> Any:
        """``transform(parameter_state[depends_on][index])``; a descriptive Exception otherwise."""
        try:
            value = parameter_state[self.depends_on]
            return self.transform(value if self.index is None else value[self.index])
        except Exception as err:  # same contract as the reference: one wrapped Exception
            where = self.depends_on if self.index is None else f"{self.depends_on}[{self.index}]"
            raise Exception(
                f"Was unable to find {where} within the following scope, make sure "
                f"DeterministicParameter dependencies are at the top level of the configuration "
                f"object. Scope: {parameter_state}") from err
