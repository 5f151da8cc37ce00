"""The simulate path (mirrors dynode.simulation)."""

from ..rhs import AbstractODEParams
from .odes import SaveAt, Solution, SolverError, build_saveat, simulate

__all__ = ["AbstractODEParams", "SaveAt", "Solution", "SolverError", "build_saveat", "simulate"]
