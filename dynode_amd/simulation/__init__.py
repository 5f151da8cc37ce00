"""The simulate path (mirrors dynode.simulation)."""

from ..rhs import AbstractODEParams
from .odes import PoissonObservation, SaveAt, Solution, SolverError, build_saveat, simulate

__all__ = ["AbstractODEParams", "PoissonObservation", "SaveAt", "Solution", "SolverError", "build_saveat", "simulate"]
