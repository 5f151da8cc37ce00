"""dynode_amd -- MI355X-native batched ODE engine behind DynODE's simulate/infer surface.

Drop-in for the simulate path of CDCgov/DynODE: ``SimulationConfig`` / ``Initializer`` /
``SolverParams`` / ODE descriptors / ``simulate(...) -> Solution``, with diffrax's adaptive
Tsit5/Dopri5 solve and the compartmental RHS fused into one hand-written HIP kernel for gfx950.
"""

from ._abi import ModelDesc  # noqa: F401
from .config import (AgeBin, Bin, Compartment, DeterministicParameter, Dimension, Dopri5,  # noqa: F401
                     Initializer, Params, SimulationConfig, SolverParams, Strain, TransmissionParams, Tsit5)
from .infer.inference import MCMCProcess, SVIProcess  # noqa: F401
from .simulation import AbstractODEParams, PoissonObservation, Solution, SolverError, simulate  # noqa: F401

__version__ = "0.1.0"
