"""Config / plugin schema of the simulate path (mirrors dynode.config, boundary only).

Restated from the reference's behaviour, not copied: only what the hot path's front-end
needs -- shapes, the recursive ``idx`` namespace, strain vectors, solver knobs.
Reference: /root/reference/src/dynode/config/ ("boundary only" row).
"""

from .bins import AgeBin, Bin, DiscretizedPositiveIntBin
from .deterministic_parameter import DeterministicParameter
from .dimension import Dimension
from .initializer import Initializer
from .params import Dopri5, Params, SolverParams, Tsit5, TransmissionParams
from .simulation_config import Compartment, SimulationConfig
from .strains import Strain

__all__ = [
    "AgeBin", "Bin", "DiscretizedPositiveIntBin", "DeterministicParameter", "Dimension",
    "Initializer", "Dopri5", "Params", "SolverParams", "Tsit5", "TransmissionParams",
    "Compartment", "SimulationConfig", "Strain",
]
