"""dynode_amd -- MI355X-native batched ODE engine behind DynODE's simulate/infer surface.

Drop-in for the simulate path of CDCgov/DynODE: ``SimulationConfig`` / ``Initializer`` /
``SolverParams`` / ODE descriptors / ``simulate(...) -> Solution``, with diffrax's adaptive
Tsit5/Dopri5 solve and the compartmental RHS fused into one hand-written HIP kernel for gfx950.
"""

from . import config, infer, seip, simulation, typing, utils  # noqa: F401
from ._abi import ModelDesc  # noqa: F401
from .config import *  # noqa: F401,F403  (every name of dynode.config, see config/__init__.py)
from .infer import (InferenceProcess, MCMCProcess, SVIProcess, checkpoint_compartment_sizes,  # noqa: F401
                    resolve_deterministic, sample_distributions, sample_then_resolve)
from .simulation import AbstractODEParams, PoissonObservation, Solution, SolverError, simulate  # noqa: F401
from .typing import (CompartmentGradients, CompartmentState, CompartmentTimeseries, DynodeName,  # noqa: F401
                     ObservedData, ODE_Eqns, UnitIntervalFloat)
from .utils import (base_equation, conditional_knots, date_to_sim_day,  # noqa: F401
                    drop_keys_with_substring, evaluate_cubic_spline, flatten_list_parameters,
                    identify_distribution_indexes, sim_day_to_date, vectorize_objects)

__version__ = "0.1.0"
