"""Minimal SIR model on dynode_amd -- counterpart of the reference's examples/sir.py.

Same structure as the reference example (Initializer, get_config, get_odeparams, simulate);
the only change a user makes is the import line and using the ``sir_ode`` descriptor.
"""

from datetime import date

import numpy as np

from dynode_amd import (Bin, Compartment, Dimension, Initializer, Params, SimulationConfig, SolverParams, Strain,
                        TransmissionParams, simulate)
from dynode_amd.rhs import SIR_ODEParams, sir_ode  # noqa: F401


class SimpleSIRInitializer(Initializer):
    def __init__(self):
        super().__init__(description="Simple SIR initializer", initialize_date=date(2022, 2, 11), population_size=1)

    def get_initial_state(self, s_0=0.9, i_0=0.1, r_0=0.0, **kwargs):
        return (np.array([s_0]), np.array([i_0]), np.array([r_0]))


def get_config(r_0=2.0, infectious_period=7.0) -> SimulationConfig:
    dimension = Dimension(name="age", bins=[Bin(name="all")])
    comps = [Compartment(name=n, dimensions=[dimension]) for n in ("s", "i", "r")]
    strain = [Strain(strain_name="test", r0=r_0, infectious_period=infectious_period)]
    params = Params(solver_params=SolverParams(),
                    transmission_params=TransmissionParams(strains=strain, strain_interactions={"test": {"test": 1.0}},
                                                           contact_matrix=np.array([[1.0]])))
    return SimulationConfig(compartments=comps, initializer=SimpleSIRInitializer(), parameters=params)


def get_odeparams(config: SimulationConfig) -> SIR_ODEParams:
    strain = config.parameters.transmission_params.strains[0]
    return SIR_ODEParams(beta=np.array(strain.r0 / strain.infectious_period), gamma=np.array(1.0 / strain.infectious_period))


if __name__ == "__main__":
    config = get_config()
    sol = simulate(ode=sir_ode, duration_days=150, initial_state=config.initializer.get_initial_state(),
                   ode_parameters=get_odeparams(config), solver_parameters=config.parameters.solver_params)
    s, i, r = [arr.squeeze().cpu().numpy() for arr in sol.ys]
    print("day   S      I      R")
    for d in (0, 30, 60, 90, 150):
        print(f"{d:3d} {s[d]:.4f} {i[d]:.4f} {r[d]:.4f}")
