"""Age-stratified SIR on dynode_amd -- counterpart of the reference's examples/sir_age_stratified.py."""

from datetime import date

import numpy as np

from dynode_amd import (Bin, Compartment, Dimension, Initializer, Params, SimulationConfig, SolverParams, Strain,
                        TransmissionParams, simulate)
from dynode_amd.infer import sample_then_resolve
from dynode_amd.rhs import SIR_ODEParams, sir_ode  # noqa: F401


class SIRInitializer(Initializer):
    def __init__(self):
        super().__init__(description="An SIR initalizer", initialize_date=date(2022, 2, 11), population_size=1000)

    def get_initial_state(self, s0_prop=0.99, i0_prop=0.01, **kwargs):
        assert s0_prop + i0_prop == 1.0, f"s0_prop and i0_prop must sum to 1.0, got {s0_prop} and {i0_prop}."
        age_demographics = np.array([0.75, 0.25])  # young : old
        return (self.population_size * s0_prop * age_demographics, self.population_size * i0_prop * age_demographics,
                np.array([0.0, 0.0]))


def get_config(r_0=2.0, infectious_period=7.0) -> SimulationConfig:
    dimension = Dimension(name="age", bins=[Bin(name="young"), Bin(name="old")])
    comps = [Compartment(name=n, dimensions=[dimension]) for n in ("s", "i", "r")]
    contact_matrix = np.array([[0.7, 0.3], [0.3, 0.7]])
    contact_matrix = contact_matrix / np.max(np.real(np.linalg.eigvals(contact_matrix)))  # spectral radius 1
    params = Params(solver_params=SolverParams(),
                    transmission_params=TransmissionParams(
                        strains=[Strain(strain_name="swo9", r0=r_0, infectious_period=infectious_period)],
                        strain_interactions={"swo9": {"swo9": 1.0}}, contact_matrix=contact_matrix))
    return SimulationConfig(compartments=comps, initializer=SIRInitializer(), parameters=params)


def get_odeparams(config: SimulationConfig) -> SIR_ODEParams:
    tp = sample_then_resolve(config.parameters.transmission_params)
    strain = tp.strains[0]
    # plain arithmetic: r0 / infectious_period may be floats, arrays or (under NUTS) torch tensors
    return SIR_ODEParams(beta=strain.r0 / strain.infectious_period, gamma=1 / strain.infectious_period,
                         contact_matrix=tp.contact_matrix)


def run_simulation(config: SimulationConfig, tf, observe=None):
    return simulate(ode=sir_ode, duration_days=tf, initial_state=config.initializer.get_initial_state(SIRConfig=config),
                    ode_parameters=get_odeparams(config), solver_parameters=config.parameters.solver_params,
                    observe=observe)


if __name__ == "__main__":
    config = get_config()
    sol = run_simulation(config, 150)
    s, i, r = [a.cpu().numpy() for a in sol.ys]  # each (151, 2)
    for d in (0, 50, 100, 150):
        print(d, s[d], i[d], r[d])
