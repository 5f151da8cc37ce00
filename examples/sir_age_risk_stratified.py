"""Age x risk SIR on dynode_amd -- counterpart of the reference's examples/sir_age_risk_stratified.py."""

from datetime import date

import numpy as np

from dynode_amd import (Bin, Compartment, Dimension, Initializer, Params, SimulationConfig, SolverParams, Strain,
                        TransmissionParams, simulate)
from dynode_amd.rhs import SIR_ODEParams, sir_age_risk_ode as sir_ode  # noqa: F401


class SIRInitializer(Initializer):
    age_demographics: list
    risk_prop: list
    s0_prop: float
    i0_prop: float

    def __init__(self, age_demographics, risk_prop, s0_prop=0.99, i0_prop=0.01):
        super().__init__(description="An age x risk SIR initializer", initialize_date=date(2022, 2, 11),
                         population_size=1000, age_demographics=list(age_demographics), risk_prop=list(risk_prop),
                         s0_prop=s0_prop, i0_prop=i0_prop)

    def get_initial_state(self, **kwargs):
        pop = self.population_size * np.outer(self.age_demographics, self.risk_prop)  # (A, R)
        return (pop * self.s0_prop, pop * self.i0_prop, np.zeros_like(pop))


def contact_tensor(age_contact_matrix, risk_contact_matrix) -> np.ndarray:
    """einsum("ij,kl->ikjl") of the two matrices (reference :113-115; pinned by its
    tests/test_age_risk_groups/test_age_risk_groups.py:12-138)."""
    return np.einsum("ij,kl->ikjl", np.asarray(age_contact_matrix, float), np.asarray(risk_contact_matrix, float))


def get_config(age_demographics=(0.25, 0.5, 0.25), risk_prop=(0.8, 0.2),
               age_contact_matrix=((1.0, 0.3, 0.1), (0.3, 1.0, 0.3), (0.1, 0.3, 1.0)),
               risk_contact_matrix=((1.0, 0.5), (0.5, 1.0)), r_0=2.0, infectious_period=7.0) -> SimulationConfig:
    age = Dimension(name="age", bins=[Bin(name=f"age_{k}") for k in range(len(age_demographics))])
    risk = Dimension(name="risk", bins=[Bin(name=f"risk_{k}") for k in range(len(risk_prop))])
    comps = [Compartment(name=n, dimensions=[age, risk]) for n in ("s", "i", "r")]
    tp = TransmissionParams(strains=[Strain(strain_name="swo9", r0=r_0, infectious_period=infectious_period)],
                            strain_interactions={"swo9": {"swo9": 1.0}},
                            contact_matrix=contact_tensor(age_contact_matrix, risk_contact_matrix))
    return SimulationConfig(compartments=comps, initializer=SIRInitializer(age_demographics, risk_prop),
                            parameters=Params(solver_params=SolverParams(), transmission_params=tp))


def get_odeparams(config: SimulationConfig) -> SIR_ODEParams:
    tp = config.parameters.transmission_params
    strain = tp.strains[0]
    return SIR_ODEParams(beta=np.asarray(strain.r0 / strain.infectious_period),
                         gamma=np.asarray(1 / strain.infectious_period), contact_matrix=tp.contact_matrix)


if __name__ == "__main__":
    config = get_config()
    sol = simulate(ode=sir_ode, duration_days=150, initial_state=config.initializer.get_initial_state(),
                   ode_parameters=get_odeparams(config), solver_parameters=config.parameters.solver_params)
    print("recovered by (age, risk) at day 150:\n", sol.ys[config.idx.r][-1].cpu().numpy())
