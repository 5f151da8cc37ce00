"""A second strain arriving from outside: multi-strain, age-stratified SEIRS in which strain B is
absent at the start and is seeded by infectious visitors around day 60.

The reference describes this mechanism in its configuration (``Strain.is_introduced``,
``introduction_time`` / ``_percentage`` / ``_scale`` / ``_ages``, src/dynode/config/strains.py:53-109)
and in ode_model.md ("I_b + N(mu, sigma) * phi * P_b" inside the force of infection) but ships no
example for it; here the same fields drive the kernel's introduction term.
"""

from datetime import date

import numpy as np

from dynode_amd import (AgeBin, Bin, Compartment, Dimension, Initializer, Params, SimulationConfig, SolverParams, Strain,
                        TransmissionParams, simulate)
from dynode_amd.rhs import SEIRS_MultiStrain_ODEParams, introduction_params, seirs_multi_strain_ode

AGES = [AgeBin(min_value=0, max_value=17), AgeBin(min_value=18, max_value=64), AgeBin(min_value=65, max_value=99)]


class ResidentStrainInitializer(Initializer):
    """Everyone susceptible except 1 % infected with the resident strain (the first one)."""

    def __init__(self, population_size=100_000, age_demographics=(0.22, 0.61, 0.17)):
        super().__init__(description="one resident strain, the others arrive later", initialize_date=date(2022, 2, 11),
                         population_size=population_size)
        self._demo = np.asarray(age_demographics, dtype=float)

    def get_initial_state(self, config: SimulationConfig, i0_prop=0.01, **kwargs):
        pop = self.population_size * self._demo
        shape = config.get_compartment("i").shape
        i_0 = np.zeros(shape)
        i_0[:, 0] = i0_prop * pop
        zeros = np.zeros(shape)
        return (pop - i_0.sum(1), zeros.copy(), i_0, zeros.copy(), zeros.copy())


def get_config(introduction_time=60.0, introduction_percentage=0.005, introduction_scale=5.0,
               introduction_ages=(AGES[1],)) -> SimulationConfig:
    strains = [
        Strain(strain_name="resident", r0=1.8, infectious_period=7.0, exposed_to_infectious=3.0),
        Strain(strain_name="newcomer", r0=2.6, infectious_period=6.0, exposed_to_infectious=2.5, is_introduced=True,
               introduction_time=introduction_time, introduction_percentage=introduction_percentage,
               introduction_scale=introduction_scale, introduction_ages=list(introduction_ages)),
    ]
    names = [s.strain_name for s in strains]
    age = Dimension(name="age", bins=AGES)
    strain_dim = Dimension(name="strain", bins=[Bin(name=n) for n in names])
    comps = [Compartment(name="s", dimensions=[age])] + [Compartment(name=c, dimensions=[age, strain_dim])
                                                          for c in ("e", "i", "r", "c")]
    contact = np.array([[0.60, 0.35, 0.05], [0.20, 0.65, 0.15], [0.10, 0.45, 0.45]])
    contact = contact / np.max(np.real(np.linalg.eigvals(contact)))
    tp = TransmissionParams(strains=strains, strain_interactions={a: {b: 1.0 for b in names} for a in names},
                            contact_matrix=contact, waning_period=(120.0, 120.0))
    return SimulationConfig(compartments=comps, initializer=ResidentStrainInitializer(),
                            parameters=Params(solver_params=SolverParams(), transmission_params=tp))


def get_odeparams(config: SimulationConfig) -> SEIRS_MultiStrain_ODEParams:
    tp = config.parameters.transmission_params
    r0 = np.array([s.r0 for s in tp.strains], dtype=float)
    t_inf = np.array([s.infectious_period for s in tp.strains], dtype=float)
    t_lat = np.array([s.exposed_to_infectious for s in tp.strains], dtype=float)
    return SEIRS_MultiStrain_ODEParams(
        beta=r0 / t_inf, gamma=1.0 / t_inf, sigma=1.0 / t_lat, omega=1.0 / np.array(tp.waning_period, dtype=float),
        contact_matrix=tp.contact_matrix, idx=config.idx,
        introduction_params=introduction_params(tp.strains, config.initializer.initialize_date))


def run_simulation(config: SimulationConfig, tf=300):
    return simulate(ode=seirs_multi_strain_ode, duration_days=tf, initial_state=config.initializer.get_initial_state(config),
                    ode_parameters=get_odeparams(config), solver_parameters=config.parameters.solver_params)


if __name__ == "__main__":
    config = get_config()
    sol = run_simulation(config)
    i = sol.ys[config.idx.i].cpu().numpy()          # (301, 3 ages, 2 strains)
    for day in (0, 50, 60, 70, 100, 150, 300):
        print(f"day {day:3d}  infectious resident {i[day, :, 0].sum():10.1f}   newcomer {i[day, :, 1].sum():10.1f}")
