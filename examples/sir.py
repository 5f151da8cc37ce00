"""Single-population SIR on dynode_amd (the smallest model of the family).

Covers the same ground as the reference's ``examples/sir.py`` -- an Initializer, a
SimulationConfig, a parameter container, one ``simulate`` call -- and then shows what the GPU
engine adds: the same call with a vector of R0 values integrates all of them in one launch.
"""

from datetime import date

import numpy as np

import dynode_amd as dyn
from dynode_amd.rhs import SIR_ODEParams, sir_ode

POPULATION = 1


class SimpleSIRInitializer(dyn.Initializer):
    """Fractions of one unit population: 90 % susceptible, 10 % infectious unless told otherwise."""

    def __init__(self):
        super().__init__(description="fractions of a unit population", initialize_date=date(2022, 2, 11),
                         population_size=POPULATION)

    def get_initial_state(self, s_0=0.9, i_0=0.1, r_0=0.0, **kwargs):
        return tuple(np.array([float(v)]) for v in (s_0, i_0, r_0))


def get_config(r_0=2.0, infectious_period=7.0) -> dyn.SimulationConfig:
    everyone = dyn.Dimension(name="age", bins=[dyn.Bin(name="all")])
    transmission = dyn.TransmissionParams(
        strains=[dyn.Strain(strain_name="test", r0=r_0, infectious_period=infectious_period)],
        strain_interactions={"test": {"test": 1.0}},
        contact_matrix=np.ones((1, 1)),
    )
    return dyn.SimulationConfig(
        compartments=[dyn.Compartment(name=c, dimensions=[everyone]) for c in "sir"],
        initializer=SimpleSIRInitializer(),
        parameters=dyn.Params(solver_params=dyn.SolverParams(), transmission_params=transmission),
    )


def get_odeparams(config: dyn.SimulationConfig) -> SIR_ODEParams:
    """beta = R0 / T_inf, gamma = 1 / T_inf."""
    strain = config.parameters.transmission_params.strains[0]
    t_inf = strain.infectious_period
    return SIR_ODEParams(beta=np.asarray(strain.r0 / t_inf), gamma=np.asarray(1.0 / t_inf))


def main():
    config = get_config()
    state0 = config.initializer.get_initial_state()
    solver = config.parameters.solver_params
    sol = dyn.simulate(ode=sir_ode, duration_days=150, initial_state=state0, ode_parameters=get_odeparams(config),
                       solver_parameters=solver)
    s, i, r = (c.squeeze().cpu().numpy() for c in sol.ys)
    print("day      S       I       R")
    for day in (0, 30, 60, 90, 150):
        print(f"{day:3d}  {s[day]:.4f}  {i[day]:.4f}  {r[day]:.4f}")
    # the batched extension: 1000 values of R0 in one launch, final sizes back as one tensor
    r0 = np.linspace(1.1, 4.0, 1000)
    sweep = dyn.simulate(ode=sir_ode, duration_days=300, initial_state=state0,
                         ode_parameters=SIR_ODEParams(beta=r0 / 7.0, gamma=np.asarray(1 / 7.0)), solver_parameters=solver)
    final = sweep.ys[config.idx.r][:, -1, 0].cpu().numpy()
    print("final size for R0 = 1.1, 2.0, 4.0:", final[0].round(3), final[310].round(3), final[-1].round(3))


if __name__ == "__main__":
    main()
