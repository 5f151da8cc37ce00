"""SEIRS on dynode_amd -- counterpart of the reference's examples/seirs.py (+ seasonal variant)."""

from datetime import date

import numpy as np

from dynode_amd import (Bin, Compartment, Dimension, Initializer, Params, SimulationConfig, SolverParams, Strain,
                        TransmissionParams, simulate)
from dynode_amd.rhs import (SEIRS_ODEParams, SEIRS_Seasonal_ODEParams, SeasonalityParams, seirs_ode,  # noqa: F401
                            seirs_ode_seasonal)


class SimpleSEIRSInitializer(Initializer):
    def __init__(self):
        super().__init__(description="Simple SEIRS initializer", initialize_date=date(2022, 2, 11), population_size=1)

    def get_initial_state(self, s_0=0.99, e_0=0.0, i_0=0.01, r_0=0.0, **kwargs):
        return tuple(np.array([v]) for v in (s_0, e_0, i_0, r_0))


def get_config(r_0=2.0, infectious_period=7.0, latent_period=3.0, waning_period=60.0) -> SimulationConfig:
    dimension = Dimension(name="age", bins=[Bin(name="all")])
    comps = [Compartment(name=n, dimensions=[dimension]) for n in ("s", "e", "i", "r")]
    tp = TransmissionParams(strains=[Strain(strain_name="test", r0=r_0, infectious_period=infectious_period)],
                            strain_interactions={"test": {"test": 1.0}}, contact_matrix=np.array([[1.0]]),
                            latent_period=latent_period, waning_period=waning_period)
    return SimulationConfig(compartments=comps, initializer=SimpleSEIRSInitializer(),
                            parameters=Params(solver_params=SolverParams(), transmission_params=tp))


def get_seirs_odeparams(config: SimulationConfig) -> SEIRS_ODEParams:
    tp = config.parameters.transmission_params
    strain = tp.strains[0]
    return SEIRS_ODEParams(beta=np.array(strain.r0 / strain.infectious_period),
                           gamma=np.array(1.0 / strain.infectious_period), sigma=np.array(1.0 / tp.latent_period),
                           omega=np.array(1.0 / tp.waning_period))


def get_seasonal_odeparams(config, forcing_amp=0.2, forcing_phase=0.0, forcing_period=365.0) -> SEIRS_Seasonal_ODEParams:
    base = get_seirs_odeparams(config)
    return SEIRS_Seasonal_ODEParams(beta=base.beta, gamma=base.gamma, sigma=base.sigma, omega=base.omega,
                                    seasonality_params=SeasonalityParams(forcing_amp=forcing_amp,
                                                                         forcing_phase=forcing_phase,
                                                                         forcing_period=forcing_period))


if __name__ == "__main__":
    config = get_config()
    y0 = config.initializer.get_initial_state()
    sol = simulate(ode=seirs_ode, duration_days=300, initial_state=y0, ode_parameters=get_seirs_odeparams(config),
                   solver_parameters=config.parameters.solver_params)
    print("SEIRS day 300:", [float(a[-1]) for a in sol.ys])
    sol = simulate(ode=seirs_ode_seasonal, duration_days=1500, initial_state=y0,
                   ode_parameters=get_seasonal_odeparams(config), solver_parameters=config.parameters.solver_params)
    print("seasonal SEIRS day 1500:", [float(a[-1]) for a in sol.ys])
