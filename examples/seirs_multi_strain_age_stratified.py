"""Multi-strain, age-stratified SEIRS on dynode_amd -- counterpart of the reference's
examples/seirs_multi_strain_age_stratified.py (any number of ages / strains compiled into the
library; the reference hard-codes 3 strains)."""

from datetime import date

import numpy as np

from dynode_amd import (Bin, Compartment, Dimension, Initializer, Params, SimulationConfig, SolverParams, Strain,
                        TransmissionParams, simulate)
from dynode_amd.rhs import SEIRS_MultiStrain_ODEParams, seirs_multi_strain_ode  # noqa: F401
from dynode_amd.utils import vectorize_objects


class SEIRSStratifiedInitializer(Initializer):
    def __init__(self, population_size=1000, age_demographics=(0.75, 0.25)):
        super().__init__(description="SEIRS initializer with age stratification", initialize_date=date(2022, 2, 11),
                         population_size=population_size)
        self._demo = np.asarray(age_demographics, dtype=float)

    def get_initial_state(self, config: SimulationConfig, s0_prop=0.99, i0_prop=0.01, **kwargs):
        demo = self._demo
        s_0 = self.population_size * s0_prop * demo
        e_0 = np.zeros(config.get_compartment("e").shape)
        dominance = np.array(vectorize_objects(config.parameters.transmission_params.strains, "r0"), dtype=float)
        dominance = dominance / dominance.sum()                      # infections split in proportion to r0
        i_0 = self.population_size * i0_prop * demo[:, None] * dominance
        r_0 = np.zeros(config.get_compartment("r").shape)
        c_0 = np.zeros(config.get_compartment("c").shape)
        return (s_0, e_0, i_0, r_0, c_0)


def get_config(r0s=(2.0, 2.5, 1.8), infectious_periods=(7.0, 6.0, 8.0), latent_periods=(3.0, 2.5, 4.0),
               waning_periods=(60.0, 80.0, 50.0), contact_matrix=((0.7, 0.3), (0.3, 0.7)),
               age_names=("young", "old"), age_demographics=(0.75, 0.25)) -> SimulationConfig:
    names = [chr(ord("A") + k) for k in range(len(r0s))]
    strains = [Strain(strain_name=n, r0=r0s[k], infectious_period=infectious_periods[k],
                      exposed_to_infectious=latent_periods[k]) for k, n in enumerate(names)]
    age = Dimension(name="age", bins=[Bin(name=a) for a in age_names])
    strain_dim = Dimension(name="strain", bins=[Bin(name=n) for n in names])
    comps = [Compartment(name="s", dimensions=[age])] + [Compartment(name=c, dimensions=[age, strain_dim])
                                                          for c in ("e", "i", "r", "c")]
    tp = TransmissionParams(strains=strains, strain_interactions={a: {b: 1.0 for b in names} for a in names},
                            contact_matrix=np.asarray(contact_matrix, dtype=float), waning_period=tuple(waning_periods))
    return SimulationConfig(compartments=comps, initializer=SEIRSStratifiedInitializer(age_demographics=age_demographics),
                            parameters=Params(solver_params=SolverParams(), transmission_params=tp))


def get_odeparams(config: SimulationConfig) -> SEIRS_MultiStrain_ODEParams:
    tp = config.parameters.transmission_params
    r0s = np.array(vectorize_objects(tp.strains, "r0"), dtype=float)
    t_inf = np.array(vectorize_objects(tp.strains, "infectious_period"), dtype=float)
    t_lat = np.array(vectorize_objects(tp.strains, "exposed_to_infectious"), dtype=float)
    return SEIRS_MultiStrain_ODEParams(beta=r0s / t_inf, gamma=1.0 / t_inf, sigma=1.0 / t_lat,
                                       omega=1.0 / np.array(tp.waning_period, dtype=float),
                                       contact_matrix=tp.contact_matrix, idx=config.idx)


if __name__ == "__main__":
    config = get_config(r0s=[2.4, 2.5, 2.8], infectious_periods=[7.0] * 3, latent_periods=[3.0] * 3,
                        waning_periods=[60.0] * 3)
    sol = simulate(ode=seirs_multi_strain_ode, duration_days=500, initial_state=config.initializer.get_initial_state(config),
                   ode_parameters=get_odeparams(config), solver_parameters=config.parameters.solver_params)
    c = sol.ys[config.idx.c].cpu().numpy()  # (501, 2, 3)
    print("cumulative incidence by strain at day 500:", c[-1].sum(axis=0))
