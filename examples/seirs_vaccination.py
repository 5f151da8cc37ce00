"""Vaccination: a two-strain, age-stratified SEIRS whose compartments carry a dose-count axis.

The reference describes vaccination in its configuration (``VaccinationDimension``,
``Strain.vaccine_efficacy``), in ``utils.evaluate_cubic_spline`` (the time-dependent dose rates) and in
ode_model.md, but ships no example with it.  Here the same pieces drive the kernel's vaccination tiers:
susceptibles of age ``a`` with ``k`` doses receive ``nu_{a,k}(t) * P_a`` doses per day and move to
``k + 1``; a dose count's susceptibility to strain ``l`` is ``1 - vaccine_efficacy_l[k]``.
"""

from datetime import date

import numpy as np

from dynode_amd import (AgeBin, Bin, Compartment, Dimension, Initializer, Params, SimulationConfig, SolverParams, Strain,
                        TransmissionParams, VaccinationDimension, simulate)
from dynode_amd.rhs import SEIRS_MultiStrain_ODEParams, VaccinationParams, seirs_multi_strain_ode

AGES = [AgeBin(min_value=0, max_value=17), AgeBin(min_value=18, max_value=64), AgeBin(min_value=65, max_value=99)]


class UnvaccinatedInitializer(Initializer):
    """Everyone starts without doses; 0.05 % of every age group is infectious, split evenly over the strains."""

    def __init__(self, population_size=100_000, age_demographics=(0.22, 0.61, 0.17)):
        super().__init__(description="nobody vaccinated yet", initialize_date=date(2022, 2, 11), population_size=population_size)
        self._demo = np.asarray(age_demographics, dtype=float)

    def get_initial_state(self, config: SimulationConfig, i0_prop=0.0005, **kwargs):
        pop = self.population_size * self._demo
        s = np.zeros(config.get_compartment("s").shape)            # (age, doses)
        i = np.zeros(config.get_compartment("i").shape)            # (age, doses, strain)
        i[:, 0, :] = i0_prop * pop[:, None] / i.shape[-1]
        s[:, 0] = pop - i[:, 0].sum(-1)
        zeros = np.zeros_like(i)
        return (s, zeros.copy(), i, zeros.copy(), zeros.copy())


def get_config(max_doses=2, efficacy=({0: 0.0, 1: 0.45, 2: 0.7}, {0: 0.0, 1: 0.25, 2: 0.5})) -> SimulationConfig:
    strains = [Strain(strain_name="alpha", r0=1.5, infectious_period=7.0, exposed_to_infectious=3.0, vaccine_efficacy=efficacy[0]),
               Strain(strain_name="beta", r0=1.8, infectious_period=6.0, exposed_to_infectious=2.5, vaccine_efficacy=efficacy[1])]
    names = [s.strain_name for s in strains]
    age = Dimension(name="age", bins=AGES)
    vax = VaccinationDimension(max_ordinal_vaccinations=max_doses)
    strain_dim = Dimension(name="strain", bins=[Bin(name=n) for n in names])
    comps = [Compartment(name="s", dimensions=[age, vax])] + [Compartment(name=c, dimensions=[age, vax, strain_dim])
                                                               for c in ("e", "i", "r", "c")]
    contact = np.array([[0.60, 0.35, 0.05], [0.20, 0.65, 0.15], [0.10, 0.45, 0.45]])
    contact = contact / np.max(np.real(np.linalg.eigvals(contact)))
    tp = TransmissionParams(strains=strains, strain_interactions={a: {b: 1.0 for b in names} for a in names},
                            contact_matrix=contact, waning_period=(150.0, 150.0))
    return SimulationConfig(compartments=comps, initializer=UnvaccinatedInitializer(),
                            parameters=Params(solver_params=SolverParams(), transmission_params=tp))


def vaccination_splines(n_ages=3, n_tiers=3, campaign_start=30.0, first_dose_rate=(0.002, 0.006, 0.012),
                        second_dose_delay=28.0):
    """Dose rates per person and day as cubic splines (utils.evaluate_cubic_spline's arguments): zero until
    the campaign starts, then rising smoothly to a plateau; second doses follow four weeks later; the
    oldest get theirs fastest.  base = 0, two knots each: +c (t - t0)^3 from t0, -c (t - t1)^3 ... ."""
    ramp = 20.0                                                     # days from start to plateau
    knots = np.zeros((n_ages, n_tiers, 3))
    coefs = np.zeros((n_ages, n_tiers, 3))
    for a in range(n_ages):
        for k in range(n_tiers - 1):                                # the last tier has nowhere to go
            t0 = campaign_start + k * second_dose_delay
            plateau = first_dose_rate[a]
            # smooth step: c[(t - t0)^3 - 2 (t - t0 - ramp/2)^3 + (t - t0 - ramp)^3] = plateau after t0 + ramp
            c = plateau / (0.75 * ramp**3)
            knots[a, k] = [t0, t0 + ramp / 2, t0 + ramp]
            coefs[a, k] = [c, -2 * c, c]
    return knots, np.zeros((n_ages, n_tiers, 4)), coefs


def get_odeparams(config: SimulationConfig) -> SEIRS_MultiStrain_ODEParams:
    tp = config.parameters.transmission_params
    r0 = np.array([s.r0 for s in tp.strains], dtype=float)
    t_inf = np.array([s.infectious_period for s in tp.strains], dtype=float)
    t_lat = np.array([s.exposed_to_infectious for s in tp.strains], dtype=float)
    n_tiers = len(config.get_compartment("s").dimensions[1])
    efficacy = np.array([[s.vaccine_efficacy[k] for k in range(n_tiers)] for s in tp.strains])
    knots, base, coefs = vaccination_splines(len(AGES), n_tiers)
    return SEIRS_MultiStrain_ODEParams(
        beta=r0 / t_inf, gamma=1.0 / t_inf, sigma=1.0 / t_lat, omega=1.0 / np.array(tp.waning_period, dtype=float),
        contact_matrix=tp.contact_matrix, idx=config.idx,
        vaccination_params=VaccinationParams(knot_locations=knots, base_equations=base, knot_coefficients=coefs,
                                             vaccine_efficacy=efficacy))


def run_simulation(config: SimulationConfig, tf=300):
    return simulate(ode=seirs_multi_strain_ode, duration_days=tf, initial_state=config.initializer.get_initial_state(config),
                    ode_parameters=get_odeparams(config), solver_parameters=config.parameters.solver_params)


if __name__ == "__main__":
    config = get_config()
    sol = run_simulation(config)
    s = sol.ys[config.idx.s].cpu().numpy()               # (301, ages, doses)
    c = sol.ys[config.idx.c].cpu().numpy()               # (301, ages, doses, strains)
    for day in (0, 30, 60, 100, 200, 300):
        print(f"day {day:3d}  susceptible by doses {s[day].sum(0).round(0)}   cumulative infections by doses {c[day].sum((0, 2)).round(0)}")
