"""The SEIP model of ode_model.md: age x immune history x vaccination count x waning state, two strains.

The reference ships the configuration classes for its production model (immune-history, vaccination and
waning dimensions, ``Strain.vaccine_efficacy``, ``strain_interactions``) and describes the equations in
ode_model.md, but has no example that runs it.  Here the same configuration objects drive the SEIP kernel:

    s[age, hist, vax, wane]     e, i, c[age, hist, vax, strain]

Recovered people return to ``s`` with the strain added to their immune history, fully protected at first;
protection wanes through the ``WaneBin`` chain; vaccine doses arrive at spline rates; the second strain's
partial escape from the first one's immunity (``strain_interactions``) lets it re-infect.
"""

import math
from datetime import date

import numpy as np

from dynode_amd import (AgeBin, Bin, Compartment, Dimension, FullStratifiedImmuneHistoryDimension, Initializer, Params,
                        SimulationConfig, SolverParams, Strain, TransmissionParams, VaccinationDimension, WaneDimension, simulate)
from dynode_amd.rhs import VaccinationParams
from dynode_amd.seip import SEIP_ODEParams, history_masks, params_from_config, seip_ode

AGES = [AgeBin(min_value=0, max_value=17), AgeBin(min_value=18, max_value=64), AgeBin(min_value=65, max_value=99)]
DEMOGRAPHICS = np.array([0.22, 0.61, 0.17])
MIN_HOMOLOGOUS_IMMUNITY = 0.1


class NaiveInitializer(Initializer):
    """Nobody has met either strain or a vaccine; 0.1 % of every age group is infectious with the first strain
    and 0.01 % with the second."""

    def __init__(self, population_size=100_000):
        super().__init__(description="immunologically naive", initialize_date=date(2022, 2, 11), population_size=population_size)

    def get_initial_state(self, config: SimulationConfig, i0_prop=(0.001, 0.0001), **kwargs):
        pop = self.population_size * DEMOGRAPHICS
        s = np.zeros(config.get_compartment("s").shape)             # (age, hist, vax, wane)
        i = np.zeros(config.get_compartment("i").shape)             # (age, hist, vax, strain)
        i[:, 0, 0, :] = pop[:, None] * np.asarray(i0_prop)[None, :]
        s[:, 0, 0, -1] = pop - i[:, 0, 0].sum(-1)                   # never protected: the last waning state
        return (s, np.zeros_like(i), i, np.zeros_like(i))


def get_config() -> SimulationConfig:
    strains = [Strain(strain_name="alpha", r0=1.8, infectious_period=7.0, exposed_to_infectious=3.0,
                      vaccine_efficacy={0: 0.0, 1: 0.35, 2: 0.6}),
               Strain(strain_name="beta", r0=2.4, infectious_period=6.0, exposed_to_infectious=2.5,
                      vaccine_efficacy={0: 0.0, 1: 0.2, 2: 0.4})]
    names = [s.strain_name for s in strains]
    age = Dimension(name="age", bins=AGES)
    hist = FullStratifiedImmuneHistoryDimension(strains)
    vax = VaccinationDimension(max_ordinal_vaccinations=2)
    wane = WaneDimension(waiting_times=[60.0, 60.0, 90.0, math.inf], base_protections=[1.0, 0.7, 0.4, 0.0])
    strain_dim = Dimension(name="strain", bins=[Bin(name=n) for n in names])
    comps = [Compartment(name="s", dimensions=[age, hist, vax, wane])] + [
        Compartment(name=c, dimensions=[age, hist, vax, strain_dim]) for c in ("e", "i", "c")]
    contact = np.array([[0.60, 0.35, 0.05], [0.20, 0.65, 0.15], [0.10, 0.45, 0.45]])
    contact = contact / np.max(np.real(np.linalg.eigvals(contact)))
    # strain_interactions[challenger][past strain]: protection a past infection gives against the challenger
    interactions = {"alpha": {"alpha": 1.0, "beta": 0.8}, "beta": {"alpha": 0.45, "beta": 1.0}}
    tp = TransmissionParams(strains=strains, strain_interactions=interactions, contact_matrix=contact)
    return SimulationConfig(compartments=comps, initializer=NaiveInitializer(),
                            parameters=Params(solver_params=SolverParams(), transmission_params=tp))


def dose_splines(n_ages, n_tiers, start=60.0, ramp=20.0, rates=(0.001, 0.004, 0.008), gap=28.0):
    """Doses per person and day (utils.evaluate_cubic_spline's arguments): smooth step from 0 to the age's rate
    over ``ramp`` days from ``start``; the next dose ``gap`` days later; the top tier is not boosted."""
    knots, coefs = np.zeros((n_ages, n_tiers, 3)), np.zeros((n_ages, n_tiers, 3))
    for a in range(n_ages):
        for k in range(n_tiers - 1):
            t0 = start + k * gap
            c = rates[a] / (0.75 * ramp**3)
            knots[a, k], coefs[a, k] = [t0, t0 + ramp / 2, t0 + ramp], [c, -2 * c, c]
    return VaccinationParams(knot_locations=knots, base_equations=np.zeros((n_ages, n_tiers, 4)), knot_coefficients=coefs,
                             vaccine_efficacy=None)


def get_odeparams(config: SimulationConfig) -> SEIP_ODEParams:
    """Object -> vector flattening, the SEIP counterpart of the reference examples' ``get_odeparams``: rates from the
    strains, waning rates and protections from the ``WaneBin`` chain, the susceptibility table from
    ``strain_interactions`` and ``Strain.vaccine_efficacy`` (``dynode_amd.seip.params_from_config`` does the mapping)."""
    n_tiers = len(config.get_compartment("s").dimensions[2])
    return params_from_config(config, vaccination_params=dose_splines(len(AGES), n_tiers),
                              min_homologous_immunity=MIN_HOMOLOGOUS_IMMUNITY)


def run_simulation(config: SimulationConfig, tf=365):
    return simulate(ode=seip_ode, duration_days=tf, initial_state=config.initializer.get_initial_state(config),
                    ode_parameters=get_odeparams(config), solver_parameters=config.parameters.solver_params)


if __name__ == "__main__":
    config = get_config()
    sol = run_simulation(config)
    idx = config.idx
    s = sol.ys[idx.s].cpu().numpy()                       # (366, age, hist, vax, wane)
    c = sol.ys[idx.c].cpu().numpy()                       # (366, age, hist, vax, strain)
    hist_names = [b.name for b in config.get_compartment("s").dimensions[1].bins]
    print("immune-history bins:", hist_names, "= bit sets", list(history_masks(2)))
    for day in (0, 60, 120, 200, 365):
        by_hist = s[day].sum((0, 2, 3)).round(0)
        by_vax = s[day].sum((0, 1, 3)).round(0)
        print(f"day {day:3d}  susceptible by history {by_hist}  by doses {by_vax}  cumulative infections by strain {c[day].sum((0, 1, 2)).round(0)}")
    reinf = c[-1][:, 1:].sum() / c[-1].sum()
    print(f"share of infections in people with a previous infection: {reinf:.3f}")
