"""How well does the vaccine protect?  NUTS on the vaccinated model of examples/seirs_vaccination.py.

Two latent quantities -- the efficacy of one dose against the first strain and the share of the remaining risk
a second dose removes -- enter the ODE through the susceptibility of the vaccination tiers
(``VaccinationParams.vaccine_efficacy``); the second strain escapes 40 % of the protection.  The likelihood is
Poisson on the daily infections by dose count (increments of the cumulative-infection compartment, summed
over ages and strains).  The gradient-solve seeds its tangents along the two latent coordinates.
"""

import torch

from dynode_amd import SimulationConfig, simulate
from dynode_amd.infer import distributions as dist
from dynode_amd.infer import handlers
from dynode_amd.infer.inference import MCMCProcess
from dynode_amd.rhs import SEIRS_MultiStrain_ODEParams, VaccinationParams, seirs_multi_strain_ode
from examples import seirs_vaccination as base

TRUTH = dict(efficacy_one_dose=0.45, second_dose_boost=0.5)
ESCAPE = 0.6           # the second strain sees 60 % of the protection


def efficacy_table(one_dose, boost):
    """[..., strains, doses] from the two latent numbers (tensors, possibly one row per chain)."""
    one_dose, boost = torch.as_tensor(one_dose, dtype=torch.float64), torch.as_tensor(boost, dtype=torch.float64)
    two = one_dose + (1.0 - one_dose) * boost
    first = torch.stack([torch.zeros_like(one_dose), one_dose, two], dim=-1)
    return torch.stack([first, ESCAPE * first], dim=-2)


def _solve(config: SimulationConfig, tf, ve):
    p = base.get_odeparams(config)
    vp = p.vaccination_params
    q = SEIRS_MultiStrain_ODEParams(beta=p.beta, gamma=p.gamma, sigma=p.sigma, omega=p.omega, contact_matrix=p.contact_matrix,
                                    vaccination_params=VaccinationParams(vp.knot_locations, vp.base_equations,
                                                                         vp.knot_coefficients, ve))
    return simulate(ode=seirs_multi_strain_ode, duration_days=tf, initial_state=config.initializer.get_initial_state(config),
                    ode_parameters=q, solver_parameters=config.parameters.solver_params)


def model(config: SimulationConfig, tf, obs_data):
    one = handlers.sample("efficacy_one_dose", dist.Beta(2.0, 2.0))
    boost = handlers.sample("second_dose_boost", dist.Beta(2.0, 2.0))
    sol = _solve(config, tf, efficacy_table(one, boost))
    c = sol.ys[config.idx.c]                                   # (tf + 1, age, doses, strain), chains in front when batched
    incidence = torch.diff(c, dim=-4).sum(dim=(-3, -1))        # new infections per day and dose count
    handlers.sample("infections_by_doses", dist.Poisson(torch.clamp(incidence, min=1e-6)), obs=obs_data)
    return sol


def synthetic_incidence(config, tf=200):
    sol = _solve(config, tf, efficacy_table(TRUTH["efficacy_one_dose"], TRUTH["second_dose_boost"]))
    return torch.diff(sol.ys[config.idx.c], dim=0).sum(dim=(1, 3)).cpu()      # (tf, doses)


if __name__ == "__main__":
    config = base.get_config()
    data = synthetic_incidence(config, 200)
    process = MCMCProcess(numpyro_model=model, num_warmup=200, num_samples=200, num_chains=32, nuts_max_tree_depth=8)
    mcmc = process.infer(config=config, tf=200, obs_data=data)
    mcmc.print_summary()
    print("truth:", TRUTH)
