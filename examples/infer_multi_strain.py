"""Which strain spreads how fast?  NUTS on the multi-strain, age-stratified SEIRS model.

The reference's multi-strain example (examples/seirs_multi_strain_age_stratified.py:46-49,187-209: 2 ages x 3 strains, every
strain with its own r0 / infectious period / latent period) with priors on the strains' parameters, the way the reference's
inference example attaches them to one strain (examples/sir_infer_parameters.py:47-58), and a Poisson likelihood on the
daily incidence by age and strain (increments of the cumulative-infection compartment ``c``), scored inside the solve kernel.

``get_config(sites=6)``: r0 and infectious period of the three strains (6 sampled dimensions: the sampler kernel's
per-dimension instances, folded potential: the gradient-solve, eight lane groups per trajectory, and the sampler kernel -- two
launches per iteration; `dyn_solver_opts::nuts_tail` would make it one at four lane groups, which measured slower).  ``sites=9`` adds the three latent periods: beyond
eight dimensions the sampler kernel's half-wave-per-chain form (``dyn_nuts_advance_mapped``, include/dynode_hip.h) behind the
same folded potential (up to sixteen sites).  The initial infections are split evenly over the strains here (the reference splits them in proportion
to r0, :153-167, which would make the initial state a function of the sampled values).
"""

import numpy as np
import torch

from dynode_amd import PoissonObservation, SimulationConfig, simulate
from dynode_amd.infer import distributions as dist
from dynode_amd.infer import handlers, sample_then_resolve
from dynode_amd.infer.inference import MCMCProcess
from dynode_amd.rhs import SEIRS_MultiStrain_ODEParams, seirs_multi_strain_ode
from examples import seirs_multi_strain_age_stratified as base

TRUTH = dict(r0s=(2.0, 2.5, 1.8), infectious_periods=(7.0, 6.0, 8.0), latent_periods=(3.0, 2.5, 4.0))


def get_config(sites: int = 6) -> SimulationConfig:
    """The static config with priors on every strain's r0 and infectious period (and, ``sites=9``, latent period)."""
    assert sites in (6, 9)
    config = base.get_config(**TRUTH)
    for k, strain in enumerate(config.parameters.transmission_params.strains):
        strain.r0 = dist.TransformedDistribution(dist.Beta(2.0, 2.0), dist.transforms.AffineTransform(1.2, 2.0))      # 1.2 .. 3.2
        strain.infectious_period = dist.TruncatedNormal(loc=7.0, scale=2.0, low=3.0, high=12.0)
        if sites == 9:
            strain.exposed_to_infectious = dist.Uniform(1.0, 6.0)
    return config


def initial_state(config: SimulationConfig, population=1000.0, i0_prop=0.01, demographics=(0.75, 0.25)):
    demo = np.asarray(demographics, dtype=float)
    n_strain = len(config.parameters.transmission_params.strains)
    shape = config.get_compartment("e").shape
    i_0 = population * i0_prop * demo[:, None] * np.full(n_strain, 1.0 / n_strain)
    return (population * (1.0 - i0_prop) * demo, np.zeros(shape), i_0, np.zeros(shape), np.zeros(shape))


def get_odeparams(config: SimulationConfig) -> SEIRS_MultiStrain_ODEParams:
    """reference :187-209 on values that may be [chains] tensors (the sampled sites)."""
    tp = config.parameters.transmission_params
    col = lambda name: torch.stack([torch.as_tensor(getattr(s, name), dtype=torch.float64) * torch.ones(()) for s in tp.strains], dim=-1)  # noqa: E731
    r0, t_inf, t_lat = col("r0"), col("infectious_period"), col("exposed_to_infectious")
    return SEIRS_MultiStrain_ODEParams(beta=r0 / t_inf, gamma=1.0 / t_inf, sigma=1.0 / t_lat, omega=1.0 / np.array(tp.waning_period, dtype=float),
                                       contact_matrix=tp.contact_matrix, idx=config.idx)


def _solve(config: SimulationConfig, tf, observe=None):
    return simulate(ode=seirs_multi_strain_ode, duration_days=tf, initial_state=initial_state(config), ode_parameters=get_odeparams(config),
                    solver_parameters=config.parameters.solver_params, observe=observe)


def model(config: SimulationConfig, tf, obs_data):
    config = config.model_copy(deep=False)
    config.parameters = config.parameters.model_copy(deep=False)
    config.parameters.transmission_params = sample_then_resolve(config.parameters.transmission_params)
    sol = _solve(config, tf, observe=PoissonObservation(compartment=config.idx.c, data=obs_data, increments=True, floor=1e-6))
    handlers.factor("incidence", sol.log_likelihood)
    return sol


def synthetic_incidence(tf=120):
    """Noiseless daily incidence by age and strain of the run at TRUTH."""
    config = base.get_config(**TRUTH)
    return torch.diff(_solve(config, tf).ys[config.idx.c], dim=0).cpu()          # (tf, ages, strains)


if __name__ == "__main__":
    data = synthetic_incidence(120)
    process = MCMCProcess(numpyro_model=model, num_warmup=300, num_samples=300, num_chains=32, nuts_max_tree_depth=8)
    mcmc = process.infer(config=get_config(6), tf=120, obs_data=data)
    mcmc.print_summary()
    print("truth:", TRUTH)
