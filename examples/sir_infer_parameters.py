"""Infer r0 and the infectious period of the age-stratified SIR model with NUTS, on dynode_amd --
counterpart of the reference's examples/sir_infer_parameters.py.

Differences from the reference script: ``handlers.sample`` / ``distributions`` come from
``dynode_amd.infer`` (numpyro is not needed), and ``model`` reduces over the time axis with a
negative index so that the same function scores one chain or all chains at once.
"""

import numpy as np
import torch

from dynode_amd import SimulationConfig, Strain
from dynode_amd.infer import distributions as dist
from dynode_amd.infer import handlers
from dynode_amd.infer.inference import MCMCProcess
from examples.sir_age_stratified import get_config as get_static_config
from examples.sir_age_stratified import run_simulation


def model(config: SimulationConfig, tf, obs_data):
    """Simulate, turn recovered counts into incidence, score it with a Poisson likelihood
    (reference examples/sir_infer_parameters.py:21-39)."""
    solution = run_simulation(config, tf)
    r = solution.ys[config.idx.r]                      # (tf + 1, age) or (chains, tf + 1, age)
    incidence = torch.diff(r, dim=-2)                  # time axis, counted from the right
    incidence = torch.clamp(incidence, min=1e-6)
    handlers.sample("inf_incidence", dist.Poisson(incidence), obs=obs_data)
    return solution


def model_fused(config: SimulationConfig, tf, obs_data):
    """The same model with the likelihood evaluated inside the solve kernel: the trajectory never
    leaves the registers, the solve returns the Poisson log-likelihood of ``diff(R)`` and its
    gradient (``dyn_solve_batch_loglik``).  Same posterior as :func:`model`."""
    from dynode_amd import PoissonObservation

    solution = run_simulation(config, tf, observe=PoissonObservation(compartment=config.idx.r, data=obs_data,
                                                                     increments=True, floor=1e-6))
    handlers.factor("inf_incidence", solution.log_likelihood)
    return solution


def get_config() -> SimulationConfig:
    """Static SIR config with the strain's r0 / infectious period replaced by priors (:42-59)."""
    sir_config = get_static_config(r_0=2.0, infectious_period=7.0)
    sir_config.parameters.transmission_params.strains = [
        Strain(strain_name="swo9",
               r0=dist.TransformedDistribution(dist.Beta(0.5, 0.5), dist.transforms.AffineTransform(1.5, 1)),
               infectious_period=dist.TruncatedNormal(loc=8, scale=2, low=2, high=15))
    ]
    return sir_config


def synthetic_incidence(tf=100):
    """Noiseless diff(R) of the static run r0 = 2, T_inf = 7 (:62-86)."""
    solution = run_simulation(get_static_config(), tf=tf)
    return torch.diff(solution.ys[get_static_config().idx.r], dim=0).cpu()


if __name__ == "__main__":
    incidence = synthetic_incidence(100)
    process = MCMCProcess(numpyro_model=model, num_warmup=500, num_samples=100, num_chains=4, nuts_max_tree_depth=10)
    mcmc = process.infer(config=get_config(), tf=100, obs_data=incidence)
    mcmc.print_summary()
    post = process.get_samples()
    print({k: (float(v.mean()), float(v.std())) for k, v in post.items()})
