"""When did the new strain arrive, and how many visitors brought it?

NUTS on the introduced-strain model of examples/seirs_introduced_strain.py: the newcomer's
``introduction_time`` and ``introduction_percentage`` carry priors (as ``Strain`` fields, the way the
reference's configs attach priors to r0 in examples/sir_infer_parameters.py:47-58), the likelihood is
Poisson on the newcomer's daily incidence (increments of its cumulative-infection compartment).
The ODE has 14 parameters but only 2 are sampled: the gradient-solve seeds its tangents along the 2
latent coordinates, so one fused launch per gradient is enough.
"""

import torch

from dynode_amd import PoissonObservation, SimulationConfig, simulate
from dynode_amd.infer import distributions as dist
from dynode_amd.infer import handlers, sample_then_resolve
from dynode_amd.infer.inference import MCMCProcess
from dynode_amd.rhs import seirs_multi_strain_ode
from examples import seirs_introduced_strain as base

TRUTH = dict(introduction_time=60.0, introduction_percentage=0.005)


def get_config() -> SimulationConfig:
    """The static config with priors on the newcomer's arrival day and size."""
    config = base.get_config(**TRUTH)
    newcomer = config.parameters.transmission_params.strains[1]
    newcomer.introduction_time = dist.Uniform(20.0, 120.0)
    newcomer.introduction_percentage = dist.TransformedDistribution(dist.Beta(2.0, 2.0), dist.transforms.AffineTransform(0.0, 0.02))
    return config


def _solve(config: SimulationConfig, tf, observe=None):
    return simulate(ode=seirs_multi_strain_ode, duration_days=tf, initial_state=config.initializer.get_initial_state(config),
                    ode_parameters=base.get_odeparams(config), solver_parameters=config.parameters.solver_params,
                    observe=observe)


def model(config: SimulationConfig, tf, obs_data):
    """Sample the priors into the config, solve, score the newcomer's incidence (likelihood fused into
    the solve kernel: only its value and gradient leave the GPU registers)."""
    config = config.model_copy(deep=False)
    config.parameters = config.parameters.model_copy(deep=False)
    config.parameters.transmission_params = sample_then_resolve(config.parameters.transmission_params)
    sol = _solve(config, tf, observe=PoissonObservation(compartment=config.idx.c, data=obs_data, increments=True, floor=1e-6))
    handlers.factor("incidence", sol.log_likelihood)
    return sol


def synthetic_incidence(tf=150):
    """Noiseless daily incidence (both strains, all ages) of the run at TRUTH."""
    sol = _solve(base.get_config(**TRUTH), tf)
    return torch.diff(sol.ys[base.get_config().idx.c], dim=0).cpu()          # (tf, ages, strains)


if __name__ == "__main__":
    data = synthetic_incidence(150)
    process = MCMCProcess(numpyro_model=model, num_warmup=300, num_samples=300, num_chains=32, nuts_max_tree_depth=8)
    mcmc = process.infer(config=get_config(), tf=150, obs_data=data)
    mcmc.print_summary()
    print("truth:", TRUTH)
