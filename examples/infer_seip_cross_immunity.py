"""How much does a past infection with the first strain protect against the second?  Inference on the SEIP model.

Latent: the cross-immunity ``strain_interactions["beta"]["alpha"]`` and the second strain's R0; data: weekly
infections by strain and immune history from the run of examples/seip_immune_history.py at TRUTH.

The model is written with torch ops from the latent sites to the ODE parameters (the susceptibility table through
``protection_table_torch``), so NUTS differentiates it like any other: the SEIP kernels have no tangent planes, the
gradient-solve is the primal solve plus ONE batched launch of perturbed rows that replay the primal's accepted
steps (``engine._replayed_tangents``; adaptive step-size control stays on).  ``--ensemble`` runs the gradient-free
affine-invariant ensemble sampler instead (``mcmc_kwargs={"sampler": "ensemble"}``, dynode_amd/infer/ensemble.py).
"""

import numpy as np
import torch

from dynode_amd import SimulationConfig, simulate
from dynode_amd.infer import distributions as dist
from dynode_amd.infer import handlers
from dynode_amd.infer.inference import MCMCProcess
from dynode_amd.seip import protection_table, seip_ode
from examples import seip_immune_history as base

TRUTH = dict(cross_immunity=0.45, r0_beta=2.4)


def _odeparams(config: SimulationConfig, cross_immunity, r0_beta):
    """examples/seip_immune_history.get_odeparams with the two latent numbers (tensors: one entry per chain / walker;
    their autograd graph, if any, is kept)."""
    from dynode_amd.seip import protection_table_torch

    p = base.get_odeparams(config)
    tp = config.parameters.transmission_params
    names = [s.strain_name for s in tp.strains]
    f64 = torch.float64
    cross, r0b = (torch.atleast_1d(torch.as_tensor(v, dtype=f64)) for v in (cross_immunity, r0_beta))
    dev = cross.device
    n = max(cross.numel(), r0b.numel())
    chi0 = torch.tensor([[tp.strain_interactions[a][b] for b in names] for a in names], dtype=f64, device=dev)
    row0 = chi0[0].expand(n, 2)
    row1 = torch.stack([cross.expand(n), chi0[1, 1].expand(n)], dim=1)
    chi = torch.stack([row0, row1], dim=1)                                   # [n, 2, 2] with chi[:, 1, 0] = cross-immunity
    s_comp = config.get_compartment("s")
    ve = np.array([[s.vaccine_efficacy[k] for k in range(len(s_comp.dimensions[2]))] for s in tp.strains])
    p.susceptibility = protection_table_torch(chi, ve, [b.base_protection for b in s_comp.dimensions[3].bins],
                                              base.MIN_HOMOLOGOUS_IMMUNITY)
    beta0 = torch.as_tensor(np.asarray(p.beta, float), dtype=f64, device=dev)
    gamma = torch.as_tensor(np.asarray(p.gamma, float), dtype=f64, device=dev)
    p.beta = torch.stack([beta0[0].expand(n), r0b.to(dev).expand(n) * gamma[1]], dim=1)
    if n == 1 and not (p.beta.requires_grad or p.susceptibility.requires_grad):
        p.susceptibility, p.beta = p.susceptibility[0], p.beta[0]
    return p


def weekly_infections(config: SimulationConfig, tf, cross_immunity, r0_beta):
    """New infections per week, by immune history of the infected and by strain: (..., weeks, hist, strain)."""
    sol = simulate(ode=seip_ode, duration_days=tf, initial_state=config.initializer.get_initial_state(config),
                   ode_parameters=_odeparams(config, cross_immunity, r0_beta), solver_parameters=config.parameters.solver_params,
                   sub_save_indices=(config.idx.c,), save_step=7)
    c = sol.ys[config.idx.c]                                    # (weeks + 1, age, hist, vax, strain), walkers in front
    return torch.diff(c, dim=-5).sum(dim=(-4, -2))


def model(config: SimulationConfig, tf, obs_data):
    cross = handlers.sample("cross_immunity", dist.Beta(2.0, 2.0))
    r0b = handlers.sample("r0_beta", dist.Uniform(1.2, 4.0))
    rate = weekly_infections(config, tf, cross, r0b)
    handlers.sample("weekly_infections", dist.Poisson(torch.clamp(rate, min=1e-6)), obs=obs_data)


if __name__ == "__main__":
    import sys

    config = base.get_config()
    if "--ensemble" in sys.argv:
        kwargs = dict(num_warmup=300, num_samples=300, num_chains=64, nuts_max_tree_depth=10, mcmc_kwargs={"sampler": "ensemble"})
    else:   # NUTS, default machinery, adaptive solver steps
        kwargs = dict(num_warmup=200, num_samples=200, num_chains=32, nuts_max_tree_depth=6)
    data = weekly_infections(config, 210, **TRUTH).cpu()
    process = MCMCProcess(numpyro_model=model, **kwargs)
    mcmc = process.infer(config=config, tf=210, obs_data=data)
    mcmc.print_summary()
    print("truth:", TRUTH)
