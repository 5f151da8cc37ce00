"""MCMCProcess: DynODE's inference front-end on the batched GPU NUTS.

Mirror of /root/reference/src/dynode/infer/inference.py:29-241 for the MCMC path: same fields
(``numpyro_model, num_samples, num_warmup, num_chains, nuts_max_tree_depth, nuts_init_strategy,
progress_bar``), ``infer(**kwargs)`` forwards the kwargs to the model, ``get_samples`` returns a
dict keyed by the site names of ``sample_then_resolve`` with shape ``(chains*samples,)`` or
``(chains, samples)``.  The reference's seed (``PRNGKey(8675314)``, :45) is kept as the default.

The model is an ordinary Python function written with ``dynode_amd.infer.handlers.sample`` /
``simulate`` (see examples/sir_infer_parameters.py).  All chains are evaluated together: latent
sites hold a ``[chains]`` tensor, so the model must reduce with NEGATIVE axes (time is axis
``-1 - compartment.ndim``) -- the one difference from a numpyro model, which sees one chain at a
time under vmap/pmap.
"""

from __future__ import annotations

import math
from collections import OrderedDict
from typing import Any, Callable, Optional

import torch
from pydantic import BaseModel, ConfigDict, Field, PositiveInt, PrivateAttr

from .. import sharding
from . import handlers
from .distributions import biject_to
from .nuts import BatchedNUTS, NUTSResult


def init_to_median(num_samples: int = 15):
    """numpyro's default NUTS initialisation: per chain, the median of a few prior draws."""
    return ("median", int(num_samples))


def init_to_sample():
    return ("sample", 1)


class Potential:
    """Negative log joint of a handlers-style model over the unconstrained latent space, batched
    over chains, with gradients from torch autograd (the solve contributes through
    ``autodiff._DiffSolve``)."""

    def __init__(self, model: Callable, model_kwargs: dict, seed: int, device):
        self.model, self.kwargs, self.device = model, model_kwargs, device
        with handlers.seed(seed), handlers.trace() as tr:
            model(**model_kwargs)
        self.latent = OrderedDict((n, s["fn"]) for n, s in tr.sites.items()
                                  if s["type"] == "sample" and not s["is_observed"])
        if not self.latent:
            raise ValueError("the model has no latent sample sites")
        for n, s in tr.sites.items():
            if s["type"] == "sample" and not s["is_observed"] and tuple(s["value"].shape) not in ((), (1,)):
                raise ValueError(f"latent site {n!r} is not scalar; vector-valued latents are not supported yet")
        self.deterministic = [n for n, s in tr.sites.items() if s["type"] == "deterministic"]
        self.bij = OrderedDict((n, biject_to(d.support)) for n, d in self.latent.items())
        self.dim = len(self.latent)

    def constrain(self, z: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((n, b(z[..., i])) for i, (n, b) in enumerate(self.bij.items()))

    def initial(self, chains: int, strategy, seed: int) -> torch.Tensor:
        gen = torch.Generator().manual_seed(seed)
        kind, n = strategy if isinstance(strategy, tuple) else strategy()
        cols = []
        for name, d in self.latent.items():
            draws = d.sample(gen, (chains, n)).reshape(chains, n)
            x = draws.median(dim=1).values if kind == "median" else draws[:, 0]
            cols.append(self.bij[name].inv(x))
        return torch.stack(cols, dim=1).to(self.device)

    def log_joint(self, z: torch.Tensor):
        """(log p(x, obs) + log|dx/dz|) per chain, and the trace."""
        x = self.constrain(z)
        with handlers.substitute(x), handlers.trace() as tr:
            self.model(**self.kwargs)
        C = z.shape[0]
        total = torch.zeros(C, dtype=torch.float64, device=z.device)
        for i, (name, b) in enumerate(self.bij.items()):
            total = total + self.latent[name].log_prob(x[name]) + b.log_abs_det_jacobian(z[:, i])
        for name, s in tr.sites.items():
            if s["type"] == "sample" and s["is_observed"]:
                lp = s["fn"].log_prob(s["value"].to(z.device))
                total = total + lp.reshape(C, -1).sum(-1)
        return total, tr

    def potential_and_grad(self, z: torch.Tensor):
        z = z.detach().requires_grad_(True)
        lj, _ = self.log_joint(z)
        (g,) = torch.autograd.grad(lj.sum(), z)
        return -lj.detach(), -g


class MCMCResult:
    """What ``infer`` returns: the sampler output plus numpyro-style accessors."""

    def __init__(self, potential: Potential, nuts: NUTSResult, num_chains: int):
        self.potential, self.nuts, self.num_chains = potential, nuts, num_chains

    def get_samples(self, group_by_chain: bool = False) -> dict:
        x = self.potential.constrain(self.nuts.samples)            # [C, N] per site
        return {n: (v if group_by_chain else v.reshape(-1)) for n, v in x.items()}

    @property
    def last_state(self):
        return self.nuts.samples[:, -1]

    def print_summary(self) -> str:
        lines = [f"{'site':32s} {'mean':>10s} {'std':>10s} {'5%':>10s} {'95%':>10s}"]
        for n, v in self.get_samples().items():
            q = torch.quantile(v, torch.tensor([0.05, 0.95], dtype=v.dtype, device=v.device))
            lines.append(f"{n:32s} {float(v.mean()):10.4f} {float(v.std()):10.4f} {float(q[0]):10.4f} {float(q[1]):10.4f}")
        lines.append(f"divergences: {int(self.nuts.diverging.sum())}, mean accept prob: {float(self.nuts.accept_prob.mean()):.3f}, "
                     f"mean leapfrogs/transition: {float(self.nuts.num_steps.double().mean()):.2f}")
        text = "\n".join(lines)
        print(text)
        return text


class InferenceProcess(BaseModel):
    """Abstract inference process (reference inference.py:29-117)."""

    model_config = ConfigDict(arbitrary_types_allowed=True)
    numpyro_model: Callable = Field(description="model(**kwargs): samples parameters, simulates, scores observations")
    inference_prngkey: int = 8675314
    _inference_complete: bool = PrivateAttr(default=False)
    _inferer: Optional[Any] = PrivateAttr(default=None)
    _inference_state: Optional[Any] = PrivateAttr(default=None)
    _inferer_kwargs: Optional[dict] = PrivateAttr(default_factory=dict)

    def infer(self, **kwargs):
        raise NotImplementedError("Inference process not implemented, please use a subclass.")

    def get_samples(self, group_by_chain=False, exclude_deterministic=True) -> dict:
        raise NotImplementedError("get_samples() process not implemented, please use a subclass.")


class MCMCProcess(InferenceProcess):
    """Fit the model with NUTS (dense mass, init-to-median), all local chains on this GPU.

    Under ``torch.distributed`` (one process per GPU) every rank runs ``num_chains / world``
    chains with a rank-offset seed; ``get_samples(gather=True)`` collects them on rank 0 over
    RCCL -- the only collective of the inference path.
    """

    num_samples: PositiveInt
    num_warmup: PositiveInt
    num_chains: PositiveInt
    nuts_max_tree_depth: PositiveInt
    nuts_init_strategy: Callable = init_to_median
    mcmc_kwargs: dict = Field(default_factory=dict)
    nuts_kwargs: dict = Field(default_factory=dict)
    progress_bar: bool = True

    def infer(self, **kwargs) -> MCMCResult:
        from ..engine import require_gpu

        device = require_gpu()
        rank, world = sharding.world()
        lo, hi = sharding.shard_bounds(self.num_chains, rank, world)
        local = hi - lo
        pot = Potential(self.numpyro_model, kwargs, self.inference_prngkey, device)
        z0 = pot.initial(self.num_chains, self.nuts_init_strategy, self.inference_prngkey)[lo:hi]
        sampler = BatchedNUTS(pot.potential_and_grad, max_tree_depth=self.nuts_max_tree_depth,
                              target_accept=self.nuts_kwargs.get("target_accept_prob", 0.8),
                              seed=self.inference_prngkey + 7919 * rank)
        total = self.num_warmup + self.num_samples

        def progress(it, warm):
            if self.progress_bar and rank == 0 and (it + 1) % max(1, total // 10) == 0:
                print(f"[nuts] {'warmup' if warm else 'sample'} {it + 1}/{total} ({local} chains on this GPU)", flush=True)

        res = sampler.run(z0, self.num_warmup, self.num_samples,
                          init_step_size=self.nuts_kwargs.get("step_size", 1.0), progress=progress)
        out = MCMCResult(pot, res, local)
        self._inference_complete, self._inferer, self._inference_state = True, out, out.last_state
        self._inferer_kwargs = kwargs
        return out

    def get_samples(self, group_by_chain=False, exclude_deterministic=True, gather: bool = False) -> dict:
        if not self._inference_complete:
            raise AssertionError("Inference process not completed, please call infer() first.")
        samples = self._inferer.get_samples(group_by_chain=True)
        if not exclude_deterministic:
            samples.update(self._deterministic_sites())
        if gather:
            gathered = {n: sharding.gather_rows(v.contiguous(), self.num_chains) for n, v in samples.items()}
            if sharding.world()[0] != 0:
                return {}
            samples = gathered
        return {n: (v if group_by_chain else v.reshape((-1,) + tuple(v.shape[2:]))) for n, v in samples.items()}

    def _deterministic_sites(self) -> dict:
        """Replay the model over the posterior draws to collect ``deterministic`` sites."""
        pot, res = self._inferer.potential, self._inferer.nuts
        C, N, D = res.samples.shape
        with torch.no_grad():
            _, tr = pot.log_joint(res.samples.reshape(C * N, D))
        out = {}
        for n, s in tr.sites.items():
            if s["type"] == "deterministic" and isinstance(s["value"], torch.Tensor):
                v = s["value"]
                out[n] = v.reshape((C, N) + tuple(v.shape[1:])) if v.dim() and v.shape[0] == C * N else v
        return out


def log_posterior_grid(potential: Potential, grids: list) -> torch.Tensor:
    """Log joint on a tensor grid of CONSTRAINED values (for quadrature checks of 2-D posteriors)."""
    mesh = torch.meshgrid(*grids, indexing="ij")
    x = torch.stack([m.reshape(-1) for m in mesh], dim=1).to(potential.device)
    z = torch.stack([b.inv(x[:, i]) for i, b in enumerate(potential.bij.values())], dim=1)
    with torch.no_grad():
        lj, _ = potential.log_joint(z)
        for i, b in enumerate(potential.bij.values()):
            lj = lj - b.log_abs_det_jacobian(z[:, i])      # density in x, not in z
    return lj.reshape(mesh[0].shape)


__all__ = ["InferenceProcess", "MCMCProcess", "MCMCResult", "Potential", "init_to_median", "init_to_sample",
           "log_posterior_grid", "math"]
