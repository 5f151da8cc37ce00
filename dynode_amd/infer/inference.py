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
import sys
from collections import OrderedDict
from typing import Any, Callable, Optional

import numpy as np
import torch
from pydantic import BaseModel, ConfigDict, Field, PositiveInt, PrivateAttr

from .. import sharding
from . import handlers
from .distributions import biject_to
from .nuts import BatchedNUTS, GraphNUTS, KernelNUTS, NUTSResult


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
        # observed data etc. live on the device from the start: no host-to-device copy per evaluation
        model_kwargs = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in model_kwargs.items()}
        self.model, self.kwargs, self.device = model, model_kwargs, device
        with handlers.seed(seed), handlers.trace() as tr:
            model(**model_kwargs)
        self.latent = OrderedDict((n, s["fn"]) for n, s in tr.sites.items()
                                  if s["type"] == "sample" and not s["is_observed"])
        if not self.latent:
            raise ValueError("the model has no latent sample sites")
        # a site may be vector- (tensor-) valued with element-wise independent distributions (numpyro: a distribution with a
        # batch shape): its elements take consecutive unconstrained coordinates, and the model sees a [chains, *shape] value
        self.shapes = OrderedDict((n, tuple(tr.sites[n]["value"].shape)) for n in self.latent)
        self.slices, offset = OrderedDict(), 0
        for n, shape in self.shapes.items():
            if tuple(self.latent[n].batch_shape) != shape:
                raise ValueError(f"latent site {n!r}: value shape {shape} is not the distribution's batch shape {tuple(self.latent[n].batch_shape)}")
            k = int(np.prod(shape)) if shape else 1
            self.slices[n] = (offset, k)
            offset += k
        self.deterministic = [n for n, s in tr.sites.items() if s["type"] == "deterministic"]
        self.bij = OrderedDict((n, biject_to(d.support)) for n, d in self.latent.items())
        self.dim = offset
        self.scalar_sites = all(shape == () for shape in self.shapes.values())
        # bijections + log priors + log-Jacobians of all latent sites as one kernel launch, when every site is in the fused
        # families (fused_sites.py: one descriptor per unconstrained coordinate, a tensor-valued site one per element);
        # otherwise the generic torch path below
        self.site_table = None
        self._ones: dict = {}
        if torch.device(device).type == "cuda":
            from . import fused_sites

            self.site_table = fused_sites.build_table(self.latent.values())

    def _coords(self, z: torch.Tensor, name: str) -> torch.Tensor:
        """The unconstrained coordinates of site ``name``: [..., *shape]."""
        o, k = self.slices[name]
        shape = self.shapes[name]
        return z[..., o] if shape == () else z[..., o:o + k].reshape(tuple(z.shape[:-1]) + shape)

    def constrain(self, z: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((n, b(self._coords(z, n))) for n, b in self.bij.items())

    def initial(self, chains: int, strategy, seed: int) -> torch.Tensor:
        gen = torch.Generator().manual_seed(seed)
        kind, n = strategy if isinstance(strategy, tuple) else strategy()
        cols = []
        for name, d in self.latent.items():
            draws = d.sample(gen, (chains, n)).reshape((chains, n) + self.shapes[name])
            x = draws.median(dim=1).values if kind == "median" else draws[:, 0]
            cols.append(self.bij[name].inv(x).reshape(chains, -1))
        return torch.cat(cols, dim=1).to(self.device)

    def log_joint(self, z: torch.Tensor):
        """(log p(x, obs) + log|dx/dz|) per chain, and the trace."""
        C = z.shape[0]
        if self.site_table is not None and z.is_cuda and z.dim() == 2 and z.dtype == torch.float64:
            from .fused_sites import LatentSites

            x_all, total = LatentSites.apply(z, self.site_table)
            x = OrderedDict((n, self._coords(x_all, n)) for n in self.bij)
            with handlers.substitute(x), handlers.trace() as tr:
                self.model(**self.kwargs)
        else:
            x = self.constrain(z)
            with handlers.substitute(x), handlers.trace() as tr:
                self.model(**self.kwargs)
            total = torch.zeros(C, dtype=torch.float64, device=z.device)
            for name, b in self.bij.items():
                lp = self.latent[name].log_prob(x[name]) + b.log_abs_det_jacobian(self._coords(z, name))
                total = total + (lp if lp.dim() == 1 else lp.reshape(C, -1).sum(-1))
        for name, s in tr.sites.items():
            if s["type"] == "sample" and s["is_observed"]:
                lp = s["fn"].log_prob(s["value"].to(z.device))
                total = total + lp.reshape(C, -1).sum(-1)
            elif s["type"] == "factor":
                v = s["value"]
                total = total + (v if v.dim() == 1 and v.shape[0] == C else v.reshape(C, -1).sum(-1))
        return total, tr

    def potential_and_grad(self, z: torch.Tensor):
        z = z.detach().requires_grad_(True)
        # every chain's parameters depend on its own row of z only: lets the differentiable solve seed
        # its tangents along the D latent coordinates instead of the P ODE parameters (autodiff.py)
        z._dynode_rowwise = True
        lj, _ = self.log_joint(z)
        minus = self._ones.get(lj.shape[0])
        if minus is None or minus.device != lj.device:
            minus = self._ones[lj.shape[0]] = -torch.ones_like(lj)
        (g,) = torch.autograd.grad(lj, z, minus)             # gradient of the potential U = -log joint
        return -lj.detach(), g

    def potential_and_grad_fd(self, z: torch.Tensor, eps: float = 1e-4):
        """The same pair by central differences of the log density along the D unconstrained coordinates: ONE batched
        evaluation of (1 + 2 D) C rows, no autograd -- for models whose solve has no tangent kernels (the SEIP family).
        The discrete solve must be a smooth function of the parameters for this to be accurate: use
        ``SolverParams(constant_step_size=...)`` (an adaptive controller changes its accept / reject decisions between
        the perturbed rows).  A sampler stays exact with an approximate gradient -- the leapfrog map is reversible and
        volume preserving for any position-dependent force, the accept step uses the true density -- only less efficient."""
        C, D = z.shape
        z = z.detach()
        steps = eps * torch.eye(D, dtype=z.dtype, device=z.device)
        rows = [z] + [z + sign * steps[d] for d in range(D) for sign in (1.0, -1.0)]
        with torch.no_grad():
            lj = self.log_joint(torch.cat(rows, dim=0))[0].reshape(1 + 2 * D, C)
        grad = torch.stack([-(lj[1 + 2 * d] - lj[2 + 2 * d]) / (2.0 * eps) for d in range(D)], dim=1)
        return -lj[0], grad

    def graphed(self, chains: int):
        """``potential_and_grad`` for a fixed number of chains as ONE HIP-graph replay.

        The model function, the handlers, pydantic copies, the ctypes call into the kernel and the
        autograd bookkeeping run once, at capture; every later evaluation is a graph launch
        (measured on cfg 4, 128 chains: 1.76 ms eager -> 0.37 ms replayed).  Falls back to the
        eager function when the model cannot be captured (e.g. it synchronises with the host).
        """
        if self.device.type != "cuda":
            return self.potential_and_grad
        static_z = torch.zeros((chains, self.dim), dtype=torch.float64, device=self.device)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):                       # warm-up: caches, allocator pools
                    self.potential_and_grad(static_z)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out_u, out_g = self.potential_and_grad(static_z)
        except Exception as err:  # pragma: no cover - depends on the user's model
            torch.cuda.synchronize()
            print(f"[dynode_amd] potential not graph-capturable ({type(err).__name__}); running eagerly", file=sys.stderr, flush=True)
            return self.potential_and_grad

        def replay(z):
            static_z.copy_(z)
            graph.replay()
            return out_u.clone(), out_g.clone()

        replay.graph = graph  # keep alive
        return replay


class _FiniteDifferenceLogJoint(torch.autograd.Function):
    """log density of a batch of unconstrained points as a differentiable tensor whose backward pass uses the
    finite-difference gradient of :meth:`Potential.potential_and_grad_fd` (for SVI on models without tangent kernels)."""

    @staticmethod
    def forward(ctx, z, potential, eps):
        u, g = potential.potential_and_grad_fd(z, eps)
        ctx.save_for_backward(g)
        return -u

    @staticmethod
    def backward(ctx, grad_out):
        (g,) = ctx.saved_tensors
        return -(g * grad_out[:, None]), None, None


class _FoldedLogJoint(torch.autograd.Function):
    """log density of a batch of unconstrained points through a `folded.FoldedPotential` (three launches: sites and parameter
    map, the gradient-solve with its fused likelihood, the combine) as a differentiable tensor -- SVI's ELBO on models with the
    structure, instead of the model's torch program and its backward pass."""

    @staticmethod
    def forward(ctx, z, folded):
        u, g = folded(z)
        ctx.save_for_backward(g)
        return -u

    @staticmethod
    def backward(ctx, grad_out):
        (g,) = ctx.saved_tensors
        return -(g * grad_out[:, None]), None


class MCMCResult:
    """What ``infer`` returns: the sampler output plus numpyro-style accessors."""

    def __init__(self, potential: Potential, nuts: NUTSResult, num_chains: int):
        self.potential, self.nuts, self.num_chains = potential, nuts, num_chains
        self.num_samples = int(nuts.samples.shape[1])       # numpyro's MCMC.num_samples

    def get_samples(self, group_by_chain: bool = False) -> dict:
        x = self.potential.constrain(self.nuts.samples)            # [C, N, *shape] per site
        return {n: (v if group_by_chain else v.reshape((-1,) + tuple(v.shape[2:]))) for n, v in x.items()}

    @property
    def last_state(self):
        return self.nuts.samples[:, -1]

    def summary(self) -> dict:
        """site -> {mean, std, median, 5.0%, 95.0%, n_eff, r_hat}: the columns of numpyro's ``MCMC.print_summary``
        (effective sample size and split R-hat from ``infer/diagnostics.py``); a vector-valued site has a row per element,
        ``name[i]`` (``name[i,j]`` ...), as numpyro prints them."""
        from .diagnostics import effective_sample_size, split_rhat

        flat = {}
        for n, v in self.get_samples(group_by_chain=True).items():
            if v.dim() == 2:
                flat[n] = v
            else:
                for idx in np.ndindex(*v.shape[2:]):
                    flat[f"{n}[{','.join(map(str, idx))}]"] = v[(slice(None), slice(None)) + idx]
        out = {}
        for n, v in flat.items():
            x = v.detach().double().cpu().numpy()
            q = np.quantile(x, [0.05, 0.5, 0.95])
            out[n] = {"mean": float(x.mean()), "std": float(x.std(ddof=1)), "median": float(q[1]), "5.0%": float(q[0]),
                      "95.0%": float(q[2]), "n_eff": effective_sample_size(x), "r_hat": split_rhat(x)}
        return out

    def print_summary(self) -> str:
        lines = [f"{'site':32s} {'mean':>10s} {'std':>10s} {'median':>10s} {'5.0%':>10s} {'95.0%':>10s} {'n_eff':>10s} {'r_hat':>7s}"]
        for n, r in self.summary().items():
            lines.append(f"{n:32s} {r['mean']:10.4f} {r['std']:10.4f} {r['median']:10.4f} {r['5.0%']:10.4f} {r['95.0%']:10.4f} "
                         f"{r['n_eff']:10.1f} {r['r_hat']:7.3f}")
        lines.append(f"divergences: {int(self.nuts.diverging.sum())}, mean accept prob: {float(self.nuts.accept_prob.mean()):.3f}, "
                     f"mean leapfrogs/transition: {float(self.nuts.num_steps.double().mean()):.2f}")
        text = "\n".join(lines)
        print(text)
        return text


class InferenceGroups:
    """What ``to_arviz`` returns when arviz is not installed: the groups of an
    ``arviz.InferenceData`` as plain dicts of tensors, every variable with leading
    ``(chain, draw)`` axes (``chain`` = 1 for SVI, prior and predictive groups, as
    ``az.from_numpyro`` lays them out).  ``to_inference_data()`` converts when arviz is present."""

    GROUPS = ("posterior", "posterior_predictive", "prior", "log_likelihood", "sample_stats", "observed_data")

    def __init__(self, **groups):
        for g in self.GROUPS:
            setattr(self, g, dict(groups.get(g) or {}))

    def groups(self) -> list:
        return [g for g in self.GROUPS if getattr(self, g)]

    def __repr__(self):
        return "InferenceGroups(" + ", ".join(f"{g}: {sorted(getattr(self, g))}" for g in self.groups()) + ")"

    def to_inference_data(self):
        import arviz as az  # noqa: F401  (optional dependency)

        conv = lambda d: {k: v.detach().cpu().numpy() for k, v in d.items()}
        return az.from_dict(**{g: conv(getattr(self, g)) for g in self.groups()})


def _as_draws(d: dict) -> dict:
    """(n, ...) -> (1, n, ...): a single "chain" axis in front, arviz's layout for unchained groups."""
    return {k: v.unsqueeze(0) for k, v in d.items() if isinstance(v, torch.Tensor)}


class InferenceProcess(BaseModel):
    """Abstract inference process (reference inference.py:29-117)."""

    model_config = ConfigDict(arbitrary_types_allowed=True)
    numpyro_model: Callable = Field(description="model(**kwargs): samples parameters, simulates, scores observations")
    inference_prngkey: int = 8675314
    _inference_complete: bool = PrivateAttr(default=False)
    _inferer: Optional[Any] = PrivateAttr(default=None)
    _inference_state: Optional[Any] = PrivateAttr(default=None)
    _inferer_kwargs: Optional[dict] = PrivateAttr(default_factory=dict)
    _folded_potential: bool = PrivateAttr(default=False)

    def infer(self, **kwargs):
        raise NotImplementedError("Inference process not implemented, please use a subclass.")

    def get_samples(self, group_by_chain=False, exclude_deterministic=True) -> dict:
        raise NotImplementedError("get_samples() process not implemented, please use a subclass.")

    def to_arviz(self):
        raise NotImplementedError("to_arviz() process not implemented, please use a subclass.")

    # ---- shared by the subclasses' to_arviz (reference inference.py:208-241, 368-405)
    def _predictive_groups(self, posterior_flat: dict, num_prior: int) -> dict:
        """posterior predictive (the model re-run on the posterior draws) and prior predictive, each
        as ONE batched solve, plus the pointwise log-likelihood of the observed sites."""
        from .predictive import Predictive

        kw = self._inferer_kwargs
        post_pred = Predictive(self.numpyro_model, posterior_samples=posterior_flat, return_observed=True)(
            rng_key=self.inference_prngkey, **kw)
        prior = Predictive(self.numpyro_model, num_samples=num_prior, return_observed=True)(
            rng_key=self.inference_prngkey, **kw)
        n = next(iter(posterior_flat.values())).shape[0]
        data = {k: torch.as_tensor(v, dtype=torch.float64) for k, v in posterior_flat.items()}
        loglik, observed = {}, {}
        with torch.no_grad(), handlers.substitute(data), handlers.trace() as tr:
            self.numpyro_model(**kw)
        for name, site in tr.sites.items():
            if site["type"] == "sample" and site["is_observed"]:
                lp = site["fn"].log_prob(site["value"])
                loglik[name] = lp if lp.dim() and lp.shape[0] == n else lp.expand((n,) + tuple(lp.shape))
                observed[name] = site["value"]
        for name, obs in observed.items():           # numpyro reports observed sites once per draw
            for grp, m in ((post_pred, n), (prior, num_prior)):
                if name in grp and tuple(grp[name].shape) == tuple(obs.shape):
                    grp[name] = obs.expand((m,) + tuple(obs.shape))
        return dict(posterior_predictive=_as_draws(post_pred), prior=_as_draws(prior), log_likelihood=loglik,
                    observed_data=observed)

    @staticmethod
    def _finish_arviz(groups: "InferenceGroups"):
        try:
            return groups.to_inference_data()
        except ImportError:
            return groups


class MCMCProcess(InferenceProcess):
    """Fit the model with NUTS (dense mass, init-to-median), all local chains on this GPU.

    Under ``torch.distributed`` (one process per GPU) every rank runs ``num_chains / world``
    chains with a rank-offset seed; ``get_samples(gather=True)`` collects them on rank 0 over
    RCCL -- the only collective of the inference path.

    The reference forwards ``nuts_kwargs`` to ``numpyro.infer.NUTS`` and ``mcmc_kwargs`` to ``numpyro.infer.MCMC``
    verbatim (reference inference.py:127-131,149-162).  Here every key is either honoured or refused -- nothing is
    dropped silently (`_check_kwargs`):

    ``nuts_kwargs``  ``target_accept_prob`` (0.8), ``step_size`` (1.0); ``adapt_step_size`` / ``adapt_mass_matrix`` /
                     ``regularize_mass_matrix`` / ``find_heuristic_step_size`` only at numpyro's defaults (True, True,
                     True, False: what the sampler kernels implement); ``forward_mode_differentiation`` either way (the
                     gradient-solve IS forward mode; the value is the same).  ``dense_mass``, ``max_tree_depth`` and
                     ``init_strategy`` raise ``TypeError`` exactly as in the reference, whose ``NUTS(...)`` call passes
                     them itself ("got multiple values for keyword argument").
    ``mcmc_kwargs``  numpyro's: ``chain_method`` ("parallel" | "vectorized" | "sequential": the chains of a rank always
                     advance together in one batch, which is what all three produce draw for draw), ``thinning``
                     (keep every n-th draw), ``jit_model_args`` (ignored: nothing is traced per argument),
                     ``postprocess_fn`` refused.  This build's own switches:
                     ``sampler``    ``"kernel"`` (default: sampler iteration as one HIP kernel, whole iteration replayed as a
                                    HIP graph), ``"graph"`` / ``"eager"`` (torch-op sampler step, replayed / op by op),
                                    ``"ensemble"`` (gradient-free stretch moves, ``infer/ensemble.py``; ``stretch``, ``thin``)
                     ``adaptation`` ``"per_chain"`` (default, numpyro's behaviour: every chain adapts its own mass matrix
                                    and step size) or ``"pooled"`` (opt-in: window statistics merged over the chains of
                                    this GPU -- shorter warm-up tails for many chains, not what numpyro does)
                     ``gradient``   ``"autograd"`` (default: tangent kernels + autograd) or ``"finite_difference"``
                                    (``fd_step``; for models without tangent kernels, with a constant solver step)
                     ``fold``       ``True`` (default) / ``False`` / ``"verbose"``: a model whose ODE parameters are monomials
                                    of its sampled sites and whose only likelihood is the solve's (``infer/folded.py``: the
                                    reference's own inference example is one) is evaluated in three launches per gradient
                                    instead of ~26; any other model silently keeps the general potential (``"verbose"`` says why)
                     ``fuse``       ``True`` (default) / ``False``: with a folded potential (at most eight sites), the
                                    gradient-solve's waves also run the sampler's side for the chains they scored
                                    (``dyn_solver_opts::nuts_tail``): a sampler iteration is ONE launch, same draws
    """

    num_samples: PositiveInt
    num_warmup: PositiveInt
    num_chains: PositiveInt
    nuts_max_tree_depth: PositiveInt
    nuts_init_strategy: Callable = init_to_median
    mcmc_kwargs: dict = Field(default_factory=dict)
    nuts_kwargs: dict = Field(default_factory=dict)
    progress_bar: bool = True

    _NUTS_FIXED = {"adapt_step_size": True, "adapt_mass_matrix": True, "regularize_mass_matrix": True,
                   "find_heuristic_step_size": False}
    _MCMC_OWN = ("sampler", "adaptation", "gradient", "fd_step", "stretch", "thin", "hip_graph", "fold", "fuse")

    def _check_kwargs(self) -> int:
        """Every key of ``nuts_kwargs`` / ``mcmc_kwargs`` is honoured or refused; returns the thinning factor."""
        for k, v in self.nuts_kwargs.items():
            if k in ("dense_mass", "max_tree_depth", "init_strategy"):
                # the reference's NUTS(model, dense_mass=True, max_tree_depth=..., init_strategy=..., **nuts_kwargs)
                raise TypeError(f"NUTS() got multiple values for keyword argument '{k}'")
            if k in ("target_accept_prob", "step_size", "forward_mode_differentiation"):
                continue
            if k in self._NUTS_FIXED:
                if bool(v) != self._NUTS_FIXED[k]:
                    raise NotImplementedError(f"nuts_kwargs[{k!r}] = {v!r}: the sampler kernels implement numpyro's default "
                                              f"({self._NUTS_FIXED[k]}) only")
                continue
            raise TypeError(f"nuts_kwargs: unsupported NUTS argument {k!r} (supported: target_accept_prob, step_size, "
                            f"forward_mode_differentiation, {', '.join(self._NUTS_FIXED)})")
        thinning = 1
        for k, v in self.mcmc_kwargs.items():
            if k in self._MCMC_OWN or k == "jit_model_args":
                continue
            if k == "chain_method":
                if v not in ("parallel", "vectorized", "sequential"):
                    raise ValueError(f"mcmc_kwargs['chain_method'] = {v!r}: only 'parallel', 'sequential' or 'vectorized' are supported")
                continue
            if k == "thinning":
                thinning = int(v)
                if thinning < 1 or self.num_samples % thinning:
                    raise ValueError("mcmc_kwargs['thinning'] must be a positive divisor of num_samples")
                continue
            if k in ("num_warmup", "num_samples", "num_chains", "progress_bar"):
                raise TypeError(f"MCMC() got multiple values for keyword argument '{k}'")
            raise TypeError(f"mcmc_kwargs: unsupported MCMC argument {k!r} (supported: chain_method, thinning, jit_model_args, "
                            f"{', '.join(self._MCMC_OWN)})")
        return thinning

    def infer(self, **kwargs) -> MCMCResult:
        from ..engine import require_gpu

        thinning = self._check_kwargs()
        if torch.cuda.is_available() or self.mcmc_kwargs.get("sampler") != "eager":
            device = require_gpu()
        else:
            # the op-by-op torch sampler, asked for by name, also runs on the host -- for models that are plain torch
            # code (the reference's tests/test_infer/test_inference_processes.py fits a Normal); a model that calls
            # simulate() still needs the GPU and says so (engine.require_gpu)
            device = torch.device("cpu")
        rank, world = sharding.world()
        lo, hi = sharding.shard_bounds(self.num_chains, rank, world)
        local = hi - lo
        pot = Potential(self.numpyro_model, kwargs, self.inference_prngkey, device)
        z0 = pot.initial(self.num_chains, self.nuts_init_strategy, self.inference_prngkey)[lo:hi]
        # default ("kernel"): an iteration = the potential (model, fused gradient-solve kernel,
        # autograd) + ONE hand-written sampler kernel (dyn_nuts_advance), captured as a HIP graph.
        # Chains adapt one by one as in numpyro; mcmc_kwargs={"adaptation": "pooled"} merges the mass-matrix
        # windows over the chains of this GPU instead.
        # mcmc_kwargs={"sampler": "graph"} replays the torch-op sampler step instead,
        # {"sampler": "eager"} (or the older {"hip_graph": False}) runs it op by op.
        kind = self.mcmc_kwargs.get("sampler", "kernel" if self.mcmc_kwargs.get("hip_graph", True) else "eager")
        if kind == "ensemble":
            # gradient-free: the walkers are the chains, every move scores half of them in one batched solve
            # (infer/ensemble.py) -- for members of the kernel family without tangent planes (SEIP)
            from .ensemble import EnsembleSampler

            sampler = EnsembleSampler(lambda z: pot.log_joint(z)[0], stretch=self.mcmc_kwargs.get("stretch", 2.0),
                                      seed=self.inference_prngkey + 7919 * rank)
            total = self.num_warmup + self.num_samples * int(self.mcmc_kwargs.get("thin", 1))

            def progress_e(it, warm):
                if self.progress_bar and rank == 0 and (it + 1) % max(1, total // 10) == 0:
                    print(f"[ensemble] {'warmup' if warm else 'sample'} {it + 1}/{total} ({local} walkers on this GPU)", flush=True)

            res = sampler.run(z0, self.num_warmup, self.num_samples, thin=int(self.mcmc_kwargs.get("thin", 1)), progress=progress_e)
            out = MCMCResult(pot, res, local)
            self._inference_complete, self._inferer, self._inference_state = True, out, out.last_state
            self._inferer_kwargs = kwargs
            return out
        from .. import _abi

        if kind == "kernel" and (pot.dim > _abi.NUTS_MAX_DIM or self.nuts_max_tree_depth > _abi.NUTS_MAX_DEPTH):
            # beyond what the sampler kernel is compiled for: the torch-op sampler under a HIP graph -- the same algorithm,
            # ~400 small launches per iteration instead of one.  Said out loud: it is a different performance class.
            import warnings

            warnings.warn(f"MCMCProcess: {pot.dim} sampled dimensions / tree depth {self.nuts_max_tree_depth} exceed the sampler kernel's "
                          f"limits ({_abi.NUTS_MAX_DIM} / {_abi.NUTS_MAX_DEPTH}, include/dynode_hip.h): running the torch-op sampler "
                          f"(mcmc_kwargs={{'sampler': 'graph'}}) instead of dyn_nuts_advance", RuntimeWarning, stacklevel=2)
            kind = "graph"
        if kind == "kernel" and pot.dim > _abi.NUTS_REG_DIM and self.mcmc_kwargs.get("adaptation", "per_chain") == "pooled":
            raise NotImplementedError(f"adaptation='pooled' is compiled for up to {_abi.NUTS_REG_DIM} sampled dimensions (this model: {pot.dim}); "
                                      "use the default per-chain adaptation")
        cls = {"kernel": KernelNUTS, "graph": GraphNUTS, "eager": BatchedNUTS}[kind]
        extra = {"adaptation": self.mcmc_kwargs.get("adaptation", "per_chain"),
                 "fuse": bool(self.mcmc_kwargs.get("fuse", True))} if kind == "kernel" else {}
        pg = pot.potential_and_grad
        if self.mcmc_kwargs.get("gradient", "autograd") == "finite_difference":
            # NUTS for models without tangent kernels: central differences over the latent coordinates (Potential above)
            fd_step = float(self.mcmc_kwargs.get("fd_step", 1e-4))
            pg = lambda z: pot.potential_and_grad_fd(z, fd_step)  # noqa: E731
            if kind == "kernel":
                extra["use_graph"] = False          # the model usually syncs with the host (numpy-built parameter tables)
        elif kind == "kernel" and self.mcmc_kwargs.get("fold", True):
            from .folded import discover

            folded = discover(pot, seed=self.inference_prngkey, verbose=self.mcmc_kwargs.get("fold") == "verbose")
            if folded is not None:
                pg = folded
        self._folded_potential = hasattr(pg, "into")
        sampler = cls(pg, max_tree_depth=self.nuts_max_tree_depth,
                      target_accept=self.nuts_kwargs.get("target_accept_prob", 0.8),
                      seed=self.inference_prngkey + 7919 * rank, **extra)
        total = self.num_warmup + self.num_samples

        def progress(it, warm):
            if self.progress_bar and rank == 0 and (it + 1) % max(1, total // 10) == 0:
                print(f"[nuts] {'warmup' if warm else 'sample'} {it + 1}/{total} ({local} chains on this GPU)", flush=True)

        try:
            res = sampler.run(z0, self.num_warmup, self.num_samples,
                              init_step_size=self.nuts_kwargs.get("step_size", 1.0), progress=progress)
        except Exception as err:
            from .folded import FoldMismatch

            if not isinstance(err, FoldMismatch):
                raise
            # the model's parameter map is not the monomial it looked like where the chains went: sample the model's own
            # log joint instead (same kernels, more launches) -- never the extrapolated one
            import sys

            print(f"[dynode_amd] {err}; restarting with the general potential", file=sys.stderr, flush=True)
            self._folded_potential = False
            sampler = cls(pot.potential_and_grad, max_tree_depth=self.nuts_max_tree_depth,
                          target_accept=self.nuts_kwargs.get("target_accept_prob", 0.8),
                          seed=self.inference_prngkey + 7919 * rank, **extra)
            res = sampler.run(z0, self.num_warmup, self.num_samples,
                              init_step_size=self.nuts_kwargs.get("step_size", 1.0), progress=progress)
        if thinning > 1:   # numpyro keeps the draws whose (1-based) index is a multiple of `thinning`
            keep = slice(thinning - 1, None, thinning)
            res.samples, res.accept_prob = res.samples[:, keep].contiguous(), res.accept_prob[:, keep].contiguous()
            res.num_steps, res.diverging = res.num_steps[:, keep].contiguous(), res.diverging[:, keep].contiguous()
        out = MCMCResult(pot, res, local)
        # which sampler ran (a class name of infer/nuts.py) and, for the kernel sampler, launches per iteration (1: fused into the
        # gradient-solve; 2: gradient-solve + dyn_nuts_advance_mapped; None: a general potential's launches + dyn_nuts_advance)
        out.sampler, out.launches_per_iteration = type(sampler).__name__, getattr(sampler, "launches_per_iteration", None)
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

    def to_arviz(self):
        """Posterior, posterior predictive, prior, pointwise log-likelihood and sampler statistics in
        arviz's layout (reference inference.py:208-241).  Returns ``arviz.InferenceData`` when arviz
        is importable, else an `InferenceGroups` with the same groups."""
        if not self._inference_complete:
            raise AssertionError("Inference process not completed, please call infer() first.")
        posterior = self.get_samples(group_by_chain=True)
        C, N = next(iter(posterior.values())).shape[:2]
        flat = {k: v.reshape((C * N,) + tuple(v.shape[2:])) for k, v in posterior.items()}
        g = self._predictive_groups(flat, self.num_samples)
        g["posterior_predictive"] = {k: v.reshape((C, N) + tuple(v.shape[2:])) for k, v in g["posterior_predictive"].items()}
        g["log_likelihood"] = {k: v.reshape((C, N) + tuple(v.shape[1:])) for k, v in g["log_likelihood"].items()}
        nuts = self._inferer.nuts
        stats = {"diverging": nuts.diverging, "acceptance_rate": nuts.accept_prob, "n_steps": nuts.num_steps,
                 "step_size": nuts.step_size[:, None].expand(C, N)}
        return self._finish_arviz(InferenceGroups(posterior=posterior, sample_stats=stats, **g))

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



def marginal_cdfs_by_quadrature(potential: Potential, z_grids: list) -> list:
    """Marginal posterior CDFs of every latent site by tensor-grid quadrature in the UNCONSTRAINED
    coordinates (uniform grids ``z_grids``, one per site), returned as ``(x_grid, cdf, pmf)`` triples
    in the constrained coordinate (``pmf`` = the probability mass of each grid cell).  Integrating in z keeps priors that are singular at the edge of
    their support (the reference's Beta(0.5, 0.5) on r0) smooth: the density in z carries the
    Jacobian, so no grid point sits on an integrable singularity."""
    mesh = torch.meshgrid(*z_grids, indexing="ij")
    z = torch.stack([m.reshape(-1) for m in mesh], dim=1).to(potential.device)
    with torch.no_grad():
        lj, _ = potential.log_joint(z)
    p = torch.exp(lj - lj.max()).reshape(mesh[0].shape).cpu()
    p = p / p.sum()
    out = []
    for i, (grid, bij) in enumerate(zip(z_grids, potential.bij.values())):
        other = tuple(d for d in range(p.dim()) if d != i)
        marginal = p.sum(other) if other else p
        cdf = torch.cumsum(marginal, 0) - 0.5 * marginal            # midpoint rule
        out.append((bij(grid.to(torch.float64)).cpu().numpy(), cdf.numpy(), marginal.numpy()))
    return out


def ks_against_quadrature(potential: Potential, draws: dict, z_grids: list, thin: int = 20, cdfs: Optional[list] = None) -> dict:
    """Kolmogorov-Smirnov test of every latent site's draws (``[chains, draws]``, thinned by ``thin`` along the draws
    axis to decorrelate them) against the marginal CDF from :func:`marginal_cdfs_by_quadrature`, with the moments of
    both: ``{site: {"ks_p", "ks_stat", "n", "mean", "sd", "quad_mean", "quad_sd", "mean_z", "sd_z"}}`` where ``mean_z`` /
    ``sd_z`` are the errors of the sample mean / standard deviation in their Monte-Carlo standard errors."""
    import numpy as np
    from scipy import stats

    from .diagnostics import effective_sample_size

    out = {}
    # ``cdfs``: marginals computed elsewhere (the tests pass the C oracle's, so that the check does not lean on this package's
    # own solves), in the order of ``potential.bij``
    for (name, v), (grid, cdf, pmf) in zip(((n, draws[n]) for n in potential.bij),
                                            cdfs if cdfs is not None else marginal_cdfs_by_quadrature(potential, z_grids)):
        x = v.detach().cpu().numpy().astype(np.float64)
        thinned = x[:, ::thin].reshape(-1)
        res = stats.kstest(thinned, lambda q: np.interp(q, grid, cdf))
        qm = float((grid * pmf).sum())
        qs = float(np.sqrt((((grid - qm) ** 2) * pmf).sum()))
        ess = effective_sample_size(x)
        # the standard deviation's own Monte-Carlo error: delta method on the squared deviations, with THEIR effective sample size
        d2 = (x - qm) ** 2
        ess2 = effective_sample_size(d2)
        sd_se = float(np.sqrt(d2.var() / max(ess2, 1.0)) / (2.0 * qs))
        out[name] = {"ks_p": float(res.pvalue), "ks_stat": float(res.statistic), "n": int(thinned.size), "mean": float(x.mean()),
                     "sd": float(x.std()), "quad_mean": qm, "quad_sd": qs, "ess": float(ess), "ess_sq": float(ess2),
                     "mean_z": float((x.mean() - qm) / (qs / np.sqrt(max(ess, 1.0)))),
                     "sd_z": float((np.sqrt(d2.mean()) - qs) / sd_se)}
    return out


class Adam:
    """Optimizer marker with numpyro's constructor (``numpyro.optim.Adam(step_size=0.1)``)."""

    def __init__(self, step_size: float = 0.1, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
        self.step_size, self.b1, self.b2, self.eps = step_size, b1, b2, eps


class AutoMultivariateNormal:
    """Full-covariance Gaussian over the unconstrained latent space (numpyro's autoguide of the same
    name): z = loc + scale_tril @ eps.  ``init_scale`` = 0.1 as in numpyro."""

    def __init__(self, dim: int, init_loc: torch.Tensor, init_scale: float = 0.1):
        self.loc = init_loc.clone().requires_grad_(True)
        self.raw_tril = (torch.eye(dim, dtype=torch.float64, device=init_loc.device) * init_scale).requires_grad_(True)

    def scale_tril(self):
        L = torch.tril(self.raw_tril)
        diag = torch.diagonal(L)
        return L - torch.diag(diag) + torch.diag(torch.nn.functional.softplus(diag) + 1e-8)

    def parameters(self):
        return [self.loc, self.raw_tril]

    def sample(self, n: int, gen: torch.Generator):
        eps = torch.randn((n, self.loc.shape[0]), dtype=torch.float64, device=self.loc.device, generator=gen)
        L = self.scale_tril()
        return self.loc + eps @ L.T, eps

    def entropy(self):
        d = self.loc.shape[0]
        return 0.5 * d * (1.0 + math.log(2 * math.pi)) + torch.log(torch.diagonal(self.scale_tril())).sum()


class SVIResult:
    def __init__(self, potential: Potential, guide: AutoMultivariateNormal, losses: torch.Tensor):
        self.potential, self.guide, self.losses = potential, guide, losses
        self.params = {"auto_loc": guide.loc.detach(), "auto_scale_tril": guide.scale_tril().detach()}


class SVIProcess(InferenceProcess):
    """Stochastic variational inference (reference inference.py:244-302): AutoMultivariateNormal
    guide, Adam(0.1), Trace_ELBO.  Each iteration scores ``num_particles`` reparameterised draws
    with ONE batched gradient-solve (numpyro's default is a single particle per step)."""

    num_iterations: PositiveInt
    num_samples: PositiveInt
    guide_init_strategy: Callable = init_to_median
    optimizer: Any = Field(default_factory=lambda: Adam(step_size=0.1))
    num_particles: PositiveInt = 8
    progress_bar: bool = True
    guide_kwargs: dict = Field(default_factory=dict)
    svi_kwargs: dict = Field(default_factory=dict)      # {"gradient": "finite_difference", "fd_step": 1e-4}: see MCMCProcess

    def infer(self, **kwargs) -> SVIResult:
        from ..engine import require_gpu

        device = require_gpu()
        pot = Potential(self.numpyro_model, kwargs, self.inference_prngkey, device)
        init = pot.initial(1, self.guide_init_strategy, self.inference_prngkey)[0]
        guide = AutoMultivariateNormal(pot.dim, init, **self.guide_kwargs)
        o = self.optimizer
        opt = torch.optim.Adam(guide.parameters(), lr=o.step_size, betas=(o.b1, o.b2), eps=o.eps)
        gen = torch.Generator(device=device).manual_seed(self.inference_prngkey)
        losses = torch.empty(self.num_iterations, dtype=torch.float64, device=device)    # (read once, at the end: no host sync per step)
        fd = self.svi_kwargs.get("gradient", "autograd") == "finite_difference"
        # a model with the structure (infer/folded.py: monomial parameter map, the solve's own likelihood -- also written out
        # in torch, like the reference's model()) is scored by three launches per step; ``svi_kwargs={"fold": False}`` keeps
        # the model's torch program
        folded = None
        if not fd and self.svi_kwargs.get("fold", True):
            from .folded import discover

            folded = discover(pot, seed=self.inference_prngkey, verbose=self.svi_kwargs.get("fold") == "verbose")
        self._folded_potential = folded is not None
        for it in range(self.num_iterations):
            opt.zero_grad()
            z, _ = guide.sample(self.num_particles, gen)
            if fd:
                lj = _FiniteDifferenceLogJoint.apply(z, pot, float(self.svi_kwargs.get("fd_step", 1e-4)))
            elif folded is not None:
                lj = _FoldedLogJoint.apply(z, folded)
            else:
                lj, _ = pot.log_joint(z)
            finite = torch.isfinite(lj)
            lj = torch.where(finite, lj, torch.zeros_like(lj))
            loss = -(lj.sum() / finite.sum().clamp_min(1) + guide.entropy())      # -ELBO, reparameterised
            loss.backward()
            opt.step()
            losses[it] = loss.detach()
            if self.progress_bar and (it + 1) % max(1, self.num_iterations // 10) == 0:
                print(f"[svi] {it + 1}/{self.num_iterations} loss {float(loss):.3f}", flush=True)
        losses = losses.cpu()
        out = SVIResult(pot, guide, losses)
        self._inference_complete, self._inferer, self._inference_state = True, out, out.params
        self._inferer_kwargs = kwargs
        return out

    def get_samples(self, _: bool = False, exclude_deterministic: bool = True) -> dict:
        """Draws from the fitted guide, constrained; ``(num_samples,)`` per site (no chains in SVI)."""
        if not self._inference_complete:
            raise AssertionError("Inference process not completed, please call infer() first.")
        res: SVIResult = self._inferer
        gen = torch.Generator(device=res.guide.loc.device).manual_seed(self.inference_prngkey)
        with torch.no_grad():
            z, _ = res.guide.sample(self.num_samples, gen)
            samples = dict(res.potential.constrain(z))
            if not exclude_deterministic:
                _, tr = res.potential.log_joint(z)
                samples.update({n: s["value"] for n, s in tr.sites.items() if s["type"] == "deterministic"})
        return samples


    def to_arviz(self):
        """Posterior predictive, prior (``num_iterations`` draws, as the reference does) and pointwise
        log-likelihood (reference inference.py:368-405); the guide's draws are reported as a
        one-chain posterior group as well."""
        if not self._inference_complete:
            raise AssertionError("Inference process not completed, please call infer() first.")
        flat = self.get_samples()
        g = self._predictive_groups(flat, self.num_iterations)
        g["log_likelihood"] = _as_draws(g["log_likelihood"])
        return self._finish_arviz(InferenceGroups(posterior=_as_draws(flat), **g))


__all__ = ["Adam", "AutoMultivariateNormal", "InferenceGroups", "InferenceProcess", "MCMCProcess", "MCMCResult", "Potential", "SVIProcess",
           "SVIResult", "init_to_median", "init_to_sample", "log_posterior_grid", "marginal_cdfs_by_quadrature"]
