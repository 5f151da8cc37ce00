"""``simulate``: DynODE's ODE-solve entry point on the MI355X engine.

Mirror of /root/reference/src/dynode/simulation/odes.py:35-198 -- same signature, argument
meaning, output structure and error behaviour -- with ``diffrax.diffeqsolve`` replaced by the
fused HIP kernel behind ``dyn_solve_batch`` (include/dynode_hip.h).  One extension: any
parameter / initial-state array may carry a leading batch axis, in which case B trajectories
are integrated in one launch and every ``ys`` entry gains a leading batch axis.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np
import torch

from .. import _abi
from ..config import SolverParams
from ..engine import solve_batch
from ..rhs import AbstractODEParams, CompartmentalODE  # noqa: F401
from ..typing import CompartmentState, is_array

_X64 = False


def enable_x64(flag: bool = True) -> None:
    """Like ``jax.config.update("jax_enable_x64", ...)``: solve in float64.  Off by default --
    the reference never enables x64, so its effective dtype is float32."""
    global _X64
    _X64 = bool(flag)


class _LazyStats(dict):
    """``Solution.stats``; ``num_steps`` (= accepted + rejected) is added on first access."""

    def __missing__(self, key):
        if key == "num_steps":
            self[key] = self["num_accepted_steps"] + self["num_rejected_steps"]
            return self[key]
        raise KeyError(key)


class SolverError(RuntimeError):
    """The solve did not reach t1 (max_steps exhausted or non-finite state); the reference raises
    from diffeqsolve in the same situations (params.py:51-55)."""


@dataclass
class SaveAt:
    """What ``build_saveat`` returns: the save grid and the per-compartment mask (SubSaveAt)."""

    ts: np.ndarray
    mask: Optional[Tuple[bool, ...]] = None


@dataclass
class Solution:
    """Result of ``simulate`` (the fields of diffrax.Solution the reference's callers use).

    ``ys`` is a tuple with one array per compartment, leading time axis, i.e.
    ``ys[c].shape == (n_save, *compartment_shape)`` -- or ``(B, n_save, *compartment_shape)`` for a
    batched call; compartments excluded by ``sub_save_indices`` come back as ``(n_save, 0)``.
    Arrays are torch tensors resident on the GPU (views of one [B, n_save, D_saved] buffer).
    """

    ts: torch.Tensor
    ys: Tuple[torch.Tensor, ...]
    stats: dict = field(default_factory=dict)
    result: torch.Tensor = None      # per-trajectory status (0 ok, 1 max_steps, 2 non-finite)
    t0: float = 0.0
    t1: float = 0.0
    log_likelihood: Optional[torch.Tensor] = None   # only with ``simulate(..., observe=...)``


@dataclass
class PoissonObservation:
    """Observation model scored INSIDE the solve kernel (``simulate(..., observe=...)``).

    ``data ~ Poisson(max(v, floor))`` where ``v`` is compartment ``compartment`` of the state at the
    save times (``increments=False``; ``data`` has one row per save time) or its increase from one
    save time to the next (``increments=True``; one row fewer) -- the likelihood of the reference's
    model(): ``incidence = max(diff(R), 1e-6)``, ``obs ~ Poisson(incidence)``
    (examples/sir_infer_parameters.py:30-38).  No trajectory is written to memory; the solution
    carries ``log_likelihood`` (differentiable with respect to the ODE parameters) instead of ``ys``.
    """

    compartment: int
    data: object                      # array-like [n_obs, *compartment_shape]
    increments: bool = True
    floor: float = 1e-6


def build_saveat(start: float, stop, step: int = 1,
                 sub_save_indices: Optional[Tuple[int, ...]] = None, n_compartments: Optional[int] = None) -> SaveAt:
    """Save grid ``linspace(start, stop, int(stop // step) + 1)`` (odes.py:177-180; ``step <= 0``
    means 1) and the compartments to keep (odes.py:182-193)."""
    if step <= 0:
        step = 1
    ts = np.linspace(start, stop, int(stop // step) + 1)
    mask = None
    if sub_save_indices is not None:
        n = n_compartments if n_compartments is not None else (max(sub_save_indices) + 1 if len(sub_save_indices) else 0)
        bad = [i for i in sub_save_indices if not (-n <= i < n)]
        if bad:
            # the reference swallows the IndexError with a print (odes.py:194-197)
            print(f"An index passed to sub_save_indices was out of range for initial_state values. Exception: {bad}")
        else:
            keep = {i % n for i in sub_save_indices}
            mask = tuple(i in keep for i in range(n))
    return SaveAt(ts, mask)


_OBS_CACHE: dict = {}


def _observation_key(data):
    """Cache key of an observation table.  A tensor is identified by the OBJECT (its id) and its in-place version counter;
    the cache entry keeps a strong reference to it (`_observation_constants`), so the id -- and the storage behind it --
    cannot be handed to another tensor while the entry lives (a freed tensor's address is routinely reused by the next one
    of the same shape, with version 0 again).  No device-to-host copy: this runs at every potential evaluation of a
    sampler.  A numpy array is identified by its bytes (observation tables are small: time x groups)."""
    if isinstance(data, torch.Tensor):
        return ("t", id(data), data._version, tuple(data.shape), str(data.dtype), str(data.device))
    arr = np.ascontiguousarray(np.asarray(data))
    return ("n", arr.dtype.str, arr.shape, arr.tobytes())


def _observation_constants(data, dtype, device, pad_tiers=None):
    """(observations as a flat device tensor of the solve dtype, sum lgamma(data + 1)); cached per observation table
    (see `_observation_key`), so a caller that refills one buffer in place, or passes the next data set, is scored on the
    new values.  ``pad_tiers`` = (tiers, slots): vaccinated models keep 2 or 4 tier slots per age on the kernel's contact
    axis (axis 2 of the observation array, after time and age); the extra slots are filled with zeros."""
    key = (_observation_key(data), dtype, str(device), pad_tiers)
    hit = _OBS_CACHE.get(key)
    if hit is not None and isinstance(data, torch.Tensor) and hit[3] is not data:
        hit = None                                  # an id that outlived its tensor (cannot happen while the entry holds it)
    if hit is None:
        if len(_OBS_CACHE) > 16:
            _OBS_CACHE.clear()
        t = data.detach() if isinstance(data, torch.Tensor) else torch.as_tensor(np.asarray(data))
        t64 = t.to(device=device, dtype=torch.float64)
        shape = tuple(t.shape)
        lg = torch.lgamma(t64 + 1.0).sum()
        if pad_tiers is not None and t64.dim() >= 3 and t64.shape[2] == pad_tiers[0] and pad_tiers[1] > pad_tiers[0]:
            extra = list(t64.shape)
            extra[2] = pad_tiers[1] - pad_tiers[0]
            t64 = torch.cat([t64, t64.new_zeros(extra)], dim=2)
        # the last member pins a tensor argument: its id stays unique for as long as this entry can be hit
        hit = (t64.to(dtype).contiguous(), lg, shape, data if isinstance(data, torch.Tensor) else None)
        _OBS_CACHE[key] = hit
    return hit[:3]


def _simulate_observed(ode, ode_parameters, packed, saveat, t1, kw, observe, differentiable, sp, n_comp, y0_arg=None):
    from ..engine import require_gpu
    from ..infer.autodiff import solve_loglik_diff

    device = require_gpu()
    comp = int(observe.compartment) % n_comp
    want = tuple(packed.shapes[comp])
    pad = None
    if packed.tiers is not None:              # vaccinated model: the caller's arrays carry the tracked tiers only
        pad = (packed.tiers, want[1])
        want = (want[0], packed.tiers) + want[2:]
    obs_t, const, shape = _observation_constants(observe.data, kw["dtype"], device, pad)
    n_obs = len(saveat.ts) - int(bool(observe.increments))
    if tuple(shape) != (n_obs,) + want:
        raise ValueError(f"observations have shape {tuple(shape)}; expected {(n_obs,) + want} "
                         f"({'increments between' if observe.increments else 'values at'} {len(saveat.ts)} save times)")
    if pad is not None and pad[1] > pad[0]:
        # the padded tier slots stay empty: each scores 0 * log(floor) - floor; take that constant out again
        cells = n_obs * int(np.prod(packed.shapes[comp])) // pad[1] * (pad[1] - pad[0])
        const = const - float(observe.floor) * cells
    kw = {k: v for k, v in kw.items() if k != "save_mask"}
    if differentiable:
        params_t = ode.param_tensor(ode_parameters, device, packed)
    else:
        params_t = torch.as_tensor(packed.params, dtype=torch.float64, device=device)
    lp, status, n_acc, n_rej = solve_loglik_diff(packed.model, packed.y0 if y0_arg is None else y0_arg, params_t, packed.contact, t1, saveat.ts, obs_t,
                                                 comp, increments=observe.increments, floor=observe.floor, **kw)
    lp = lp - const
    batched = packed.batch is not None
    unb = (lambda t: t) if batched else (lambda t: t[0])
    stats = _LazyStats({"num_accepted_steps": unb(n_acc), "num_rejected_steps": unb(n_rej), "max_steps": sp.max_steps})
    from ..engine import _dev
    empty = torch.empty((lp.shape[0], len(saveat.ts), 0) if batched else (len(saveat.ts), 0), dtype=kw["dtype"], device=device)
    return Solution(ts=_dev(saveat.ts, kw["dtype"], device), ys=tuple(empty for _ in range(n_comp)), stats=stats,
                    result=unb(status), t0=0.0, t1=t1, log_likelihood=unb(lp))


def simulate(ode, duration_days, initial_state: CompartmentState, ode_parameters, solver_parameters: SolverParams,
             sub_save_indices: Optional[Tuple[int, ...]] = None, save_step: int = 1, *, dtype=None,
             throw: bool = True, observe: Optional["PoissonObservation"] = None) -> Solution:
    """Solve ``ode`` for ``duration_days`` days from ``initial_state`` (reference odes.py:35-145).

    Parameters follow the reference one for one.  ``ode`` is a :class:`CompartmentalODE` descriptor
    (see ``dynode_amd.rhs``); ``ode_parameters`` must be an instance of exactly ``ode.params_type``.

    Raises
    ------
    TypeError        ``initial_state`` holds something that is not an array (odes.py:93-98)
    AssertionError   wrong parameter type, or non-numeric ``duration_days`` (odes.py:100-112)
    SolverError      max_steps exhausted / non-finite state (only when ``throw``)
    """
    if any(not is_array(c) for c in initial_state):
        raise TypeError("Please pass numpy / torch arrays (not lists or scalars) as initial_state to ODEs")
    if not isinstance(ode, CompartmentalODE):
        raise TypeError(
            "ode must be a dynode_amd.rhs.CompartmentalODE descriptor: arbitrary Python callables cannot run "
            "inside the HIP kernel (see dynode_amd/rhs.py for the supported RHS family)")
    expected = ode.params_type
    assert type(ode_parameters) is expected, (
        f"passed {type(ode_parameters)} ode parameters, but your ODE model expects {expected}")
    assert isinstance(duration_days, (int, float)) and not isinstance(duration_days, bool), (
        "tf must be of type int or float")

    # a compartment of the initial state may itself come from a latent site: its gradient flows through dy0 seeds
    state_grad = ode.state_wants_grad(initial_state)
    differentiable = ode.wants_grad(ode_parameters) or state_grad
    packed = ode.pack(initial_state, ode_parameters, with_params=not differentiable)
    saveat = build_saveat(0.0, duration_days, save_step, sub_save_indices, len(initial_state))
    if dtype is None:
        dtype = torch.float64 if _X64 else torch.float32
    sp = solver_parameters
    kw = dict(t0=0.0, method=sp.solver_method.method, dtype=dtype, rtol=sp.ode_solver_rel_tolerance,
              atol=sp.ode_solver_abs_tolerance, max_steps=sp.max_steps,
              constant_dt=sp.constant_step_size if sp.constant_step_size > 0.0 else 0.0,
              # reference odes.py:115-131: the ConstantStepSize branch does not look at discontinuity_points
              jump_ts=() if sp.constant_step_size > 0.0 else sp.discontinuity_points, save_mask=saveat.mask)
    y0_arg = packed.y0
    if state_grad:
        from ..engine import require_gpu

        y0_arg = ode.state_tensor(initial_state, packed, require_gpu())
    if observe is not None:
        return _simulate_observed(ode, ode_parameters, packed, saveat, float(duration_days), kw, observe,
                                  differentiable, sp, len(initial_state), y0_arg)
    if differentiable:
        # a parameter carries an autograd graph (NUTS / SVI potential): differentiable solve.  No
        # host synchronisation here: a failed trajectory leaves +inf rows, which turn the
        # log-density non-finite and are rejected by the sampler like a divergence.
        throw = False
        from ..engine import require_gpu
        from ..infer.autodiff import solve_batch_diff

        params_t = ode.param_tensor(ode_parameters, require_gpu(), packed)
        res = solve_batch_diff(packed.model, y0_arg, params_t, packed.contact, float(duration_days), saveat.ts, **kw)
    else:
        res = solve_batch(packed.model, packed.y0, packed.params, packed.contact, float(duration_days), saveat.ts, **kw)

    if throw:
        bad = int((res.status != _abi.STATUS_OK).sum())
        if bad:
            first = int(torch.nonzero(res.status != _abi.STATUS_OK)[0])
            code = int(res.status[first])
            why = "max_steps reached" if code == _abi.STATUS_MAX_STEPS else "non-finite state"
            raise SolverError(f"{bad} of {res.status.numel()} trajectories failed ({why} at index {first}); "
                              f"raise SolverParams.max_steps or pass throw=False to inspect Solution.result")

    batched = packed.batch is not None
    n_save = res.ys.shape[1]
    ys, pos = [], 0
    keep = saveat.mask if saveat.mask is not None else (True,) * len(initial_state)
    for shape, saved in zip(packed.shapes, keep):
        if not saved:
            empty = (res.ys.shape[0], n_save, 0) if batched else (n_save, 0)
            ys.append(torch.empty(empty, dtype=res.ys.dtype, device=res.ys.device))
            continue
        size = int(np.prod(shape))
        block = res.ys[:, :, pos:pos + size].reshape((res.ys.shape[0], n_save) + tuple(shape))
        if packed.tiers is not None:                      # vaccination: drop the padded tier slots (axis after age)
            block = block[:, :, :, :packed.tiers]
        if packed.history_perm is not None:               # SEIP: back to the reference's immune-history bin order
            block = block[:, :, :, packed.history_perm]
        ys.append(block if batched else block[0])
        pos += size
    unb = (lambda t: t) if batched else (lambda t: t[0])
    stats = _LazyStats({
        "num_accepted_steps": unb(res.n_accept),
        "num_rejected_steps": unb(res.n_reject),
        "max_steps": sp.max_steps,
    })
    if not differentiable:        # inside a sampler's potential nobody reads it: one launch less per gradient
        stats["num_steps"]
    from ..engine import _dev
    ts = _dev(saveat.ts, res.ys.dtype, res.ys.device)
    return Solution(ts=ts, ys=tuple(ys), stats=stats, result=unb(res.status), t0=0.0, t1=float(duration_days))
