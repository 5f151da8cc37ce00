"""Differentiable batched solve: the fused kernel as a ``torch.autograd.Function``.

The reference obtains d(log-posterior)/d(parameters) by letting numpyro differentiate through
``diffeqsolve`` (examples/sir_infer_parameters.py:21-39 under
src/dynode/infer/inference.py:149-163).  Here the kernel itself returns the Jacobian of the saved
trajectory with respect to the P entries of the parameter vector (forward mode, identity seeds,
``dyn_solve_batch_jvp``); the backward pass is one contraction with the incoming cotangent, so any
torch code around ``simulate`` -- ``get_odeparams`` arithmetic, ``diff``/``clamp``, the Poisson
log-likelihood -- is differentiated by torch autograd.  An initial state that itself depends on a latent
site (a sampled initial-infection scale) is differentiated the same way: its D entries join the P parameters
as extra columns and are seeded through the kernel's ``dy0`` planes (`_joined`).
"""

from __future__ import annotations

import ctypes

import torch

from .. import _abi
from ..engine import _DTYPES, _METHODS, solve_batch, solve_batch_loglik


def _supported_nd(model, method: str, dtype, nd: int) -> bool:
    opts = _abi.SolverOptsC(_METHODS[method], _DTYPES[dtype], 1e-5, 1e-6, 10**6, 0.0, None, 0)
    return bool(_abi.lib().dyn_is_supported_jvp(ctypes.byref(model.c()), ctypes.byref(opts), nd))


def direction_chunks(model, method: str, dtype, P: int) -> list:
    """Split P seed directions into chunks the library has kernels for (largest first)."""
    if model.family == 1:      # SEIP: central differences of replayed solves take any number of directions at once (engine)
        return [P]
    sizes = [n for n in range(P, 0, -1) if _supported_nd(model, method, dtype, n)]
    if not sizes:
        from .. import jit

        if jit.enabled():          # no tangent kernel for this shape yet: build the 2-direction one (1 if P == 1)
            jit.ensure_kernel(model, dtype, method, min(P, 2))
            sizes = [n for n in range(P, 0, -1) if _supported_nd(model, method, dtype, n)]
    if not sizes:
        raise RuntimeError(f"no tangent kernel compiled for {model} (method={method}, dtype={dtype})")
    chunks, left = [], P
    while left:
        n = next((s for s in sizes if s <= left), None)
        if n is None:
            raise RuntimeError(f"cannot tile {P} tangent directions with compiled kernels {sizes}")
        chunks.append(n)
        left -= n
    return chunks


_SEEDS: dict = {}


def _identity_seeds(B: int, P: int, start: int, n: int, dtype, device) -> torch.Tensor:
    """[B, n, P] rows start..start+n of the identity for every batch member (constant: cached)."""
    key = (B, P, start, n, dtype, str(device))
    t = _SEEDS.get(key)
    if t is None:
        if len(_SEEDS) > 64:
            _SEEDS.clear()
        t = torch.eye(P, dtype=dtype, device=device)[start:start + n].unsqueeze(0).expand(B, n, P).contiguous()
        _SEEDS[key] = t
    return t


# ------------------------------------------------------------------ seeding along the latent coordinates
_BASIS: dict = {}


def _rowwise_leaf(params: torch.Tensor):
    """The sampler's unconstrained coordinates z [B, D] if ``params`` [B, P] is computed from them row
    by row (`Potential.potential_and_grad` marks its leaf with ``_dynode_rowwise``) and D < P; else None."""
    if params.grad_fn is None:
        return None
    seen, leaves, stack = set(), {}, [params.grad_fn]
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        var = getattr(fn, "variable", None)
        if var is not None:
            leaves[id(var)] = var
        stack.extend(f for f, _ in fn.next_functions)
    if len(leaves) != 1:
        return None
    (leaf,) = leaves.values()
    if not getattr(leaf, "_dynode_rowwise", False) or leaf.dim() != 2 or leaf.shape[0] != params.shape[0]:
        return None
    return leaf if leaf.shape[1] < params.shape[1] else None


def _latent_seeds(params: torch.Tensor, leaf: torch.Tensor, dtype) -> torch.Tensor:
    """[B, D, P]: d params[b, :] / d z[b, k] -- one batched backward through the (small) graph from z to
    the parameter matrix, so that the tangent solve runs D directions instead of P."""
    B, P = params.shape
    key = (B, P, params.dtype, str(params.device))
    basis = _BASIS.get(key)
    if basis is None:
        if len(_BASIS) > 16:
            _BASIS.clear()
        basis = torch.eye(P, dtype=params.dtype, device=params.device)[:, None, :].expand(P, B, P).contiguous()
        _BASIS[key] = basis
    (rows,) = torch.autograd.grad(params, leaf, grad_outputs=basis, is_grads_batched=True, retain_graph=True)
    return rows.permute(1, 2, 0).to(dtype).contiguous()          # [P, B, D] -> [B, D, P]


def _split_joined(model, q: torch.Tensor, seeds: torch.Tensor, y0, dtype):
    """``q`` = [params | y0] when the initial state carries an autograd graph (`_joined`), else the parameter
    matrix alone: -> (params, y0, dparams seeds, dy0 seeds or None) for one tangent solve."""
    P = model.param_dim
    if q.shape[1] == P:
        return q.to(dtype), y0, seeds, None
    return q[:, :P].to(dtype).contiguous(), q[:, P:].to(dtype).contiguous(), seeds[:, :, :P], seeds[:, :, P:]


def _joined(params: torch.Tensor, y0) -> torch.Tensor:
    """[params | y0] per trajectory when ``y0`` is a tensor that requires grad (a compartment computed from a latent
    site): the initial state is then differentiated like P + D more parameters, through the kernel's dy0 planes."""
    if not (isinstance(y0, torch.Tensor) and y0.requires_grad):
        return params
    y0b = y0 if y0.dim() == 2 else y0.unsqueeze(0).expand(params.shape[0], y0.shape[0])
    return torch.cat([params, y0b.to(device=params.device, dtype=params.dtype)], dim=1)


class _DiffSolveLatent(torch.autograd.Function):
    """`_DiffSolve` differentiated along the sampler's D latent coordinates instead of the P ODE
    parameters: ``seeds`` [B, D, P (+ state)] = d [params | y0] / d z; the gradient is returned to ``leaf`` directly."""

    @staticmethod
    def forward(ctx, leaf, params, seeds, model, y0, contact, t1, save_ts, kw):
        dtype = kw.get("dtype", torch.float32)
        method = kw.get("method", "tsit5")
        D = seeds.shape[1]
        pk, y0, dps, dys = _split_joined(model, params, seeds, y0, dtype)
        ctx.set_materialize_grads(False)
        jac, res, start = [], None, 0
        for n in direction_chunks(model, method, dtype, D):
            res = solve_batch(model, y0, pk, contact, t1, save_ts, dparams=dps[:, start:start + n].contiguous(),
                              dy0=None if dys is None else dys[:, start:start + n].contiguous(), **kw)
            jac.append(res.dys)
            start += n
        J = jac[0] if len(jac) == 1 else torch.cat(jac, dim=2)     # [B, n_save, D, D_saved]
        ctx.save_for_backward(J)
        ctx.in_dtype = leaf.dtype
        ctx.mark_non_differentiable(res.status, res.n_accept, res.n_reject)
        return res.ys, res.status, res.n_accept, res.n_reject

    @staticmethod
    def backward(ctx, g_ys, *_unused):
        (J,) = ctx.saved_tensors
        if g_ys is None:
            return (None,) * 9
        g = torch.einsum("btd,btkd->bk", g_ys.to(J.dtype), J)
        return (g.to(ctx.in_dtype),) + (None,) * 8


class _DiffLogLikLatent(torch.autograd.Function):
    """`_DiffLogLik` along the latent coordinates (see `_DiffSolveLatent`)."""

    @staticmethod
    def forward(ctx, leaf, params, seeds, model, y0, contact, t1, save_ts, obs, comp, increments, floor, kw):
        dtype = kw.get("dtype", torch.float32)
        method = kw.get("method", "tsit5")
        D = seeds.shape[1]
        pk, y0, dps, dys = _split_joined(model, params, seeds, y0, dtype)
        ctx.set_materialize_grads(False)
        grads, logp, stats, start = [], None, None, 0
        for n in direction_chunks(model, method, dtype, D):
            lp, dlp, st, na, nr = solve_batch_loglik(model, y0, pk, contact, t1, save_ts, obs, comp,
                                                     dparams=dps[:, start:start + n].contiguous(),
                                                     dy0=None if dys is None else dys[:, start:start + n].contiguous(),
                                                     increments=increments, floor=floor, **kw)
            logp = lp if logp is None else logp
            stats = (st, na, nr)
            grads.append(dlp)
            start += n
        G = grads[0] if len(grads) == 1 else torch.cat(grads, dim=1)      # [B, D]
        ctx.save_for_backward(G)
        ctx.in_dtype = leaf.dtype
        ctx.mark_non_differentiable(*stats)
        return (logp,) + stats

    @staticmethod
    def backward(ctx, g, *_unused):
        (G,) = ctx.saved_tensors
        if g is None:
            return (None,) * 13
        return ((g.unsqueeze(-1) * G).to(ctx.in_dtype),) + (None,) * 12


class _DiffSolve(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, model, y0, contact, t1, save_ts, kw):
        dtype = kw.get("dtype", torch.float32)
        method = kw.get("method", "tsit5")
        B, P = params.shape             # P counts the state columns too when `params` is [params | y0] (`_joined`)
        ctx.set_materialize_grads(False)          # no zero tensors for the integer outputs' "gradients"
        jac, res, start = [], None, 0
        for n in direction_chunks(model, method, dtype, P):
            pk, y0_n, dps, dys = _split_joined(model, params.detach(), _identity_seeds(B, P, start, n, dtype, params.device), y0, dtype)
            res = solve_batch(model, y0_n, pk, contact, t1, save_ts, dparams=dps.contiguous(),
                              dy0=None if dys is None else dys.contiguous(), **kw)
            jac.append(res.dys)
            start += n
        J = jac[0] if len(jac) == 1 else torch.cat(jac, dim=2)     # [B, n_save, P, D_saved]
        ctx.save_for_backward(J)
        ctx.in_dtype = params.dtype
        ctx.mark_non_differentiable(res.status, res.n_accept, res.n_reject)
        ctx.meta = (res.saved, res.sizes)
        return res.ys, res.status, res.n_accept, res.n_reject

    @staticmethod
    def backward(ctx, g_ys, *_unused):
        (J,) = ctx.saved_tensors
        if g_ys is None:
            return None, None, None, None, None, None, None
        g = torch.einsum("btd,btpd->bp", g_ys.to(J.dtype), J)
        return g.to(ctx.in_dtype), None, None, None, None, None, None


def solve_batch_diff(model, y0, params: torch.Tensor, contact, t1, save_ts, **kw):
    """Like ``engine.solve_batch`` but differentiable with respect to ``params`` ([B, P] tensor)."""
    from ..engine import BatchResult, save_mask_bytes

    seen = dict(plain=True, model=model, y0=y0, params=params, contact=contact, t1=t1, save_ts=save_ts, kw=dict(kw))
    params = _joined(params, y0)          # [params | y0] when the initial state carries a graph of its own
    y0 = y0.detach() if isinstance(y0, torch.Tensor) else y0
    leaf = _rowwise_leaf(params)
    if leaf is not None:      # D latent coordinates < P parameters: D tangent directions are enough
        seeds = _latent_seeds(params, leaf, kw.get("dtype", torch.float32))
        ys, status, n_acc, n_rej = _DiffSolveLatent.apply(leaf, params.detach(), seeds, model, y0, contact, t1, save_ts, kw)
    else:
        ys, status, n_acc, n_rej = _DiffSolve.apply(params, model, y0, contact, t1, save_ts, kw)
    _, saved, sizes = save_mask_bytes(model, kw.get("save_mask"))
    res = BatchResult(ys, status, n_acc, n_rej, saved, sizes)
    from . import folded

    # only inside `folded.recording()` (structure discovery of a sampler's potential: a model that scores the saved rows in
    # torch, like the reference's own inference example, may be the solve's fused likelihood written out)
    folded.note(dict(seen, result=res))
    return res


class _DiffLogLik(torch.autograd.Function):
    """Poisson observation log-likelihood of the solve, fused into the tangent kernel
    (``dyn_solve_batch_loglik``): value [B] and, for backward, d value / d params [B, P]."""

    @staticmethod
    def forward(ctx, params, model, y0, contact, t1, save_ts, obs, comp, increments, floor, kw):
        dtype = kw.get("dtype", torch.float32)
        method = kw.get("method", "tsit5")
        B, P = params.shape
        ctx.set_materialize_grads(False)
        grads, logp, stats, start = [], None, None, 0
        for n in direction_chunks(model, method, dtype, P):
            pk, y0_n, dps, dys = _split_joined(model, params.detach(), _identity_seeds(B, P, start, n, dtype, params.device), y0, dtype)
            lp, dlp, st, na, nr = solve_batch_loglik(model, y0_n, pk, contact, t1, save_ts, obs, comp, dparams=dps.contiguous(),
                                                     dy0=None if dys is None else dys.contiguous(),
                                                     increments=increments, floor=floor, **kw)
            logp = lp if logp is None else logp
            stats = (st, na, nr)
            grads.append(dlp)
            start += n
        G = grads[0] if len(grads) == 1 else torch.cat(grads, dim=1)      # [B, P]
        ctx.save_for_backward(G)
        ctx.in_dtype = params.dtype
        ctx.mark_non_differentiable(*stats)
        return (logp,) + stats

    @staticmethod
    def backward(ctx, g, *_unused):
        (G,) = ctx.saved_tensors
        if g is None:
            return (None,) * 11
        return ((g.unsqueeze(-1) * G).to(ctx.in_dtype),) + (None,) * 10


def solve_loglik_diff(model, y0, params: torch.Tensor, contact, t1, save_ts, obs, obs_compartment: int, *,
                      increments: bool = True, floor: float = 1e-6, **kw):
    """``(logp [B], status, n_accept, n_reject)``, differentiable with respect to ``params``."""
    from . import folded

    seen = dict(model=model, y0=y0, params=params, contact=contact, t1=t1, save_ts=save_ts, obs=obs, comp=int(obs_compartment),
                increments=bool(increments), floor=float(floor), kw=dict(kw))
    params = _joined(params, y0)
    y0 = y0.detach() if isinstance(y0, torch.Tensor) else y0
    leaf = _rowwise_leaf(params)
    if leaf is not None:
        seeds = _latent_seeds(params, leaf, kw.get("dtype", torch.float32))
        out = _DiffLogLikLatent.apply(leaf, params.detach(), seeds, model, y0, contact, t1, save_ts, obs,
                                      int(obs_compartment), bool(increments), float(floor), kw)
    else:
        out = _DiffLogLik.apply(params, model, y0, contact, t1, save_ts, obs, int(obs_compartment), bool(increments),
                                float(floor), kw)
    folded.note(dict(seen, result=out[0]))      # only inside `folded.recording()` (structure discovery of a sampler's potential)
    return out
