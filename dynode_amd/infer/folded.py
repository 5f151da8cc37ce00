"""The sampler's potential in three launches: sites + parameter map, tangent solve with the likelihood, combine.

A numpyro model such as the reference's ``model(config, tf, obs_data)`` (examples/sir_infer_parameters.py:21-39) is, between
its sample sites and ``diffeqsolve``, a small program: ``get_odeparams`` turns the sampled ``r0`` / ``infectious_period`` into
``beta = r0 / T``, ``gamma = 1 / T`` (examples/sir.py:87-92 and the rest of SURVEY row A6), packs them, and the likelihood
adds a constant.  XLA fuses that program into the solve; run op by op it is ~26 launches of 2-5 us around an 80 us solve
(`profiles/r01_fused_nuts_iteration.md`).  Every member of the reference's ``get_odeparams`` family is a MONOMIAL map of the
site values, ``p_j = c_j * prod_i x_i ** e_ji``.  `discover` recognises that structure numerically -- one batched evaluation
of the model on a handful of probe rows, a log-linear fit, and an exact check on the rows the fit did not need -- and then
the potential and its gradient are

    dyn_latent_param_map   z -> x, log prior + log|dx/dz| (+ derivative), parameter rows, seeds d params / d z
    dyn_solve_batch_loglik parameters + seeds -> log-likelihood and its derivative along z
    dyn_potential_combine  u = -(lp + ll + offset), g = -(dlp + dll)

written straight into the sampler kernel's input buffers (`KernelNUTS`: 4 launches per iteration with `dyn_nuts_advance`).
A model that does not have the structure (a second solve, a parameter that is a sum of sites, a sampled initial state, a
likelihood outside the solve, sites outside the fused families) is NOT folded: `discover` returns None and the sampler keeps
the general torch-autograd potential -- the same GPU kernels, more launches.

What is verified, and what is not.  The structure is FITTED on 2 n + 6 probe rows around the centre (z ~ 0.8 N(0, 1)) and then
CHECKED on held-out rows it did not see, which reach into the tails (every coordinate at +-3 and +-5.5, and rows drawn
with three times the spread): the parameter map must be the fitted monomial there to 1e-9 and the folded potential and
gradient must equal the general ones wherever those are finite.  A model whose parameter map is piecewise (a clamp or
``where`` on a rate, a floor on a period) with the break inside that range is therefore refused.  A break farther out than any
held-out row cannot be seen up front; for that, `KernelNUTS` calls `FoldedPotential.verify` at the chains' CURRENT positions
a few times during warm-up and `MCMCProcess` restarts with the general potential if they ever disagree (`FoldMismatch`).
"""

from __future__ import annotations

import contextlib
import ctypes
import os
import sys
import threading
from typing import Optional

import torch

from .. import _abi

_STATE = threading.local()


class FoldMismatch(RuntimeError):
    """The folded potential disagreed with the model's own log joint at positions the sampler visited."""


@contextlib.contextmanager
def recording():
    """Collect the arguments (and result) of every ``autodiff.solve_loglik_diff`` call made inside the block."""
    calls: list = []
    prev = getattr(_STATE, "calls", None)
    _STATE.calls = calls
    try:
        yield calls
    finally:
        _STATE.calls = prev


def note(call: dict) -> None:
    calls = getattr(_STATE, "calls", None)
    if calls is not None:
        calls.append(call)


def _fit_monomials(x: torch.Tensor, params: torch.Tensor):
    """(coef [P], expo [P, n]) with params[:, j] == coef[j] * prod_i x[:, i] ** expo[j, i] on EVERY probe row, or None."""
    R, n = x.shape
    P = params.shape[1]
    usable = [i for i in range(n) if bool((x[:, i] > 0).all())]          # a site that can be <= 0 cannot carry a power
    A = torch.cat([torch.ones(R, 1, dtype=torch.float64), torch.log(x[:, usable])], dim=1) if usable else torch.ones(R, 1, dtype=torch.float64)
    coef, expo = torch.zeros(P, dtype=torch.float64), torch.zeros(P, n, dtype=torch.float64)
    for j in range(P):
        v = params[:, j]
        if bool((v == v[0]).all()):                 # a constant (also zero or negative ones)
            coef[j] = v[0]
            continue
        if not (bool((v > 0).all()) or bool((v < 0).all())):
            return None
        sign = 1.0 if float(v[0]) > 0 else -1.0
        sol = torch.linalg.lstsq(A, torch.log(v.abs())[:, None]).solution[:, 0]
        e = sol[1:]
        snapped = torch.round(e * 2.0) / 2.0        # the family's exponents are +-1; allow halves, keep anything else as fitted
        e = torch.where((e - snapped).abs() < 1e-8, snapped, e)
        logc = (torch.log(v.abs()) - torch.log(x[:, usable]) @ e).mean()
        coef[j] = sign * torch.exp(logc)
        for col, i in enumerate(usable):
            expo[j, i] = e[col]
    logx = torch.log(x.clamp_min(1e-300))           # columns with a zero exponent contribute 0 whatever their sign
    pred = coef[None, :] * torch.exp((expo[None] * logx[:, None, :]).sum(-1))
    if not bool(((pred - params).abs() <= 1e-11 * params.abs() + 1e-300).all()):
        return None
    return coef, expo


class FoldedPotential:
    """``into(z, u_out, g_out)``: potential and gradient of ``pot`` for a fixed number of chains, three launches."""

    def __init__(self, pot, call: dict, coef: torch.Tensor, expo: torch.Tensor, offset: float):
        self.pot, self.call, self.offset = pot, call, float(offset)
        dev = pot.device
        self.coef, self.expo = coef.to(dev).contiguous(), expo.to(dev).contiguous()
        self.n, self.P = pot.dim, int(coef.shape[0])
        self.dtype = call["kw"].get("dtype", torch.float32)
        self._buf: dict = {}
        self._split: dict = {}
        self._lean: set = set()

    def _ensure_lean(self, n_dir: int) -> None:
        """A sampler is about to take thousands of gradient-solves on this model: have the lean twin of its tangent kernel
        built (`jit.ensure_lean_twin`; a no-op for the shapes compiled in and where no lean instance applies)."""
        if n_dir in self._lean:
            return
        self._lean.add(n_dir)
        kw = self.call["kw"]
        jumps = kw.get("jump_ts")
        if self.call["model"].family != 0 or (jumps is not None and len(jumps) > 0) or float(kw.get("constant_dt", 0.0) or 0.0) > 0.0:
            return
        from .. import jit

        try:
            jit.ensure_lean_twin(self.call["model"], self.dtype, kw.get("method", "tsit5"), n_dir, self.call["comp"], self.call["increments"])
        except Exception as err:  # pragma: no cover - a failed build must not stop the run: the general instance is there
            print(f"[dynode_amd] lean twin not built ({type(err).__name__}: {str(err)[:120]}); the general tangent instance runs", file=sys.stderr, flush=True)

    SPLIT_MAX_ROWS = int(os.environ.get("DYNODE_FOLD_SPLIT_ROWS", "2048"))   # chains x directions up to which the GPU is far from full (cfg 4: 128 x 2)

    def split_directions(self, C: int) -> bool:
        """One tangent direction per trajectory (n C rows, n_dir = 1) instead of n in one: with a few hundred chains the solve
        is the serial latency of ONE trajectory, and a trajectory that carries one tangent plane instead of n does a third less
        (n = 2) on that path -- measured on cfg 4's gradient-solve: 89.6 -> 73.8 us, identical bits."""
        hit = self._split.get(C)
        if hit is None:
            from .autodiff import _supported_nd

            method = self.call["kw"].get("method", "tsit5")
            hit = self._split[C] = (self.n > 1 and C * self.n <= self.SPLIT_MAX_ROWS and self.call["model"].family == 0
                                    and _supported_nd(self.call["model"], method, self.dtype, 1))
        return hit

    def rows_per_chain(self, C: int) -> int:
        """Trajectories a chain occupies in the gradient-solve's batch: 1 with every direction in one row, else -- one direction
        per row -- the number of sites rounded up to a power of two (the rows beyond the sites are padding: the chain's
        parameters, zero seeds).  A wave holds a power of two of trajectories, so whole chains then fall into waves: what the
        one-launch sampler iteration needs (`dyn_solver_opts::nuts_tail`), six sites included."""
        if not self.split_directions(C):
            return 1
        rows = 1
        while rows < self.n:
            rows <<= 1
        return rows

    def _split_arg(self, C: int) -> int:
        """``split_directions`` of the C ABI: 0, or the rows per chain."""
        return self.rows_per_chain(C) if self.split_directions(C) else 0

    def _buffers(self, C: int):
        b = self._buf.get(C)
        if b is None:
            dev, f64 = self.pot.device, torch.float64
            split = self.split_directions(C)
            rows = C * self.rows_per_chain(C)                                    # split: every chain once per direction (+ padding)
            b = self._buf[C] = dict(x=torch.empty((C, self.n), dtype=f64, device=dev), lp=torch.empty(C, dtype=f64, device=dev),
                                    dlp=torch.empty((C, self.n), dtype=f64, device=dev),
                                    params=torch.empty((rows, self.P), dtype=self.dtype, device=dev),
                                    seeds=torch.empty((rows, 1 if split else self.n, self.P), dtype=self.dtype, device=dev))
        return b

    def into(self, z: torch.Tensor, u_out: torch.Tensor, g_out: torch.Tensor) -> None:
        C = z.shape[0]
        for t in (z, u_out, g_out):
            if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
                raise ValueError("FoldedPotential.into needs contiguous float64 device tensors")
        if tuple(z.shape) != (C, self.n) or tuple(g_out.shape) != (C, self.n) or tuple(u_out.shape) != (C,):
            raise ValueError(f"shapes: z, g [C, {self.n}], u [C]")
        lp, dlp, ll, dll, stride = self.parts(z)
        rc = _abi.lib().dyn_potential_combine(C, self.n, lp.data_ptr(), dlp.data_ptr(), ll.data_ptr(), dll.data_ptr(), self.offset,
                                              self._split_arg(C), u_out.data_ptr(), g_out.data_ptr(),
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc:
            raise RuntimeError(f"dyn_potential_combine: {_abi.ERR_NAMES.get(rc, rc)}")

    def parts(self, z: torch.Tensor):
        """The two launches in front of the combine: ``(lp [C], dlp [C, n], ll, dll [C, rows >= n], ll_stride)`` with
        ``u = -(lp + ll[::ll_stride] + offset)``, ``g = -(dlp + dll[:, :n])`` -- `dyn_nuts_advance` forms these itself
        (`dyn_nuts_state.pot_*`; ``pot_dll_stride = dll.shape[1]``)."""
        self.map_now(z)
        return self.solve_current(z.shape[0])

    def map_now(self, z: torch.Tensor) -> None:
        """`dyn_latent_param_map` at ``z``: fills this potential's buffers (x, lp, dlp, parameter rows, seeds)."""
        from ..engine import _DTYPES

        C = z.shape[0]
        if not (z.is_cuda and z.dtype == torch.float64 and z.is_contiguous() and tuple(z.shape) == (C, self.n)):
            raise ValueError(f"FoldedPotential needs a contiguous float64 device tensor [C, {self.n}]")
        b = self._buffers(C)
        arr, n = self.pot.site_table
        rc = _abi.lib().dyn_latent_param_map(arr, n, C, z.data_ptr(), b["x"].data_ptr(), b["lp"].data_ptr(), b["dlp"].data_ptr(),
                                             self.P, self.coef.data_ptr(), self.expo.data_ptr(), _DTYPES[self.dtype],
                                             self._split_arg(C), b["params"].data_ptr(), b["seeds"].data_ptr(),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc:
            raise RuntimeError(f"dyn_latent_param_map: {_abi.ERR_NAMES.get(rc, rc)}")

    def advance_mapped(self, st, C: int) -> int:
        """`dyn_nuts_advance_mapped`: the sampler kernel also fills this potential's buffers for the position it hands out,
        so the NEXT gradient is `solve_current` alone (a sampler iteration: two launches)."""
        from ..engine import _DTYPES

        b = self._buffers(C)
        arr, n = self.pot.site_table
        return _abi.lib().dyn_nuts_advance_mapped(ctypes.byref(st), arr, n, self.P, self.coef.data_ptr(), self.expo.data_ptr(),
                                                  _DTYPES[self.dtype], self._split_arg(C), b["x"].data_ptr(),
                                                  b["lp"].data_ptr(), b["dlp"].data_ptr(), b["params"].data_ptr(),
                                                  b["seeds"].data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    def pack_tail(self, st, C: int):
        """`dyn_nuts_tail_pack`: the (host) blob that lets the gradient-solve run the sampler's side of an iteration itself
        (``solve_current(C, nuts_tail=...)``: ONE launch per iteration).  ``st`` must carry this potential's prior-side
        buffers in ``pot_lp`` / ``pot_dlp``.  Returns the blob (keep it alive for as long as launches use it), or None when
        the library refuses (more than eight sampled sites)."""
        from ..engine import _DTYPES

        from .. import jit

        L = _abi.lib()
        b = self._buffers(C)
        arr, n = self.pot.site_table
        # the gradient-solve's instance needs its twin with the sampler behind it: built in for the inference examples' shapes,
        # built on first use for any other float32 shape (dynode_amd/jit.py)
        n_dir = 1 if self.split_directions(C) else self.n
        if n > _abi.NUTS_REG_DIM or not jit.ensure_fused_twin(self.call["model"], self.dtype, self.call["kw"].get("method", "tsit5"), n_dir):
            return None
        blob = ctypes.create_string_buffer(int(L.dyn_nuts_tail_size()))
        rc = L.dyn_nuts_tail_pack(ctypes.byref(st), arr, n, self.P, self.coef.data_ptr(), self.expo.data_ptr(),
                                  _DTYPES[self.dtype], self._split_arg(C), b["x"].data_ptr(), b["lp"].data_ptr(),
                                  b["dlp"].data_ptr(), b["params"].data_ptr(), b["seeds"].data_ptr(), blob)
        if rc == -7:
            return None
        if rc:
            raise RuntimeError(f"dyn_nuts_tail_pack: {_abi.ERR_NAMES.get(rc, rc)}")
        return blob

    def solve_current(self, C: int, nuts_tail=None):
        """The gradient-solve on the parameter rows and seeds the buffers hold: ``(lp, dlp, ll, dll, ll_stride)``.
        ``nuts_tail`` (a `pack_tail` blob): the same launch also advances the sampler (`engine.solve_batch_loglik`)."""
        from ..engine import solve_batch_loglik
        from .autodiff import direction_chunks

        b, c = self._buffers(C), self.call
        split = self.split_directions(C)
        method = c["kw"].get("method", "tsit5")
        self._ensure_lean(1 if split else min(self.n, 2))
        if split:      # rows C trajectories with one direction each (rows >= n per chain): ll [rows C], dll [rows C, 1]
            ll, dll, *_ = solve_batch_loglik(c["model"], c["y0"], b["params"], c["contact"], c["t1"], c["save_ts"], c["obs"],
                                             c["comp"], dparams=b["seeds"], increments=c["increments"], floor=c["floor"],
                                             nuts_tail=None if nuts_tail is None else ctypes.addressof(nuts_tail), **c["kw"])
        else:
            ll, grads, start = None, [], 0
            chunks = list(direction_chunks(c["model"], method, self.dtype, self.n))
            if nuts_tail is not None and len(chunks) != 1:
                from ..engine import SolveError

                raise SolveError(-7, "nuts_tail: the directions of a chain take more than one launch")
            for nd in chunks:
                seeds = b["seeds"] if nd == self.n else b["seeds"][:, start:start + nd].contiguous()
                lp_, dlp_, *_ = solve_batch_loglik(c["model"], c["y0"], b["params"], c["contact"], c["t1"], c["save_ts"], c["obs"],
                                                   c["comp"], dparams=seeds, increments=c["increments"], floor=c["floor"],
                                                   nuts_tail=None if nuts_tail is None else ctypes.addressof(nuts_tail), **c["kw"])
                ll = lp_ if ll is None else ll
                grads.append(dlp_)
                start += nd
            dll = grads[0] if len(grads) == 1 else torch.cat(grads, dim=1)
        rows = self.rows_per_chain(C)
        return b["lp"], b["dlp"], ll, dll.reshape(C, rows if split else self.n), rows

    def deviation(self, z: torch.Tensor):
        """(max |du| / (1 + |u|), max |dg| / (1 + |g|), rows finite in the general potential but not in the folded one) at the
        rows of ``z`` where the general potential is finite.  Evaluates into scratch buffers of its own batch size, so the
        sampler's buffers (which hold the map of the position it is about to evaluate) are not touched -- callers pass a
        batch size the sampler does not use (`verify` pads by one row)."""
        z = z.detach().to(torch.float64).contiguous()
        u_ref, g_ref = self.pot.potential_and_grad(z)
        u, g = self(z)
        ok = torch.isfinite(u_ref) & torch.isfinite(g_ref).all(-1)
        lost = int((ok & ~(torch.isfinite(u) & torch.isfinite(g).all(-1))).sum())
        if not bool(ok.any()):
            return 0.0, 0.0, lost
        du = ((u - u_ref).abs() / (1.0 + u_ref.abs()))[ok]
        dg = ((g - g_ref).abs() / (1.0 + g_ref.abs()))[ok]
        fin = torch.isfinite(du) & torch.isfinite(dg).all(-1)
        return float(du[fin].max()) if bool(fin.any()) else 0.0, float(dg[fin].max()) if bool(fin.any()) else 0.0, lost

    # (a likelihood the model wrote out in torch, `_written_out_likelihood`, is the same float32 solve scored by other
    # arithmetic: cfg 4 in the bulk of its posterior reads 2.3e-5 of the potential and 3e-2 on gradients of tens to hundreds)
    written_out = False

    def verify(self, z: torch.Tensor, rtol_u: Optional[float] = None, rtol_g: Optional[float] = None) -> bool:
        """Folded == general potential (value and gradient) at the rows of ``z`` where the general one is finite."""
        rtol_u = (2e-4 if self.written_out else 1e-5) if rtol_u is None else rtol_u
        rtol_g = (2e-3 if self.written_out else 1e-4) if rtol_g is None else rtol_g
        z = torch.cat([z.detach().to(torch.float64), z.detach().to(torch.float64)[:1]], dim=0)   # C + 1 rows: buffers of their own
        du, dg, lost = self.deviation(z)
        return du <= rtol_u and dg <= rtol_g and lost == 0

    def __call__(self, z: torch.Tensor):
        """The ``potential_and_grad`` signature (fresh outputs), for the samplers that are not `KernelNUTS`."""
        z = z.detach().to(torch.float64).contiguous()
        u = torch.empty(z.shape[0], dtype=torch.float64, device=z.device)
        g = torch.empty_like(z)
        self.into(z, u, g)
        return u, g


def _written_out_likelihood(pot, z: torch.Tensor, plain: dict, why):
    """A model that solves WITHOUT ``observe=`` and scores the saved rows in torch -- the reference's own inference example:
    ``incidence = clip(diff(solution.ys[r]), 1e-6); sample("obs", Poisson(incidence), obs=data)``
    (examples/sir_infer_parameters.py:21-39) -- may be the solve's fused likelihood written out.  Recognised by VALUE, not
    by reading the model: the one observed site must be a Poisson whose rate equals, element for element, one saved
    compartment's values or increments, floored at a constant.  Returns the equivalent ``solve_loglik_diff`` record (with its
    ``result`` on these rows), or None.  `discover` then holds the folded potential against the model's own log joint on
    probe and held-out rows like any other, and the sampler re-checks it at the chains' positions."""
    from ..engine import solve_batch_loglik
    from . import distributions as dist

    zz = z.detach().clone().requires_grad_(True)
    zz._dynode_rowwise = True
    with recording() as again:
        _, tr = pot.log_joint(zz)
    again = [c for c in again if c.get("plain")]
    observed = [s for s in tr.sites.values() if s["type"] == "sample" and s["is_observed"]]
    if len(again) != 1 or len(observed) != 1 or any(s["type"] == "factor" for s in tr.sites.values()):
        return why("the log joint is not ONE solve scored by ONE observed site")
    site, res, R = observed[0], again[0]["result"], z.shape[0]
    rate, obs = getattr(site["fn"], "rate", None), site["value"]
    if not isinstance(site["fn"], dist.Poisson) or not isinstance(rate, torch.Tensor) or not isinstance(obs, torch.Tensor):
        return why("the observed site is not a Poisson with a tensor rate")
    kw = {k: v for k, v in plain["kw"].items() if k != "save_mask"}      # (as simulate() hands them to the fused-likelihood solve)
    if plain["kw"].get("save_mask") is not None or plain["model"].family != 0 or rate.shape[0] != R or obs.dim() != rate.dim() - 1:
        return why("sub-saved rows, a model family without the fused likelihood, or observations that carry a batch axis")
    ys, n_save = res.ys.detach(), res.ys.shape[1]
    rate_flat, pos = rate.detach().reshape(R, rate.shape[1], -1), 0
    for comp, size in enumerate(res.sizes):
        block = ys[:, :, pos:pos + size]
        pos += size
        for increments in (True, False):
            cand = block[:, 1:] - block[:, :-1] if increments else block
            if tuple(cand.shape) != tuple(rate_flat.shape):
                continue
            same = rate_flat == cand
            floored = ~same
            if bool(floored.any()):
                floor = float(rate_flat[floored].max())
                if not bool((rate_flat[floored] == floor).all()) or not bool((cand[floored] < floor).all()):
                    continue
            else:   # nothing floored on these rows: the clamp's bound from the autograd node, if the rate came out of one
                fn = rate.grad_fn
                floor = float(getattr(fn, "_saved_min", 0.0) or 0.0) if fn is not None and type(fn).__name__.startswith("Clamp") else 0.0
            if not floor > 0.0:     # (the kernel's likelihood needs one: an unfloored non-positive rate is NaN in the model)
                return why("the observed site's rate is a saved compartment's values or increments WITHOUT a positive floor")
            obs_t = obs.detach().to(device=ys.device, dtype=ys.dtype).reshape(-1).contiguous()
            if obs_t.numel() != rate_flat.shape[1] * size:
                continue
            call = dict(model=plain["model"], y0=plain["y0"], params=plain["params"], contact=plain["contact"], t1=plain["t1"],
                        save_ts=plain["save_ts"], obs=obs_t, comp=comp, increments=increments, floor=floor, kw=kw)
            p = plain["params"].detach()
            ll = solve_batch_loglik(call["model"], call["y0"], p.to(ys.dtype) if p.dtype != ys.dtype else p, call["contact"], call["t1"],
                                    call["save_ts"], obs_t, comp, dparams=torch.zeros((R, 1, p.shape[1]), dtype=ys.dtype, device=ys.device),
                                    increments=increments, floor=floor, **kw)[0]
            call["result"], call["written_out"] = ll, site["name"] if "name" in site else True
            return call
    return why("the observed site's rate is not a saved compartment's values or increments, floored at a constant")


def discover(pot, seed: int = 0, verbose: bool = False) -> Optional[FoldedPotential]:
    """`FoldedPotential` of ``pot`` (an `inference.Potential`) if the model has the structure, else None."""
    why = lambda msg: (print(f"[dynode_amd] potential not folded: {msg}") if verbose else None)  # noqa: E731
    if pot.site_table is None or torch.device(pot.device).type != "cuda":
        return why("latent sites outside the fused families, or no GPU")
    n = pot.dim
    R = 2 * n + 6
    gen = torch.Generator().manual_seed(1234 + seed)
    z = (0.8 * torch.randn((R, n), generator=gen, dtype=torch.float64)).to(pot.device)
    with recording() as seen:
        u_ref, g_ref = pot.potential_and_grad(z)
    calls, plain = [c for c in seen if not c.get("plain")], [c for c in seen if c.get("plain")]
    written_out = False
    if not calls and len(plain) == 1 and bool(torch.isfinite(u_ref).all()):
        call = _written_out_likelihood(pot, z, plain[0], why)
        if call is None:
            return None
        calls, written_out = [call], True
    if len(calls) != 1:
        return why(f"{len(calls)} fused-likelihood solves per evaluation (need exactly one)")
    call = calls[0]
    params, y0 = call.pop("params"), call["y0"]
    if isinstance(y0, torch.Tensor) and (y0.requires_grad or y0.dim() != 1):
        return why("the initial state depends on the batch or on a latent site")
    if not (isinstance(params, torch.Tensor) and params.dim() == 2 and params.shape[0] == R):
        return why("parameter matrix is not [chains, P]")
    if not bool(torch.isfinite(u_ref).all()):
        return why("non-finite potential on the probe rows")
    from .fused_sites import LatentSites

    x, lp_sites = LatentSites.apply(z, pot.site_table)
    fit = _fit_monomials(x.cpu(), params.detach().to(torch.float64).cpu())
    if fit is None:
        return why("the parameter rows are not monomials of the site values")
    # everything else in the log joint must be a constant: log joint - log prior - log-likelihood of the solve
    # (a written-out likelihood is scored by torch in float64 from the saved float32 rows, the fused one in the kernel with
    # float32 logarithms and increments formed in registers: the same sum to about 1e-6 of its size -- measured 1.2e-6 on cfg 4 --, not to the last bit)
    rest = (-u_ref) - lp_sites - call.pop("result").detach()
    offset = float(rest.mean())
    call.pop("written_out", None)
    if not bool(((rest - offset).abs() <= (1e-5 if written_out else 1e-9) * (1.0 + u_ref.abs())).all()):
        return why("the log joint has terms besides the priors and the solve's likelihood "
                   f"(largest deviation from a constant {float(((rest - offset).abs() / (1.0 + u_ref.abs())).max()):.3g} of the potential)")
    folded = FoldedPotential(pot, call, fit[0], fit[1], offset)
    folded.written_out = written_out
    u, g = folded(z)
    # same kernels on (up to the last bit of a float64 product) the same parameter rows: agreement far inside the
    # sampler's own float32 solve noise, and a loud refusal otherwise
    scale_u, scale_g = 1.0 + u_ref.abs(), 1.0 + g_ref.abs()
    if not (bool(((u - u_ref).abs() <= 1e-5 * scale_u).all()) and bool(((g - g_ref).abs() <= 1e-4 * scale_g).all())):
        return why(f"folded and general potential differ on the probe rows (max |du| {float((u - u_ref).abs().max()):.3g}, "
                   f"max |dg| {float((g - g_ref).abs().max()):.3g})")
    # ---- held-out rows the fit never saw, reaching into the tails: a piecewise parameter map (clamp / where / floor on a
    # rate) whose break lies inside |z| <= 5.5 fails here
    z2 = held_out_rows(n, gen).to(pot.device)
    with recording() as seen2:
        pot.potential_and_grad(z2)
    calls2 = [c for c in seen2 if bool(c.get("plain")) == written_out]
    if len(calls2) != 1 or len(seen2) != 1:
        return why(f"{len(seen2)} solves on the held-out rows (control flow depends on the position)")
    p2 = calls2[0]["params"].detach().to(torch.float64).cpu()
    x2, _ = LatentSites.apply(z2, pot.site_table)
    x2 = x2.cpu()
    pred = fit[0][None, :] * torch.exp((fit[1][None] * torch.log(x2.clamp_min(1e-300))[:, None, :]).sum(-1))
    okp = torch.isfinite(p2) & torch.isfinite(pred)
    if not bool((((pred - p2).abs() <= 1e-9 * p2.abs() + 1e-300) | ~okp).all()) or not bool((okp == torch.isfinite(p2)).all()):
        worst = float(((pred - p2).abs() / (p2.abs() + 1e-300))[okp].max()) if bool(okp.any()) else float("nan")
        return why(f"the parameter rows leave the fitted monomials away from the centre (held-out rows, worst relative deviation {worst:.3g})")
    # (same kernels on the same parameter rows, so this can only fail through the map above; in the tails the potential is
    # 1e3-1e5 with float32 solves behind it, hence the looser bars than on the probe rows)
    # (a written-out likelihood: out there a day's increment of the scored compartment sinks to the rounding noise of the
    # float32 rows it is the difference of -- 6e-5 on counts of a thousand -- and torch's diff of the saved rows and the
    # kernel's increment carry different noise into log(rate): cfg 4 reads 9e-4 / 1.2e-2 at |z| = 5.5, where the density is
    # exp(-1e5); the probe rows above hold the two to 1e-5 / 1e-4, and the sampler re-checks at the chains' positions)
    du, dg, lost = folded.deviation(z2)
    if du > (5e-3 if written_out else 1e-4) or dg > (5e-2 if written_out else 1e-2) or lost:
        return why(f"folded and general potential differ on the held-out rows (relative: value {du:.3g}, gradient {dg:.3g}; {lost} rows lost)")
    return folded


def held_out_rows(n: int, gen: torch.Generator) -> torch.Tensor:
    """Validation positions of `discover`: every unconstrained coordinate at +-3 and +-5.5 with the others at 0, and
    2 n + 6 rows drawn with three times the spread of the probe rows, clipped to |z| <= 6."""
    axis = torch.zeros((4 * n, n), dtype=torch.float64)
    for i in range(n):
        for q, v in enumerate((-5.5, -3.0, 3.0, 5.5)):
            axis[4 * i + q, i] = v
    wide = (2.4 * torch.randn((2 * n + 6, n), generator=gen, dtype=torch.float64)).clamp(-6.0, 6.0)
    return torch.cat([axis, wide], dim=0)
