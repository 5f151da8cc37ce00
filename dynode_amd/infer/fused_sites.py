"""Latent sites of a model as one kernel launch (``dyn_latent_sites``, csrc/latent_kernel.hip).

`Potential.log_joint` evaluates, for every chain, the bijection of each scalar latent site, its
log prior and the log-Jacobian -- op by op that is ~50 tiny launches per site and gradient.  When
every latent site has constant parameters from the supported families (the
reference's priors, examples/sir_infer_parameters.py:47-58: affine-transformed Beta, truncated
normal; plus Normal and Uniform), the whole block is one launch with analytic derivatives.  Sites
outside that set keep the generic torch path (`distributions.py`), which stays the definition
the kernel is tested against (tests/test_gpu_infer.py).
"""

from __future__ import annotations

import ctypes
import math
from typing import Optional

import torch

from .. import _abi
from . import distributions as D


def _scalar(t, shape=(), elem=None) -> Optional[float]:
    """A distribution parameter as a python float: the parameter itself for a scalar site, element `elem` (row-major) of it
    broadcast to the site's batch shape for a tensor-valued one.  None if it is not a constant."""
    if isinstance(t, torch.Tensor):
        if t.requires_grad:
            return None
        if elem is not None:
            try:
                return float(torch.broadcast_to(t, shape).reshape(-1)[elem])
            except RuntimeError:
                return None
        return float(t) if t.numel() == 1 else None
    return float(t)


def describe(dist, elem=None) -> Optional[_abi.SiteDescC]:
    """``dyn_site_desc`` of a distribution object -- of element `elem` of a distribution with a batch shape (element-wise
    independent: each element is a scalar site of its own) --, or None if it is outside the fused families."""
    shape = tuple(dist.batch_shape)
    if (elem is None) != (shape == ()):
        return None
    sc = lambda t: _scalar(t, shape, elem)  # noqa: E731
    aff_loc, aff_scale = 0.0, 1.0
    base = dist
    if isinstance(dist, D.TransformedDistribution):
        base = dist.base
        for t in dist.transforms:
            if not isinstance(t, D.AffineTransform):
                return None
            loc, scale = sc(t.loc), sc(t.scale)
            if loc is None or scale is None:
                return None
            aff_loc, aff_scale = loc + scale * aff_loc, scale * aff_scale
    d = _abi.SiteDescC()
    d.aff_loc, d.aff_scale = aff_loc, aff_scale
    d.lo, d.hi = (float(v) for v in dist.support)        # (the bijection's interval: one per site, as distributions.biject_to has it)
    d.base_lo, d.base_hi = -math.inf, math.inf
    if type(base) is D.Normal:
        vals = (sc(base.loc), sc(base.scale))
        d.dist = _abi.DIST_NORMAL
    elif type(base) is D.Uniform:
        vals = (sc(base.low), sc(base.high))
        if elem is not None and (float(base.low.min()) != float(base.low.max()) or float(base.high.min()) != float(base.high.max())):
            return None     # (element-wise different bounds under the site's one bijection interval: the generic path's business)
        d.dist = _abi.DIST_UNIFORM
    elif type(base) is D.Beta:
        a, b = sc(base.a), sc(base.b)
        vals = (a, b, None if a is None or b is None else math.lgamma(a) + math.lgamma(b) - math.lgamma(a + b))
        d.dist = _abi.DIST_BETA
    elif type(base) is D.TruncatedNormal:
        vals = (sc(base.loc), sc(base.scale), sc(base._logz))
        lo, hi = sc(base.low), sc(base.high)
        if lo is None or hi is None:
            return None
        d.base_lo, d.base_hi = lo, hi
        d.dist = _abi.DIST_TRUNCNORMAL
    else:
        return None
    if any(v is None for v in vals) or aff_scale == 0.0 or not d.lo < d.hi:
        return None
    for i, v in enumerate(vals):
        d.p[i] = v
    return d


def build_table(dists) -> Optional[tuple]:
    """(ctypes array of descriptors, n) for the latent sites -- one descriptor per unconstrained coordinate, a tensor-valued
    site contributing one per element in row-major order --, or None if any site is not fusable."""
    descs = []
    for dist in dists:
        shape = tuple(dist.batch_shape)
        if shape == ():
            descs.append(describe(dist))
        else:
            k = 1
            for v in shape:
                k *= int(v)
            if k > _abi.MAX_SITES:
                return None
            descs.extend(describe(dist, i) for i in range(k))
    if not descs or len(descs) > _abi.MAX_SITES or any(d is None for d in descs):
        return None
    return (_abi.SiteDescC * len(descs))(*descs), len(descs)


class LatentSites(torch.autograd.Function):
    """z [C, n] -> (x [C, n] constrained values, lp [C] = sum_i log prior(x_i) + log|dx_i/dz_i|)."""

    @staticmethod
    def forward(ctx, z, table):
        arr, n = table
        if not z.is_cuda or z.dtype != torch.float64 or z.dim() != 2 or z.shape[1] != n:
            raise ValueError("LatentSites expects a float64 device tensor [chains, sites]")
        zc = z.detach().contiguous()
        C = zc.shape[0]
        x, dx, dlp = torch.empty_like(zc), torch.empty_like(zc), torch.empty_like(zc)
        lp = torch.empty(C, dtype=torch.float64, device=z.device)
        rc = _abi.lib().dyn_latent_sites(arr, n, C, zc.data_ptr(), x.data_ptr(), lp.data_ptr(), dx.data_ptr(),
                                         dlp.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc:
            raise RuntimeError(f"dyn_latent_sites: {_abi.ERR_NAMES.get(rc, rc)}")
        ctx.save_for_backward(dx, dlp)
        return x, lp

    @staticmethod
    def backward(ctx, gx, glp):
        dx, dlp = ctx.saved_tensors
        g = None
        if gx is not None:
            g = gx * dx
        if glp is not None:
            g = glp.unsqueeze(-1) * dlp if g is None else torch.addcmul(g, glp.unsqueeze(-1), dlp)
        return g, None
