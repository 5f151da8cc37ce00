"""Latent sites of a model as one kernel launch (``dyn_latent_sites``, csrc/latent_kernel.hip).

`Potential.log_joint` evaluates, for every chain, the bijection of each scalar latent site, its
log prior and the log-Jacobian -- op by op that is ~50 tiny launches per site and gradient.  When
every latent site is a scalar with constant parameters from the supported families (the
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


def _scalar(t) -> Optional[float]:
    if isinstance(t, torch.Tensor):
        if t.numel() != 1 or t.requires_grad:
            return None
        return float(t)
    return float(t)


def describe(dist) -> Optional[_abi.SiteDescC]:
    """``dyn_site_desc`` of a distribution object, or None if it is outside the fused families."""
    aff_loc, aff_scale = 0.0, 1.0
    base = dist
    if isinstance(dist, D.TransformedDistribution):
        base = dist.base
        for t in dist.transforms:
            if not isinstance(t, D.AffineTransform):
                return None
            aff_loc, aff_scale = t.loc + t.scale * aff_loc, t.scale * aff_scale
    d = _abi.SiteDescC()
    d.aff_loc, d.aff_scale = aff_loc, aff_scale
    d.lo, d.hi = (float(v) for v in dist.support)
    d.base_lo, d.base_hi = -math.inf, math.inf
    if type(base) is D.Normal:
        vals = (_scalar(base.loc), _scalar(base.scale))
        d.dist = _abi.DIST_NORMAL
    elif type(base) is D.Uniform:
        vals = (_scalar(base.low), _scalar(base.high))
        d.dist = _abi.DIST_UNIFORM
    elif type(base) is D.Beta:
        a, b = _scalar(base.a), _scalar(base.b)
        vals = (a, b, None if a is None or b is None else math.lgamma(a) + math.lgamma(b) - math.lgamma(a + b))
        d.dist = _abi.DIST_BETA
    elif type(base) is D.TruncatedNormal:
        vals = (_scalar(base.loc), _scalar(base.scale), _scalar(base._logz))
        d.base_lo, d.base_hi = float(base.low), float(base.high)
        d.dist = _abi.DIST_TRUNCNORMAL
    else:
        return None
    if any(v is None for v in vals) or aff_scale == 0.0 or not d.lo < d.hi:
        return None
    for i, v in enumerate(vals):
        d.p[i] = v
    return d


def build_table(dists) -> Optional[tuple]:
    """(ctypes array of descriptors, n) for the latent sites, or None if any site is not fusable."""
    descs = [describe(d) for d in dists]
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
