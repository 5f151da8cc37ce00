"""Gradient-free posterior sampling on the batched solve: the affine-invariant ensemble sampler.

NUTS needs the gradient-solve (tangent kernels).  Members of the kernel family without tangent planes -- the SEIP
model -- are still solved in large batches, which is exactly what an ensemble sampler consumes: every iteration
scores half of the walkers in ONE batched solve.  Algorithm: the stretch move of Goodman & Weare (2010) with the
parallel half-ensemble update of Foreman-Mackey et al. (2013, "emcee"): walker k of one half proposes
``z' = z_j + Z (z_k - z_j)`` with ``z_j`` drawn from the other half and ``Z ~ g(z) ~ 1 / sqrt(z)`` on ``[1/a, a]``,
accepted with probability ``min(1, Z^(D-1) p(z') / p(z_k))``.  The moves are made in the unconstrained
coordinates of the model's latent sites (the same ones NUTS uses).  Not part of the reference (which samples with
numpyro's NUTS only); selected with ``MCMCProcess(..., mcmc_kwargs={"sampler": "ensemble"})``.
"""

from __future__ import annotations

from typing import Callable, Optional

import torch

from .nuts import NUTSResult


class EnsembleSampler:
    def __init__(self, log_density: Callable[[torch.Tensor], torch.Tensor], stretch: float = 2.0, seed: int = 0):
        self.log_density = log_density
        self.a = float(stretch)
        self.seed = int(seed)
        self.evals = 0

    def _score(self, z: torch.Tensor) -> torch.Tensor:
        self.evals += 1
        with torch.no_grad():
            lp = self.log_density(z)
        return torch.where(torch.isfinite(lp), lp, torch.full_like(lp, -float("inf")))   # failed solves are rejected

    def run(self, z0: torch.Tensor, num_warmup: int, num_samples: int, thin: int = 1,
            progress: Optional[Callable[[int, bool], None]] = None, **_unused) -> NUTSResult:
        """``z0`` [walkers, D]: the walkers ARE the chains.  Returns the per-walker histories in NUTSResult's
        layout (accept_prob holds 0 / 1 per move, num_steps = 1: one density evaluation per walker and move)."""
        C, D = z0.shape
        if C % 2 or C < 2 * D + 2:
            raise ValueError(f"the ensemble sampler needs an even number of walkers, at least 2 D + 2 = {2 * D + 2} "
                             f"(got num_chains = {C} for {D} latent coordinates)")
        gen = torch.Generator(device=z0.device)
        gen.manual_seed(self.seed)
        z = z0.clone().double()
        if float(z.std(dim=0).min()) == 0.0:          # e.g. init_to_median: all walkers on one point -- spread them
            z = z + 0.1 * torch.randn(z.shape, generator=gen, device=z.device, dtype=z.dtype)
        lp = self._score(z)
        for _ in range(20):                            # walkers that start where the model fails are re-drawn near good ones
            bad = ~torch.isfinite(lp)
            if not bool(bad.any()):
                break
            good = torch.nonzero(~bad).reshape(-1)
            if good.numel() == 0:
                raise RuntimeError("no walker starts at a point of finite density")
            pick = good[torch.randint(good.numel(), (int(bad.sum()),), generator=gen, device=z.device)]
            z[bad] = z[pick] + 0.05 * torch.randn((int(bad.sum()), D), generator=gen, device=z.device, dtype=z.dtype)
            lp = self._score(z)
        half = C // 2
        halves = (torch.arange(0, half, device=z.device), torch.arange(half, C, device=z.device))
        total = num_warmup + num_samples * thin
        samples = torch.empty((C, num_samples, D), dtype=z.dtype, device=z.device)
        accepted = torch.zeros((C, num_samples), dtype=z.dtype, device=z.device)
        a = self.a
        for it in range(total):
            moved = torch.zeros(C, dtype=z.dtype, device=z.device)
            for mine, other in (halves, halves[::-1]):
                u = torch.rand(half, generator=gen, device=z.device, dtype=z.dtype)
                zz = ((a - 1.0) * u + 1.0) ** 2 / a                     # inverse CDF of g(z) ~ 1 / sqrt(z) on [1/a, a]
                partner = other[torch.randint(half, (half,), generator=gen, device=z.device)]
                prop = z[partner] + zz[:, None] * (z[mine] - z[partner])
                lp_prop = self._score(prop)
                log_ratio = (D - 1) * torch.log(zz) + lp_prop - lp[mine]
                take = torch.log(torch.rand(half, generator=gen, device=z.device, dtype=z.dtype)) < log_ratio
                z[mine] = torch.where(take[:, None], prop, z[mine])
                lp[mine] = torch.where(take, lp_prop, lp[mine])
                moved[mine] = take.to(z.dtype)
            if it >= num_warmup and (it - num_warmup) % thin == thin - 1:
                n = (it - num_warmup) // thin
                samples[:, n] = z
                accepted[:, n] = moved
            if progress is not None:
                progress(it, it < num_warmup)
        eye = torch.eye(D, dtype=z.dtype, device=z.device).expand(C, D, D).clone()
        return NUTSResult(samples=samples, accept_prob=accepted, num_steps=torch.ones((C, num_samples), dtype=torch.int32, device=z.device),
                          diverging=torch.zeros((C, num_samples), dtype=torch.bool, device=z.device),
                          step_size=torch.full((C,), a, dtype=z.dtype, device=z.device), inverse_mass=eye,
                          potential_evals=self.evals)
