"""Batched No-U-Turn samplers: every chain of a GPU in ONE batched gradient-solve per leapfrog.

What numpyro does under ``MCMC(NUTS(model, dense_mass=True, max_tree_depth, init_to_median))``
(reference src/dynode/infer/inference.py:149-163), re-designed for a GPU whose potential
evaluation is ONE fused gradient-solve launch for every chain at once: the tree is built with the
iterative (checkpointed) algorithm of Hoffman-Gelman / Phan et al. so that at any moment all
chains are at the same tree depth and leaf index -- checkpoint bookkeeping is host arithmetic,
per-chain differences (direction, U-turn, divergence, proposal choice) are masks.  Warm-up is
Stan's windowed scheme: dual-averaging step size (target accept 0.8), dense mass matrix from
Welford covariance with shrinkage, per chain (chains adapt independently, as in numpyro).

Chains shard embarrassingly over GPUs (one process per GPU, `dynode_amd.sharding`); nothing here
communicates.
"""

from __future__ import annotations

import math
import os
import sys
from dataclasses import dataclass
from typing import Callable, Optional

import torch


def _bdot(a, b):
    return (a * b).sum(-1)


def _mv(m, v):
    return torch.einsum("cij,cj->ci", m, v)


@dataclass
class NUTSResult:
    samples: torch.Tensor       # [C, num_samples, D] unconstrained
    accept_prob: torch.Tensor   # [C, num_samples]
    num_steps: torch.Tensor     # [C, num_samples] leapfrogs per transition
    diverging: torch.Tensor     # [C, num_samples]
    step_size: torch.Tensor     # [C]
    inverse_mass: torch.Tensor  # [C, D, D]
    potential_evals: int        # batched gradient-solves issued


class _DualAveraging:
    """Nesterov dual averaging on log step size, per chain (Stan / numpyro defaults)."""

    def __init__(self, step_size, t0=10.0, kappa=0.75, gamma=0.05):
        self.t0, self.kappa, self.gamma = t0, kappa, gamma
        self.restart(step_size)

    def restart(self, step_size):
        self.mu = torch.log(10.0 * step_size)
        self.x_bar = torch.zeros_like(step_size)
        self.g_bar = torch.zeros_like(step_size)
        self.t = 0

    def update(self, g):
        self.t += 1
        w = 1.0 / (self.t + self.t0)
        self.g_bar = (1 - w) * self.g_bar + w * g
        x = self.mu - math.sqrt(self.t) / self.gamma * self.g_bar
        wx = self.t ** (-self.kappa)
        self.x_bar = (1 - wx) * self.x_bar + wx * x
        return torch.exp(x), torch.exp(self.x_bar)


class _Welford:
    def __init__(self, C, D, device):
        self.n = 0
        self.mean = torch.zeros((C, D), dtype=torch.float64, device=device)
        self.m2 = torch.zeros((C, D, D), dtype=torch.float64, device=device)

    def update(self, x):
        self.n += 1
        d = x - self.mean
        self.mean = self.mean + d / self.n
        self.m2 = self.m2 + torch.einsum("ci,cj->cij", d, x - self.mean)

    def covariance(self):
        cov = self.m2 / (self.n - 1)
        scaled = (self.n / (self.n + 5.0)) * cov
        eye = torch.eye(cov.shape[-1], dtype=cov.dtype, device=cov.device)
        return scaled + 1e-3 * (5.0 / (self.n + 5.0)) * eye


def _adaptation_windows(num_warmup: int, init_buffer: int = 75):
    """Stan's schedule: 75-step initial buffer, doubling 25-step windows, 50-step final buffer."""
    if num_warmup < 20:
        return []
    init, term, base = init_buffer, 50, 25
    if init + base + term > num_warmup:
        init, term = int(0.15 * num_warmup), int(0.1 * num_warmup)
        base = num_warmup - init - term
    ends, start, size = [], init, base
    while start + size <= num_warmup - term:
        nxt = start + size
        if nxt + 2 * size > num_warmup - term:   # stretch the last window to the final buffer
            nxt = num_warmup - term
        ends.append((start, nxt))
        start, size = nxt, size * 2
    return ends


class LockstepNUTS:
    """Reference implementation: all chains build their trees in lockstep (simple, but every
    transition costs as many gradient-solves as its DEEPEST chain needs).  Kept for cross-checks.

    NUTS for C independent chains over a D-dimensional unconstrained space.

    ``potential_and_grad(z [C, D]) -> (U [C], dU/dz [C, D])`` evaluates every chain in one call.
    """

    def __init__(self, potential_and_grad: Callable, max_tree_depth: int = 10, target_accept: float = 0.8,
                 max_delta_energy: float = 1000.0, seed: int = 0):
        self.pg = potential_and_grad
        self.max_depth = int(max_tree_depth)
        self.target = float(target_accept)
        self.max_de = float(max_delta_energy)
        self.seed = int(seed)
        self.evals = 0

    # -------------------------------------------------------------- pieces
    def _eval(self, z):
        self.evals += 1
        u, g = self.pg(z)
        bad = ~torch.isfinite(u) | ~torch.isfinite(g).all(-1)
        u = torch.where(bad, torch.full_like(u, math.inf), u)
        g = torch.where(bad[:, None], torch.zeros_like(g), g)
        return u, g

    def _kinetic(self, imm, r):
        return 0.5 * _bdot(r, _mv(imm, r))

    def _leapfrog(self, z, r, g, eps, imm):
        r = r - 0.5 * eps[:, None] * g
        z = z + eps[:, None] * _mv(imm, r)
        u, g = self._eval(z)
        r = r - 0.5 * eps[:, None] * g
        return z, r, u, g

    @staticmethod
    def _is_turning(imm, r_left, r_right, r_sum):
        rs = r_sum - 0.5 * (r_left + r_right)
        return (_bdot(_mv(imm, r_left), rs) <= 0) | (_bdot(_mv(imm, r_right), rs) <= 0)

    def _find_reasonable_step_size(self, z, u, g, imm, mm_sqrt, eps, gen):
        """Double/halve each chain's step until the one-step accept probability crosses 0.8."""
        C = z.shape[0]
        r = _mv(mm_sqrt, torch.randn(z.shape, dtype=z.dtype, device=z.device, generator=gen))
        e0 = u + self._kinetic(imm, r)
        direction = torch.zeros(C, dtype=z.dtype, device=z.device)
        active = torch.ones(C, dtype=torch.bool, device=z.device)
        for _ in range(50):
            _, r1, u1, _ = self._leapfrog(z, r, g, eps, imm)
            de = u1 + self._kinetic(imm, r1) - e0
            de = torch.where(torch.isnan(de), torch.full_like(de, math.inf), de)
            d = torch.where(-de > math.log(self.target), torch.ones_like(de), -torch.ones_like(de))
            first = direction == 0
            direction = torch.where(first, d, direction)
            active = active & (d == direction) & (eps > 1e-10) & (eps < 1e7)
            if not bool(active.any()):
                break
            eps = torch.where(active, eps * (2.0 ** direction), eps)
        return eps

    def _initial_kernel(self, z, u, g, init_step_size, step_size, inverse_mass, gen):
        """(eps, imm, mm_sqrt) a run starts from.  Default: identity metric and a line search from ``init_step_size``
        (numpyro's ``find_reasonable_step_size``).  ``step_size`` ([C] or a float) / ``inverse_mass`` ([C, D, D] or [D, D])
        start the run from a GIVEN kernel instead -- numpyro's ``NUTS(step_size=..., inverse_mass_matrix=...)``; with
        ``num_warmup = 0`` nothing adapts and every transition uses exactly that kernel (the stationarity checks of
        tests/test_gpu_infer.py start chains at exact posterior draws and need the kernel held fixed)."""
        C, D = z.shape
        dt, dev = z.dtype, z.device
        eye = torch.eye(D, dtype=dt, device=dev).expand(C, D, D).contiguous()
        imm, mm_sqrt = eye.clone(), eye.clone()
        if inverse_mass is not None:
            imm = torch.as_tensor(inverse_mass, dtype=dt, device=dev).expand(C, D, D).contiguous()
            mm_sqrt = torch.linalg.cholesky(torch.linalg.inv(imm)).contiguous()
        if step_size is not None:
            eps = torch.as_tensor(step_size, dtype=dt, device=dev).expand(C).contiguous().clone()
        else:
            eps = self._find_reasonable_step_size(z, u, g, imm, mm_sqrt,
                                                  torch.full((C,), float(init_step_size), dtype=dt, device=dev), gen)
        return eps, imm, mm_sqrt

    # -------------------------------------------------------------- one transition for all chains
    def _transition(self, z, u, g, eps, imm, mm_sqrt, gen):
        C, D = z.shape
        dev, dt = z.device, z.dtype
        r0 = _mv(mm_sqrt, torch.randn((C, D), dtype=dt, device=dev, generator=gen))
        e0 = u + self._kinetic(imm, r0)
        zl, rl, gl = z.clone(), r0.clone(), g.clone()
        zr, rr, gr = z.clone(), r0.clone(), g.clone()
        zp, up, gp = z.clone(), u.clone(), g.clone()
        weight = torch.zeros(C, dtype=dt, device=dev)          # log sum of exp(-delta energy)
        r_sum = r0.clone()
        turning = torch.zeros(C, dtype=torch.bool, device=dev)
        diverging = torch.zeros(C, dtype=torch.bool, device=dev)
        sum_acc = torch.zeros(C, dtype=dt, device=dev)
        n_prop = torch.zeros(C, dtype=torch.long, device=dev)

        for depth in range(self.max_depth):
            alive = ~turning & ~diverging
            if not bool(alive.any()):
                break
            right = torch.rand(C, device=dev, generator=gen) < 0.5
            sgn = torch.where(right, torch.ones(C, dtype=dt, device=dev), -torch.ones(C, dtype=dt, device=dev))
            # ---- subtree of 2^depth leaves grown from the chosen end (iterative, checkpointed)
            zc = torch.where(right[:, None], zr, zl)
            rc = torch.where(right[:, None], rr, rl)
            gc = torch.where(right[:, None], gr, gl)
            s_first_r = None
            s_zp, s_up, s_gp = zc.clone(), up.clone(), gc.clone()
            s_weight = torch.full((C,), -math.inf, dtype=dt, device=dev)
            s_rsum = torch.zeros((C, D), dtype=dt, device=dev)
            s_turn = torch.zeros(C, dtype=torch.bool, device=dev)
            s_div = torch.zeros(C, dtype=torch.bool, device=dev)
            s_acc = torch.zeros(C, dtype=dt, device=dev)
            s_n = torch.zeros(C, dtype=torch.long, device=dev)
            r_ck = torch.zeros((self.max_depth, C, D), dtype=dt, device=dev)
            rs_ck = torch.zeros((self.max_depth, C, D), dtype=dt, device=dev)
            for leaf in range(2 ** depth):
                grow = alive & ~s_turn & ~s_div
                if not bool(grow.any()):
                    break
                zn, rn, un, gn = self._leapfrog(zc, rc, gc, eps * sgn, imm)
                de = un + self._kinetic(imm, rn) - e0
                de = torch.where(torch.isnan(de), torch.full_like(de, math.inf), de)
                lw = -de
                div = de > self.max_de
                acc = torch.exp(torch.clamp(-de, max=0.0))
                if leaf == 0:
                    s_first_r = rn.clone()
                # uniform (multinomial) choice inside the subtree
                new_w = torch.logaddexp(s_weight, lw)
                take = torch.rand(C, device=dev, generator=gen).to(dt) < torch.exp(lw - new_w)
                upd = grow
                sel = upd & take
                s_zp = torch.where(sel[:, None], zn, s_zp)
                s_up = torch.where(sel, un, s_up)
                s_gp = torch.where(sel[:, None], gn, s_gp)
                s_weight = torch.where(upd, new_w, s_weight)
                s_rsum = torch.where(upd[:, None], s_rsum + rn, s_rsum)
                s_div = s_div | (upd & div)
                s_acc = torch.where(upd, s_acc + acc, s_acc)
                s_n = s_n + upd.long()
                zc = torch.where(upd[:, None], zn, zc)
                rc = torch.where(upd[:, None], rn, rc)
                gc = torch.where(upd[:, None], gn, gc)
                # U-turn checks of every balanced sub-subtree that this leaf completes
                idx_max = bin(leaf >> 1).count("1")
                trailing = 0
                while (leaf >> trailing) & 1:
                    trailing += 1
                idx_min = idx_max - trailing + 1
                if leaf % 2 == 0:
                    r_ck[idx_max] = torch.where(upd[:, None], rn, r_ck[idx_max])
                    rs_ck[idx_max] = torch.where(upd[:, None], s_rsum, rs_ck[idx_max])
                else:
                    t = torch.zeros(C, dtype=torch.bool, device=dev)
                    for i in range(idx_max, idx_min - 1, -1):
                        sub = s_rsum - rs_ck[i] + r_ck[i]
                        t = t | self._is_turning(imm, r_ck[i], rn, sub)
                    s_turn = s_turn | (upd & t)
            # ---- merge the subtree into the trajectory (biased progressive sampling)
            grew = alive & (s_n > 0)
            ok = grew & ~s_turn & ~s_div
            move = ok & (torch.rand(C, device=dev, generator=gen).to(dt) < torch.exp(torch.clamp(s_weight - weight, max=0.0)))
            zp = torch.where(move[:, None], s_zp, zp)
            up = torch.where(move, s_up, up)
            gp = torch.where(move[:, None], s_gp, gp)
            ext_r = grew & right
            ext_l = grew & ~right
            zr = torch.where(ext_r[:, None], zc, zr); rr = torch.where(ext_r[:, None], rc, rr); gr = torch.where(ext_r[:, None], gc, gr)
            zl = torch.where(ext_l[:, None], zc, zl); rl = torch.where(ext_l[:, None], rc, rl); gl = torch.where(ext_l[:, None], gc, gl)
            weight = torch.where(grew, torch.logaddexp(weight, s_weight), weight)
            r_sum = torch.where(grew[:, None], r_sum + s_rsum, r_sum)
            sum_acc = sum_acc + torch.where(grew, s_acc, torch.zeros_like(s_acc))
            n_prop = n_prop + torch.where(grew, s_n, torch.zeros_like(s_n))
            diverging = diverging | (grew & s_div)
            turning = turning | (grew & (s_turn | self._is_turning(imm, rl, rr, r_sum)))
        accept = sum_acc / n_prop.clamp_min(1).to(dt)
        return zp, up, gp, accept, n_prop, diverging

    # -------------------------------------------------------------- driver
    def run(self, z0: torch.Tensor, num_warmup: int, num_samples: int, init_step_size: float = 1.0,
            progress: Optional[Callable] = None) -> NUTSResult:
        z = z0.clone().to(torch.float64)
        C, D = z.shape
        dev = z.device
        gen = torch.Generator(device=dev).manual_seed(self.seed)
        eye = torch.eye(D, dtype=torch.float64, device=dev).expand(C, D, D).contiguous()
        imm, mm_sqrt = eye.clone(), eye.clone()
        u, g = self._eval(z)
        eps = torch.full((C,), float(init_step_size), dtype=torch.float64, device=dev)
        eps = self._find_reasonable_step_size(z, u, g, imm, mm_sqrt, eps, gen)
        da = _DualAveraging(eps)
        windows = _adaptation_windows(num_warmup)
        win_i, welford = 0, _Welford(C, D, dev)
        out_z = torch.empty((C, num_samples, D), dtype=torch.float64, device=dev)
        out_acc = torch.empty((C, num_samples), dtype=torch.float64, device=dev)
        out_n = torch.empty((C, num_samples), dtype=torch.long, device=dev)
        out_div = torch.empty((C, num_samples), dtype=torch.bool, device=dev)
        eps_avg = eps.clone()
        for it in range(num_warmup + num_samples):
            warm = it < num_warmup
            z, u, g, acc, n_prop, div = self._transition(z, u, g, eps, imm, mm_sqrt, gen)
            if warm:
                eps, eps_avg = da.update(self.target - acc)
                if win_i < len(windows) and windows[win_i][0] <= it < windows[win_i][1]:
                    welford.update(z)
                    if it + 1 == windows[win_i][1]:
                        imm = welford.covariance()
                        mm_sqrt = torch.linalg.cholesky(torch.linalg.inv(imm))
                        welford = _Welford(C, D, dev)
                        win_i += 1
                        eps = self._find_reasonable_step_size(z, u, g, imm, mm_sqrt, eps_avg, gen)
                        da.restart(eps)
                if it + 1 == num_warmup:
                    eps = eps_avg
            else:
                j = it - num_warmup
                out_z[:, j], out_acc[:, j], out_n[:, j], out_div[:, j] = z, acc, n_prop, div
            if progress is not None:
                progress(it, warm)
        return NUTSResult(out_z, out_acc, out_n, out_div, eps, imm, self.evals)


class BatchedNUTS(LockstepNUTS):
    """Asynchronous batched NUTS: every batched gradient-solve advances EVERY chain by one leapfrog.

    Each chain carries its own position in the algorithm (transition count, tree depth, leaf index,
    direction, checkpoints, adaptation state) as rows of state tensors, so no chain ever waits for
    another chain's deeper tree: the number of gradient-solves is the per-chain leapfrog count
    (about 5-6 per transition for the cfg-4 posterior) instead of the per-transition maximum over
    all chains (about 25 with 128 chains in lockstep).  Same transition kernel as `LockstepNUTS`
    (iterative checkpointed U-turn test, multinomial sampling, biased progressive merge, Stan
    warm-up); after a mass-matrix update the step size restarts dual averaging from its running
    average instead of a fresh line search.
    """

    def run(self, z0: torch.Tensor, num_warmup: int, num_samples: int, init_step_size: float = 1.0,
            progress: Optional[Callable] = None, step_size=None, inverse_mass=None) -> NUTSResult:
        C, D = z0.shape
        dev, dt = z0.device, torch.float64
        Dm = self.max_depth
        total = num_warmup + num_samples
        gen = torch.Generator(device=dev).manual_seed(self.seed)
        ar = torch.arange(C, device=dev)

        def rand():
            return torch.rand(C, device=dev, generator=gen).to(dt)

        z = z0.clone().to(dt)
        eye = torch.eye(D, dtype=dt, device=dev).expand(C, D, D).contiguous()
        u, g = self._eval(z)
        eps, imm, mm_sqrt = self._initial_kernel(z, u, g, init_step_size, step_size, inverse_mass, gen)
        eps_avg = eps.clone()
        # dual averaging state, per chain
        da_mu, da_xbar, da_gbar = torch.log(10.0 * eps), torch.zeros(C, dtype=dt, device=dev), torch.zeros(C, dtype=dt, device=dev)
        da_t = torch.zeros(C, dtype=dt, device=dev)
        # adaptation windows, per chain pointer
        windows = _adaptation_windows(num_warmup)
        w_start = torch.tensor([w[0] for w in windows] + [total + 1], device=dev)
        w_end = torch.tensor([w[1] for w in windows] + [total + 2], device=dev)
        wi = torch.zeros(C, dtype=torch.long, device=dev)
        wf_n = torch.zeros(C, dtype=dt, device=dev)
        wf_mean = torch.zeros((C, D), dtype=dt, device=dev)
        wf_m2 = torch.zeros((C, D, D), dtype=dt, device=dev)

        it = torch.zeros(C, dtype=torch.long, device=dev)
        out_z = torch.zeros((C, num_samples, D), dtype=dt, device=dev)
        out_acc = torch.zeros((C, num_samples), dtype=dt, device=dev)
        out_n = torch.zeros((C, num_samples), dtype=torch.long, device=dev)
        out_div = torch.zeros((C, num_samples), dtype=torch.bool, device=dev)

        def new_tree(z, u, g):
            r0 = _mv(mm_sqrt, torch.randn((C, D), dtype=dt, device=dev, generator=gen))
            return dict(e0=u + self._kinetic(imm, r0), zl=z.clone(), rl=r0.clone(), gl=g.clone(), zr=z.clone(),
                        rr=r0.clone(), gr=g.clone(), zp=z.clone(), up=u.clone(), gp=g.clone(),
                        weight=torch.zeros(C, dtype=dt, device=dev), r_sum=r0.clone(),
                        sum_acc=torch.zeros(C, dtype=dt, device=dev), n_prop=torch.zeros(C, dtype=torch.long, device=dev),
                        depth=torch.zeros(C, dtype=torch.long, device=dev))

        def new_subtree(T):
            right = rand() < 0.5
            sgn = torch.where(right, torch.ones(C, dtype=dt, device=dev), -torch.ones(C, dtype=dt, device=dev))
            return dict(right=right, sgn=sgn, leaf=torch.zeros(C, dtype=torch.long, device=dev),
                        zc=torch.where(right[:, None], T["zr"], T["zl"]), rc=torch.where(right[:, None], T["rr"], T["rl"]),
                        gc=torch.where(right[:, None], T["gr"], T["gl"]),
                        zp=T["zp"].clone(), up=T["up"].clone(), gp=T["gp"].clone(),
                        weight=torch.full((C,), -math.inf, dtype=dt, device=dev), rsum=torch.zeros((C, D), dtype=dt, device=dev),
                        turn=torch.zeros(C, dtype=torch.bool, device=dev), div=torch.zeros(C, dtype=torch.bool, device=dev),
                        acc=torch.zeros(C, dtype=dt, device=dev), n=torch.zeros(C, dtype=torch.long, device=dev))

        def merge(old: dict, new: dict, mask):
            for k_ in old:
                m = mask.reshape((C,) + (1,) * (old[k_].dim() - 1))
                old[k_] = torch.where(m, new[k_], old[k_])

        T = new_tree(z, u, g)
        Sb = new_subtree(T)
        r_ck = torch.zeros((Dm, C, D), dtype=dt, device=dev)
        rs_ck = torch.zeros((Dm, C, D), dtype=dt, device=dev)
        levels = torch.arange(Dm, device=dev)
        n_iter = 0
        while True:
            active = it < total
            if not bool(active.any()):
                break
            n_iter += 1
            # ---- one leapfrog for every chain, from its own subtree frontier
            zn, rn, un, gn = self._leapfrog(Sb["zc"], Sb["rc"], Sb["gc"], eps * Sb["sgn"], imm)
            de = un + self._kinetic(imm, rn) - T["e0"]
            de = torch.where(torch.isnan(de), torch.full_like(de, math.inf), de)
            lw, div = -de, de > self.max_de
            acc = torch.exp(torch.clamp(-de, max=0.0))
            new_w = torch.logaddexp(Sb["weight"], lw)
            sel = active & (rand() < torch.exp(lw - new_w))
            Sb["zp"] = torch.where(sel[:, None], zn, Sb["zp"]); Sb["up"] = torch.where(sel, un, Sb["up"])
            Sb["gp"] = torch.where(sel[:, None], gn, Sb["gp"])
            Sb["weight"] = torch.where(active, new_w, Sb["weight"])
            Sb["rsum"] = torch.where(active[:, None], Sb["rsum"] + rn, Sb["rsum"])
            Sb["div"] = Sb["div"] | (active & div)
            Sb["acc"] = Sb["acc"] + torch.where(active, acc, torch.zeros_like(acc))
            Sb["n"] = Sb["n"] + active.long()
            Sb["zc"] = torch.where(active[:, None], zn, Sb["zc"]); Sb["rc"] = torch.where(active[:, None], rn, Sb["rc"])
            Sb["gc"] = torch.where(active[:, None], gn, Sb["gc"])
            # ---- checkpointed U-turn test, leaf index per chain
            leaf = Sb["leaf"]
            idx_max = sum(((leaf >> (b + 1)) & 1) for b in range(Dm))
            ones, trailing = torch.ones_like(leaf), torch.zeros_like(leaf)
            for b in range(Dm):
                ones = ones & ((leaf >> b) & 1)
                trailing = trailing + ones
            idx_min = idx_max - trailing + 1
            even = (leaf & 1) == 0
            slot = idx_max.clamp(0, Dm - 1).view(1, C, 1).expand(1, C, D)
            wmask = (active & even)[:, None]
            r_ck.scatter_(0, slot, torch.where(wmask, rn, r_ck.gather(0, slot)[0]).unsqueeze(0))
            rs_ck.scatter_(0, slot, torch.where(wmask, Sb["rsum"], rs_ck.gather(0, slot)[0]).unsqueeze(0))
            in_range = (levels[:, None] <= idx_max[None, :]) & (levels[:, None] >= idx_min[None, :])     # [Dm, C]
            sub = Sb["rsum"][None] - rs_ck + r_ck                                                     # [Dm, C, D]
            rs = sub - 0.5 * (r_ck + rn[None])
            vl = torch.einsum("cij,lcj->lci", imm, r_ck)
            vr = _mv(imm, rn)
            turn_l = ((vl * rs).sum(-1) <= 0) | ((vr[None] * rs).sum(-1) <= 0)
            Sb["turn"] = Sb["turn"] | (active & ~even & (turn_l & in_range).any(0))
            Sb["leaf"] = leaf + active.long()
            # ---- subtree complete?
            sub_done = active & (Sb["turn"] | Sb["div"] | (Sb["leaf"] >= (1 << T["depth"])))
            ok = sub_done & ~Sb["turn"] & ~Sb["div"]
            move = ok & (rand() < torch.exp(torch.clamp(Sb["weight"] - T["weight"], max=0.0)))
            T["zp"] = torch.where(move[:, None], Sb["zp"], T["zp"]); T["up"] = torch.where(move, Sb["up"], T["up"])
            T["gp"] = torch.where(move[:, None], Sb["gp"], T["gp"])
            er, el = sub_done & Sb["right"], sub_done & ~Sb["right"]
            for end, m in (("r", er), ("l", el)):
                T["z" + end] = torch.where(m[:, None], Sb["zc"], T["z" + end])
                T["r" + end] = torch.where(m[:, None], Sb["rc"], T["r" + end])
                T["g" + end] = torch.where(m[:, None], Sb["gc"], T["g" + end])
            T["weight"] = torch.where(sub_done, torch.logaddexp(T["weight"], Sb["weight"]), T["weight"])
            T["r_sum"] = torch.where(sub_done[:, None], T["r_sum"] + Sb["rsum"], T["r_sum"])
            T["sum_acc"] = T["sum_acc"] + torch.where(sub_done, Sb["acc"], torch.zeros_like(acc))
            T["n_prop"] = T["n_prop"] + torch.where(sub_done, Sb["n"], torch.zeros_like(Sb["n"]))
            T["depth"] = T["depth"] + sub_done.long()
            stop = sub_done & (Sb["turn"] | Sb["div"] | self._is_turning(imm, T["rl"], T["rr"], T["r_sum"]) |
                               (T["depth"] >= Dm))
            # ---- transition complete: adapt, record, start the next one
            if bool(stop.any()):
                warm = it < num_warmup
                a_prob = T["sum_acc"] / T["n_prop"].clamp_min(1).to(dt)
                z = torch.where(stop[:, None], T["zp"], z); u = torch.where(stop, T["up"], u)
                g = torch.where(stop[:, None], T["gp"], g)
                # dual averaging (warm-up only)
                upd = stop & warm
                t1 = da_t + 1.0
                w = 1.0 / (t1 + 10.0)
                gbar = (1 - w) * da_gbar + w * (self.target - a_prob)
                x = da_mu - torch.sqrt(t1) / 0.05 * gbar
                wx = t1 ** (-0.75)
                xbar = (1 - wx) * da_xbar + wx * x
                da_t = torch.where(upd, t1, da_t); da_gbar = torch.where(upd, gbar, da_gbar)
                da_xbar = torch.where(upd, xbar, da_xbar)
                eps = torch.where(upd, torch.exp(x), eps)
                eps_avg = torch.where(upd, torch.exp(xbar), eps_avg)
                # mass matrix windows
                in_win = upd & (it >= w_start[wi]) & (it < w_end[wi])
                n1 = wf_n + 1.0
                d = z - wf_mean
                mean1 = wf_mean + d / n1[:, None]
                m21 = wf_m2 + torch.einsum("ci,cj->cij", d, z - mean1)
                wf_n = torch.where(in_win, n1, wf_n); wf_mean = torch.where(in_win[:, None], mean1, wf_mean)
                wf_m2 = torch.where(in_win[:, None, None], m21, wf_m2)
                close = in_win & (it + 1 == w_end[wi])
                if bool(close.any()):
                    nn = wf_n.clamp_min(2.0)
                    cov = wf_m2 / (nn - 1.0)[:, None, None]
                    reg = (nn / (nn + 5.0))[:, None, None] * cov + 1e-3 * (5.0 / (nn + 5.0))[:, None, None] * eye
                    imm = torch.where(close[:, None, None], reg, imm)
                    mm_sqrt = torch.where(close[:, None, None], torch.linalg.cholesky(torch.linalg.inv(imm)), mm_sqrt)
                    wf_n = torch.where(close, torch.zeros_like(wf_n), wf_n)
                    wf_mean = torch.where(close[:, None], torch.zeros_like(wf_mean), wf_mean)
                    wf_m2 = torch.where(close[:, None, None], torch.zeros_like(wf_m2), wf_m2)
                    wi = wi + close.long()
                    # restart dual averaging around the running average step size
                    eps = torch.where(close, eps_avg, eps)
                    da_mu = torch.where(close, torch.log(10.0 * eps_avg), da_mu)
                    da_t = torch.where(close, torch.zeros_like(da_t), da_t)
                    da_gbar = torch.where(close, torch.zeros_like(da_gbar), da_gbar)
                    da_xbar = torch.where(close, torch.zeros_like(da_xbar), da_xbar)
                eps = torch.where(upd & (it + 1 == num_warmup), eps_avg, eps)
                # record post-warm-up draws at each chain's own sample index
                rec = stop & ~warm
                j = (it - num_warmup).clamp(0, num_samples - 1)
                jz = j.view(C, 1, 1).expand(C, 1, D)
                out_z.scatter_(1, jz, torch.where(rec[:, None], z, out_z.gather(1, jz)[:, 0]).unsqueeze(1))
                j1 = j.view(C, 1)
                out_acc.scatter_(1, j1, torch.where(rec, a_prob, out_acc.gather(1, j1)[:, 0]).unsqueeze(1))
                out_n.scatter_(1, j1, torch.where(rec, T["n_prop"], out_n.gather(1, j1)[:, 0]).unsqueeze(1))
                out_div.scatter_(1, j1, torch.where(rec, Sb["div"], out_div.gather(1, j1)[:, 0]).unsqueeze(1))
                it = it + stop.long()
                merge(T, new_tree(z, u, g), stop)
                if progress is not None:
                    progress(int(it.min()) - 1, int(it.min()) <= num_warmup)
            # ---- next subtree for every chain that finished one (new transition or next doubling)
            if bool(sub_done.any()):
                merge(Sb, new_subtree(T), sub_done)
                fresh = sub_done[None, :, None]
                r_ck = torch.where(fresh, torch.zeros_like(r_ck), r_ck)
                rs_ck = torch.where(fresh, torch.zeros_like(rs_ck), rs_ck)
        return NUTSResult(out_z, out_acc, out_n, out_div, eps, imm, self.evals)


class GraphNUTS(LockstepNUTS):
    """`BatchedNUTS` with the whole iteration as ONE HIP-graph replay.

    The asynchronous sampler's iteration is a pure tensor program: state tensors in, state tensors
    out, every data-dependent decision a mask.  Written that way it can be captured once -- the
    potential (the user's model function, the fused gradient-solve kernel, autograd), the leapfrog,
    tree bookkeeping, adaptation and sample recording together -- and replayed; the host only draws
    a block of random numbers and checks for completion every `block` iterations.  Mass-matrix
    updates are computed in the graph and applied (with their Cholesky factor and the step-size
    restart) at the next block boundary, i.e. at most `block` iterations late, which is immaterial
    for warm-up.  On CPU tensors the same step function runs eagerly (used by the tests).
    """

    def __init__(self, *args, block: int = 16, use_graph: bool = True, **kw):
        super().__init__(*args, **kw)
        self.block, self.use_graph = int(block), bool(use_graph)

    # ---------------------------------------------------------------- one iteration, pure function
    def _step(self, S: dict, K: dict, ru, rn):
        """S: state tensors, K: constants, ru [3, C] uniforms, rn [C, D] normals -> new state dict."""
        C, D, Dm, dt = K["C"], K["D"], self.max_depth, torch.float64
        N = dict(S)  # new state (entries replaced below)
        active = S["it"] < K["total"]
        imm, mm_sqrt = S["imm"], S["mm_sqrt"]
        # ---- one leapfrog for every chain from its subtree frontier
        zn, rn_, un, gn = self._leapfrog(S["zc"], S["rc"], S["gc"], S["eps"] * S["sgn"], imm)
        de = un + self._kinetic(imm, rn_) - S["e0"]
        de = torch.where(torch.isnan(de), torch.full_like(de, math.inf), de)
        lw, div = -de, de > self.max_de
        acc = torch.exp(torch.clamp(-de, max=0.0))
        new_w = torch.logaddexp(S["s_weight"], lw)
        sel = active & (ru[0] < torch.exp(lw - new_w))
        s_zp = torch.where(sel[:, None], zn, S["s_zp"]); s_up = torch.where(sel, un, S["s_up"])
        s_gp = torch.where(sel[:, None], gn, S["s_gp"])
        s_weight = torch.where(active, new_w, S["s_weight"])
        s_rsum = torch.where(active[:, None], S["s_rsum"] + rn_, S["s_rsum"])
        s_div = S["s_div"] | (active & div)
        s_acc = S["s_acc"] + torch.where(active, acc, torch.zeros_like(acc))
        s_n = S["s_n"] + active.long()
        zc = torch.where(active[:, None], zn, S["zc"]); rc = torch.where(active[:, None], rn_, S["rc"])
        gc = torch.where(active[:, None], gn, S["gc"])
        # ---- checkpointed U-turn test, leaf index per chain
        leaf = S["leaf"]
        idx_max = torch.zeros_like(leaf)
        for b in range(Dm):
            idx_max = idx_max + ((leaf >> (b + 1)) & 1)
        ones, trailing = torch.ones_like(leaf), torch.zeros_like(leaf)
        for b in range(Dm):
            ones = ones & ((leaf >> b) & 1)
            trailing = trailing + ones
        idx_min = idx_max - trailing + 1
        even = (leaf & 1) == 0
        lv = K["levels"]
        wmask = (active & even)[None, :, None] & (lv[:, None, None] == idx_max[None, :, None])     # [Dm, C, 1]
        r_ck = torch.where(wmask, rn_[None], S["r_ck"])
        rs_ck = torch.where(wmask, s_rsum[None], S["rs_ck"])
        in_range = (lv[:, None] <= idx_max[None, :]) & (lv[:, None] >= idx_min[None, :])
        rs = (s_rsum[None] - rs_ck + r_ck) - 0.5 * (r_ck + rn_[None])
        vl = torch.einsum("cij,lcj->lci", imm, r_ck)
        vr = _mv(imm, rn_)
        turn_l = ((vl * rs).sum(-1) <= 0) | ((vr[None] * rs).sum(-1) <= 0)
        s_turn = S["s_turn"] | (active & ~even & (turn_l & in_range).any(0))
        leaf = leaf + active.long()
        # ---- subtree complete -> merge into the tree
        depth = S["depth"]
        sub_done = active & (s_turn | s_div | (leaf >= (torch.ones_like(depth) << depth)))
        ok = sub_done & ~s_turn & ~s_div
        move = ok & (ru[1] < torch.exp(torch.clamp(s_weight - S["weight"], max=0.0)))
        zp = torch.where(move[:, None], s_zp, S["zp"]); up = torch.where(move, s_up, S["up"])
        gp = torch.where(move[:, None], s_gp, S["gp"])
        er, el = (sub_done & S["right"])[:, None], (sub_done & ~S["right"])[:, None]
        zr, rr, gr = torch.where(er, zc, S["zr"]), torch.where(er, rc, S["rr"]), torch.where(er, gc, S["gr"])
        zl, rl, gl = torch.where(el, zc, S["zl"]), torch.where(el, rc, S["rl"]), torch.where(el, gc, S["gl"])
        weight = torch.where(sub_done, torch.logaddexp(S["weight"], s_weight), S["weight"])
        r_sum = torch.where(sub_done[:, None], S["r_sum"] + s_rsum, S["r_sum"])
        sum_acc = S["sum_acc"] + torch.where(sub_done, s_acc, torch.zeros_like(s_acc))
        n_prop = S["n_prop"] + torch.where(sub_done, s_n, torch.zeros_like(s_n))
        depth = depth + sub_done.long()
        stop = sub_done & (s_turn | s_div | self._is_turning(imm, rl, rr, r_sum) | (depth >= Dm))
        # ---- transition complete: adapt, record, next transition
        it = S["it"]
        warm = it < K["num_warmup"]
        a_prob = sum_acc / n_prop.clamp_min(1).to(dt)
        z = torch.where(stop[:, None], zp, S["z"]); u = torch.where(stop, up, S["u"]); g = torch.where(stop[:, None], gp, S["g"])
        upd = stop & warm
        t1 = S["da_t"] + 1.0
        w = 1.0 / (t1 + 10.0)
        gbar = (1 - w) * S["da_gbar"] + w * (self.target - a_prob)
        x = S["da_mu"] - torch.sqrt(t1) / 0.05 * gbar
        wx = t1 ** (-0.75)
        xbar = (1 - wx) * S["da_xbar"] + wx * x
        N["da_t"] = torch.where(upd, t1, S["da_t"]); N["da_gbar"] = torch.where(upd, gbar, S["da_gbar"])
        N["da_xbar"] = torch.where(upd, xbar, S["da_xbar"])
        eps = torch.where(upd, torch.exp(x), S["eps"])
        eps_avg = torch.where(upd, torch.exp(xbar), S["eps_avg"])
        wi = S["wi"]
        ws, we = K["w_start"][wi], K["w_end"][wi]
        in_win = upd & (it >= ws) & (it < we)
        n1 = S["wf_n"] + 1.0
        d = z - S["wf_mean"]
        mean1 = S["wf_mean"] + d / n1[:, None]
        m21 = S["wf_m2"] + torch.einsum("ci,cj->cij", d, z - mean1)
        wf_n = torch.where(in_win, n1, S["wf_n"]); wf_mean = torch.where(in_win[:, None], mean1, S["wf_mean"])
        wf_m2 = torch.where(in_win[:, None, None], m21, S["wf_m2"])
        close = in_win & (it + 1 == we)
        nn = wf_n.clamp_min(2.0)
        reg = (nn / (nn + 5.0))[:, None, None] * (wf_m2 / (nn - 1.0)[:, None, None]) + \
            1e-3 * (5.0 / (nn + 5.0))[:, None, None] * K["eye"]
        N["pend_cov"] = torch.where(close[:, None, None], reg, S["pend_cov"])
        N["need_mm"] = S["need_mm"] | close
        N["wf_n"] = torch.where(close, torch.zeros_like(wf_n), wf_n)
        N["wf_mean"] = torch.where(close[:, None], torch.zeros_like(wf_mean), wf_mean)
        N["wf_m2"] = torch.where(close[:, None, None], torch.zeros_like(wf_m2), wf_m2)
        N["wi"] = wi + close.long()
        N["eps"] = torch.where(upd & (it + 1 == K["num_warmup"]), eps_avg, eps)
        N["eps_avg"] = eps_avg
        rec = stop & ~warm
        j = (it - K["num_warmup"]).clamp(0, K["num_samples"] - 1)
        jz = j.view(C, 1, 1).expand(C, 1, D)
        N["out_z"] = S["out_z"].scatter(1, jz, torch.where(rec[:, None], z, S["out_z"].gather(1, jz)[:, 0]).unsqueeze(1))
        j1 = j.view(C, 1)
        N["out_acc"] = S["out_acc"].scatter(1, j1, torch.where(rec, a_prob, S["out_acc"].gather(1, j1)[:, 0]).unsqueeze(1))
        N["out_n"] = S["out_n"].scatter(1, j1, torch.where(rec, n_prop, S["out_n"].gather(1, j1)[:, 0]).unsqueeze(1))
        N["out_div"] = S["out_div"].scatter(1, j1, torch.where(rec, s_div, S["out_div"].gather(1, j1)[:, 0]).unsqueeze(1))
        N["it"] = it + stop.long()
        N["z"], N["u"], N["g"] = z, u, g
        # new tree for the chains that finished a transition
        r0 = _mv(mm_sqrt, rn)
        sm, sv = stop[:, None], stop
        N["e0"] = torch.where(sv, u + self._kinetic(imm, r0), S["e0"])
        N["zl"], N["rl"], N["gl"] = torch.where(sm, z, zl), torch.where(sm, r0, rl), torch.where(sm, g, gl)
        N["zr"], N["rr"], N["gr"] = torch.where(sm, z, zr), torch.where(sm, r0, rr), torch.where(sm, g, gr)
        N["zp"], N["up"], N["gp"] = torch.where(sm, z, zp), torch.where(sv, u, up), torch.where(sm, g, gp)
        N["weight"] = torch.where(sv, torch.zeros_like(weight), weight)
        N["r_sum"] = torch.where(sm, r0, r_sum)
        N["sum_acc"] = torch.where(sv, torch.zeros_like(sum_acc), sum_acc)
        N["n_prop"] = torch.where(sv, torch.zeros_like(n_prop), n_prop)
        N["depth"] = torch.where(sv, torch.zeros_like(depth), depth)
        # new subtree for the chains that finished one (next doubling or first of a new transition)
        right = torch.where(sub_done, ru[2] < 0.5, S["right"])
        N["right"] = right
        N["sgn"] = torch.where(right, torch.ones_like(S["sgn"]), -torch.ones_like(S["sgn"]))
        sd, sdv = sub_done[:, None], sub_done
        N["leaf"] = torch.where(sdv, torch.zeros_like(leaf), leaf)
        N["zc"] = torch.where(sd, torch.where(right[:, None], N["zr"], N["zl"]), zc)
        N["rc"] = torch.where(sd, torch.where(right[:, None], N["rr"], N["rl"]), rc)
        N["gc"] = torch.where(sd, torch.where(right[:, None], N["gr"], N["gl"]), gc)
        N["s_zp"] = torch.where(sd, N["zp"], s_zp); N["s_up"] = torch.where(sdv, N["up"], s_up)
        N["s_gp"] = torch.where(sd, N["gp"], s_gp)
        N["s_weight"] = torch.where(sdv, torch.full_like(s_weight, -math.inf), s_weight)
        N["s_rsum"] = torch.where(sd, torch.zeros_like(s_rsum), s_rsum)
        N["s_turn"] = s_turn & ~sdv
        N["s_div"] = s_div & ~sdv
        N["s_acc"] = torch.where(sdv, torch.zeros_like(s_acc), s_acc)
        N["s_n"] = torch.where(sdv, torch.zeros_like(s_n), s_n)
        fresh = sub_done[None, :, None]
        N["r_ck"] = torch.where(fresh, torch.zeros_like(r_ck), r_ck)
        N["rs_ck"] = torch.where(fresh, torch.zeros_like(rs_ck), rs_ck)
        return N

    # ---------------------------------------------------------------- driver
    def run(self, z0: torch.Tensor, num_warmup: int, num_samples: int, init_step_size: float = 1.0,
            progress: Optional[Callable] = None, step_size=None, inverse_mass=None) -> NUTSResult:
        C, D = z0.shape
        dev, dt, Dm = z0.device, torch.float64, self.max_depth
        total = num_warmup + num_samples
        gen = torch.Generator(device=dev).manual_seed(self.seed)
        z = z0.clone().to(dt)
        eye = torch.eye(D, dtype=dt, device=dev).expand(C, D, D).contiguous()
        u, g = self._eval(z)
        eps, imm0, mms0 = self._initial_kernel(z, u, g, init_step_size, step_size, inverse_mass, gen)
        windows = _adaptation_windows(num_warmup)
        K = dict(C=C, D=D, total=total, num_warmup=num_warmup, num_samples=num_samples, eye=eye,
                 levels=torch.arange(Dm, device=dev),
                 w_start=torch.tensor([w[0] for w in windows] + [total + 1], device=dev),
                 w_end=torch.tensor([w[1] for w in windows] + [total + 2], device=dev))
        zf = lambda *s: torch.zeros(s, dtype=dt, device=dev)
        zl_ = lambda *s: torch.zeros(s, dtype=torch.long, device=dev)
        zb = lambda *s: torch.zeros(s, dtype=torch.bool, device=dev)
        r0 = _mv(mms0, torch.randn((C, D), dtype=dt, device=dev, generator=gen))
        right = torch.rand(C, device=dev, generator=gen) < 0.5
        S = dict(z=z, u=u, g=g, eps=eps, eps_avg=eps.clone(), da_mu=torch.log(10.0 * eps), da_xbar=zf(C), da_gbar=zf(C),
                 da_t=zf(C), imm=imm0.clone(), mm_sqrt=mms0.clone(), wi=zl_(C), wf_n=zf(C), wf_mean=zf(C, D), wf_m2=zf(C, D, D),
                 it=zl_(C), pend_cov=eye.clone(), need_mm=zb(C),
                 e0=u + self._kinetic(imm0, r0), zl=z.clone(), rl=r0.clone(), gl=g.clone(), zr=z.clone(), rr=r0.clone(),
                 gr=g.clone(), zp=z.clone(), up=u.clone(), gp=g.clone(), weight=zf(C), r_sum=r0.clone(), sum_acc=zf(C),
                 n_prop=zl_(C), depth=zl_(C), right=right,
                 sgn=torch.where(right, torch.ones(C, dtype=dt, device=dev), -torch.ones(C, dtype=dt, device=dev)),
                 leaf=zl_(C), zc=z.clone(), rc=r0.clone(), gc=g.clone(), s_zp=z.clone(), s_up=u.clone(), s_gp=g.clone(),
                 s_weight=torch.full((C,), -math.inf, dtype=dt, device=dev), s_rsum=zf(C, D), s_turn=zb(C), s_div=zb(C),
                 s_acc=zf(C), s_n=zl_(C), r_ck=zf(Dm, C, D), rs_ck=zf(Dm, C, D),
                 out_z=zf(C, num_samples, D), out_acc=zf(C, num_samples), out_n=zl_(C, num_samples),
                 out_div=zb(C, num_samples))
        S = {k: v.contiguous() for k, v in S.items()}
        Kb = self.block
        ru_buf = torch.zeros((Kb, 3, C), dtype=dt, device=dev)
        rn_buf = torch.zeros((Kb, C, D), dtype=dt, device=dev)
        kidx = torch.zeros(1, dtype=torch.long, device=dev)

        def one_step():
            ru = ru_buf.index_select(0, kidx)[0]
            rn = rn_buf.index_select(0, kidx)[0]
            new = self._step(S, K, ru, rn)
            for key, val in new.items():
                if val is not S[key]:
                    S[key].copy_(val)
            kidx.add_(1)

        graph = None
        blocks = 0
        while True:
            if not bool((S["it"] < total).any()):
                break
            # ---- block boundary (host): fresh randoms, deferred mass-matrix updates, progress
            ru_buf.copy_(torch.rand((Kb, 3, C), device=dev, generator=gen).to(dt))
            rn_buf.copy_(torch.randn((Kb, C, D), dtype=dt, device=dev, generator=gen))
            kidx.zero_()
            if bool(S["need_mm"].any()):
                need = S["need_mm"]
                imm = torch.where(need[:, None, None], S["pend_cov"], S["imm"])
                S["imm"].copy_(imm)
                S["mm_sqrt"].copy_(torch.where(need[:, None, None], torch.linalg.cholesky(torch.linalg.inv(imm)), S["mm_sqrt"]))
                warm = need & (S["it"] < num_warmup)          # restart dual averaging around the running average
                S["eps"].copy_(torch.where(warm, S["eps_avg"], S["eps"]))
                S["da_mu"].copy_(torch.where(warm, torch.log(10.0 * S["eps_avg"]), S["da_mu"]))
                for key in ("da_t", "da_gbar", "da_xbar"):
                    S[key].copy_(torch.where(warm, torch.zeros_like(S[key]), S[key]))
                S["need_mm"].zero_()
            if self.use_graph and dev.type == "cuda" and graph is None and blocks >= 1:
                try:
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        one_step()
                    kidx.zero_()                          # capture does not execute
                except Exception as err:  # pragma: no cover - depends on the model
                    torch.cuda.synchronize()
                    print(f"[dynode_amd] NUTS iteration not graph-capturable ({type(err).__name__}: {str(err)[:120]}); eager", file=sys.stderr, flush=True)
                    graph, self.use_graph = None, False
                    kidx.zero_()
            for _ in range(Kb):
                if graph is not None:
                    graph.replay()
                else:
                    one_step()
            self.evals += Kb if graph is not None else 0
            blocks += 1
            if progress is not None:
                m = int(S["it"].min())
                progress(max(m - 1, 0), m <= num_warmup)
        return NUTSResult(S["out_z"], S["out_acc"], S["out_n"], S["out_div"], S["eps"], S["imm"], self.evals)


class KernelNUTS(LockstepNUTS):
    """The asynchronous sampler with its bookkeeping as ONE hand-written HIP kernel per iteration.

    `GraphNUTS` replays about 400 small torch kernels per iteration around the potential.  Here an
    iteration is [the potential's own kernels] + `dyn_nuts_advance` (csrc/nuts_kernel.hip, one GPU
    thread per chain: leapfrog halves, tree, adaptation incl. the mass-matrix Cholesky, recording,
    counter-based Philox randomness), captured together in one HIP graph.  The host replays the
    graph and looks at the per-chain transition counters every `block` iterations.  Needs the HIP
    library and device tensors (no CPU form); dimension <= 32 (one compiled instance per dimension up to 8, one thread per chain; a half wave
    per chain beyond), tree depth <= 10.

    ``adaptation="per_chain"`` is numpyro's behaviour: every chain estimates its own dense mass
    matrix from its own window.  ``"pooled"`` merges the window statistics of all chains of this
    call (order-independent fixed-point sums, so runs stay reproducible) and every chain uses the
    merged estimate (and ends warm-up with the geometric mean of the final step sizes of the chains
    that finished before it): with an asynchronous batch the run lasts as long as its slowest chain, and
    a chain whose own window caught a rare tail excursion otherwise ends warm-up with a several
    times too small step (measured on cfg 4, 1024 chains: slowest chain 17.9 leapfrogs per draw
    against a mean of 4.7).  Warm-up draws are discarded either way; after warm-up each chain's
    kernel is fixed, so the sampler stays exact.
    """

    def __init__(self, *args, block: int = 64, use_graph: bool = True, adaptation: str = "per_chain", fuse: bool = True, **kw):
        super().__init__(*args, **kw)
        self.fuse = bool(fuse)
        # iterations captured into one HIP graph: between two graph launches the GPU idles for about 8 us (a tenth of a cfg 4
        # iteration), between two kernel nodes of one graph it does not
        self.unroll = max(1, int(os.environ.get("DYNODE_NUTS_UNROLL", "16")))
        if int(block) % self.unroll:
            self.unroll = 1
        if adaptation not in ("per_chain", "pooled"):
            raise ValueError("adaptation must be 'per_chain' or 'pooled'")
        self.block, self.use_graph, self.adaptation = int(block), bool(use_graph), adaptation
        self.monitor: Optional[Callable] = None     # diagnostics hook: called with the state dict every block
        # a folded potential is re-checked against the model's own log joint at the chains' current positions after these
        # blocks of warm-up (infer/folded.py: a parameter map that is piecewise far from the centre); a mismatch raises
        self.recheck_blocks = (1, 3, 7)

    def run(self, z0: torch.Tensor, num_warmup: int, num_samples: int, init_step_size: float = 1.0,
            progress: Optional[Callable] = None, step_size=None, inverse_mass=None) -> NUTSResult:
        import ctypes

        from .. import _abi

        C, D = z0.shape
        dev, dt, Dm = z0.device, torch.float64, self.max_depth
        if dev.type != "cuda":
            raise RuntimeError("KernelNUTS runs on the GPU only (dyn_nuts_advance); use BatchedNUTS on CPU tensors")
        # pooled windows see (chains x window) draws: the first one can open after 25 transitions instead
        # of Stan's single-chain 75 (the identity-metric phase is the most expensive part of warm-up)
        early = self.adaptation == "pooled" and C >= 16 and num_warmup >= 150
        windows = _adaptation_windows(num_warmup, 25 if early else 75)
        if D > _abi.NUTS_MAX_DIM or Dm > _abi.NUTS_MAX_DEPTH or len(windows) > _abi.NUTS_MAX_WINDOWS:
            raise ValueError(f"KernelNUTS supports dim <= {_abi.NUTS_MAX_DIM}, max_tree_depth <= {_abi.NUTS_MAX_DEPTH}")
        if D > _abi.NUTS_REG_DIM and self.adaptation == "pooled":
            raise ValueError(f"pooled adaptation windows are compiled for dim <= {_abi.NUTS_REG_DIM}")
        L = _abi.lib()
        total = num_warmup + num_samples
        gen = torch.Generator(device=dev).manual_seed(self.seed)
        z = z0.clone().to(dt).contiguous()
        u, g = self._eval(z)
        eps, imm0, mms0 = self._initial_kernel(z, u, g, init_step_size, step_size, inverse_mass, gen)
        zf = lambda *s: torch.zeros(s, dtype=dt, device=dev)
        zi = lambda *s: torch.zeros(s, dtype=torch.int32, device=dev)
        W = max(len(windows), 1)
        r0 = _mv(mms0, torch.randn((C, D), dtype=dt, device=dev, generator=gen))
        right = torch.rand(C, device=dev, generator=gen) < 0.5
        sgn = torch.where(right, 1.0, -1.0).to(dt)
        r_half = r0 - 0.5 * (eps * sgn)[:, None] * g
        S = dict(z=z, u=u, g=g, eps=eps, eps_avg=eps.clone(), da_mu=torch.log(10.0 * eps), da_xbar=zf(C), da_gbar=zf(C),
                 da_t=zf(C), imm=imm0.clone(), mm_sqrt=mms0.clone(), wf_n=zf(C), wf_mean=zf(C, D), wf_m2=zf(C, D, D),
                 e0=u + self._kinetic(imm0, r0), zl=z.clone(), rl=r0.clone(), gl=g.clone(), zr=z.clone(), rr=r0.clone(),
                 gr=g.clone(), zp=z.clone(), up=u.clone(), gp=g.clone(), weight=zf(C), r_sum=r0.clone(), sum_acc=zf(C),
                 sgn=sgn, zc=z.clone(), rc=r0.clone(), gc=g.clone(), r_half=r_half, s_zp=z.clone(), s_up=u.clone(),
                 s_gp=g.clone(), s_weight=torch.full((C,), -math.inf, dtype=dt, device=dev), s_rsum=zf(C, D),
                 s_acc=zf(C), r_ck=zf(C, Dm, D), rs_ck=zf(C, Dm, D),
                 z_eval=z + (eps * sgn)[:, None] * _mv(imm0, r_half), u_new=zf(C), g_new=zf(C, D),
                 it=zi(C), wi=zi(C), n_prop=zi(C), depth=zi(C), right=right.to(torch.int32), leaf=zi(C), s_turn=zi(C),
                 s_div=zi(C), s_n=zi(C), rng_ctr=torch.zeros(C, dtype=torch.int64, device=dev),
                 pool=torch.zeros((W + 1, 1 + D + D * D), dtype=torch.int64, device=dev),
                 pool_ro=torch.zeros((W + 1, 1 + D + D * D), dtype=torch.int64, device=dev), pend=zi(C),
                 out_z=zf(C, num_samples, D), out_acc=zf(C, num_samples), out_n=zi(C, num_samples),
                 out_div=zi(C, num_samples))
        S = {k: v.contiguous() for k, v in S.items()}
        st = _abi.NutsStateC()
        st.n_chains, st.dim, st.max_depth = C, D, Dm
        st.num_warmup, st.num_samples, st.n_windows = num_warmup, num_samples, len(windows)
        st.pooled = int(self.adaptation == "pooled")
        for i, (a, b) in enumerate(windows):
            st.w_start[i], st.w_end[i] = a, b
        st.seed = (self.seed * 0x9E3779B97F4A7C15 + 0x1234567) & (2 ** 64 - 1)
        st.target_accept, st.max_delta_energy = self.target, self.max_de
        for name in _abi.NUTS_POINTER_FIELDS:
            t = S[name]
            want = torch.int32 if name in _abi.NUTS_INT32_FIELDS else torch.int64 if name in _abi.NUTS_INT64_FIELDS else dt
            assert t.dtype == want and t.is_contiguous() and t.device == dev, name
            setattr(st, name, t.data_ptr())

        # a folded potential (infer/folded.py) hands its parts to the kernel, and the kernel prepares the next position's
        # prior side and parameter rows itself: an iteration is the gradient-solve and dyn_nuts_advance_mapped
        folded = self.pg if hasattr(self.pg, "solve_current") else None
        keep_alive = []
        if folded is not None:
            folded.map_now(S["z_eval"])

        # ... and where the library can fuse the sampler's side into the gradient-solve (dyn_solver_opts::nuts_tail: whole
        # chains inside a wave, at most eight sites), an iteration is that ONE launch.  DYNODE_NUTS_FUSE=0 keeps the two.
        tail = {"blob": None}
        if folded is not None and self.fuse and hasattr(folded, "pack_tail") and os.environ.get("DYNODE_NUTS_FUSE", "1") != "0":
            b = folded._buffers(C)
            st.pot_lp, st.pot_dlp, st.pot_offset = b["lp"].data_ptr(), b["dlp"].data_ptr(), float(folded.offset)
            tail["blob"] = folded.pack_tail(st, C)
        self.launches_per_iteration = None

        def iteration():
            if tail["blob"] is not None:
                from ..engine import SolveError

                try:
                    keep_alive[:] = folded.solve_current(C, nuts_tail=tail["blob"])[:4]
                    if st.pooled:   # readers of the next launch see the pool as it stands now (nuts_kernel.hip: advance)
                        S["pool_ro"].copy_(S["pool"])
                    self.launches_per_iteration = 1
                    return
                except SolveError as err:
                    if err.code != -7:
                        raise
                    tail["blob"] = None    # nothing was enqueued: this and every later iteration take the two launches
            if folded is not None:
                self.launches_per_iteration = 2
                lp_, dlp_, ll_, dll_, stride = folded.solve_current(C)
                keep_alive[:] = [lp_, dlp_, ll_, dll_]
                st.pot_lp, st.pot_dlp, st.pot_ll, st.pot_dll = lp_.data_ptr(), dlp_.data_ptr(), ll_.data_ptr(), dll_.data_ptr()
                st.pot_offset, st.pot_ll_stride, st.pot_dll_stride = float(folded.offset), int(stride), int(dll_.shape[1])
                rc = folded.advance_mapped(st, C)
            else:
                u_, g_ = self.pg(S["z_eval"])
                S["u_new"].copy_(u_)
                S["g_new"].copy_(g_)
                rc = L.dyn_nuts_advance(ctypes.byref(st), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc:
                raise RuntimeError(f"dyn_nuts_advance: {_abi.ERR_NAMES.get(rc, rc)}")

        graph = None
        blocks = 0
        while bool((S["it"] < total).any()):
            if self.use_graph and graph is None and blocks >= 1:
                try:
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        for _ in range(self.unroll):
                            iteration()
                except Exception as err:  # pragma: no cover - depends on the model
                    torch.cuda.synchronize()
                    print(f"[dynode_amd] NUTS iteration not graph-capturable ({type(err).__name__}: {str(err)[:120]}); eager", file=sys.stderr, flush=True)
                    graph, self.use_graph = None, False
            if graph is not None:
                for _ in range(self.block // self.unroll):
                    graph.replay()
            else:
                for _ in range(self.block):
                    iteration()
            self.evals += self.block
            blocks += 1
            if folded is not None and blocks in self.recheck_blocks and hasattr(folded, "verify"):
                if not folded.verify(S["z"].clone()):
                    from .folded import FoldMismatch

                    raise FoldMismatch(f"folded potential != model log joint at the chains' positions after {blocks * self.block} iterations")
            if self.monitor is not None:
                self.monitor(S)
            if progress is not None:
                m = int(S["it"].min())
                progress(max(m - 1, 0), m <= num_warmup)
        self._keep = (graph, st, S, tail["blob"])
        return NUTSResult(S["out_z"], S["out_acc"], S["out_n"].long(), S["out_div"].bool(), S["eps"], S["imm"], self.evals)
