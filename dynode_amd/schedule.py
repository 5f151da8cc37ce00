"""Cost-ordered dispatch of a batch: which trajectories share a wave, and which waves start first.

OPT-IN since round 3 (``solve_batch(order="forecast")``).  The default dispatch needs no forecast: batches beyond one
resident round run as a work-pulling grid (csrc/solve_kernel.hpp, ``Solver::run``), which removes the lock-step waiting and
the round structure for any batch in its given order.  What a forecast still adds on top of that is longest-first order at the
tail of a launch; a caller who has one -- this module's learned regression, or knowledge of its own -- passes it as the
queue (``order=<int32 tensor>``, `dyn_solve_batch_ordered`).

The solve kernels step the 2..32 trajectories of a wavefront in lock-step and the GPU starts waves in index order, so a
launch is shorter when neighbours in the batch need similar numbers of steps and the expensive ones come first (longest
processing time first): 10-15 % on the BASELINE shapes (`tools/probes/probe_sorted.py`).  The outputs
cannot change -- every trajectory is computed from its own inputs and lands in its own rows, `dyn_solve_batch_ordered` only
permutes the assignment of trajectories to grid slots (tests/test_gpu_parity.py compares bit for bit).

What is needed is a forecast of the step count of a trajectory before it is solved.  The kernels return the exact count
(`n_accept + n_reject`) after every launch, so the forecast is learned from the solver's own output: a ridge regression of
the step count on a quadratic form in the standardised logarithms of the parameters that vary over the batch (SEIR-type
models: correlation 0.93 with the true count on unseen draws of the BASELINE cfg 3 prior against 0.75-0.81 for a linear
model, 0.98 when the strains are relabelled canonically first -- `CostModel`).  `solve_batch(order="forecast")` trains on the launches
it sees until `TRAIN_ROWS` trajectories of a (model, solver settings) pair have been observed, then orders every later batch
with two small launches (`dyn_cost_order`: forecast + bucket, counting sort) in front of the solve.  Nothing synchronises
with the host after the first training launch (which looks at the batch once to pick the varying parameters); a training launch
costs a feature matrix and a 150 x 150 solve on the device next to it (a millisecond or two for 16384 rows), the first two or
three launches of a (model, settings) pair only.

The forecast is keyed on the model, the solver settings and the CONTENT of the inputs all trajectories share (contact matrix,
shared initial state, discontinuity points): change one and a new forecast is learned.  `reset()` forgets everything.
The reference has no counterpart (diffrax under `vmap` on XLA:CPU runs the samples one after another).
"""

from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import torch

from . import _abi

TRAIN_ROWS = 32768          # trajectories observed per (model, settings) before the forecast is frozen
MIN_ROWS_PER_FEATURE = 12   # first fit once this many rows per regression coefficient were seen
MIN_BATCH = 1024            # smaller batches are launched in the given order ...
MIN_WORK = 1 << 21          # ... and so are launches below this many state values (B x D): the two extra launches cost
                            # ~20 us; cfg 2 (4096 x 24, 0.2 ms) and cfg 5's share (8192 x 136, 0.65 ms, one wave per SIMD)
                            # gain less than that inside a host loop
MAX_FIT_ROWS = 65536        # rows of one launch that enter the normal equations
KEY_SCALE = 4.0             # buckets per predicted step attempt (4096 buckets: up to 1024 attempts)
RIDGE = 1e-4
ONE_ROUND_WAVES = int(os.environ.get("DYNODE_ONE_ROUND", "1")) * 2048   # 1024 SIMDs x 2 resident waves of the <= 256-register kernels: up to here a launch is one round (DYNODE_ONE_ROUND=0: never deal)
MIN_R2 = 0.5                # a forecast that explains less of the step-count variance than this is not used (given order)


def capacity(n: int) -> int:
    """Compiled feature capacity of `dyn_cost_order` for n features (``dyn_cost_order_capacity``)."""
    return next(c for c in (4, 8, 16, 24, 32) if n <= c)


def enabled() -> bool:
    return os.environ.get("DYNODE_ORDER", "1") != "0"


class _Regression:
    """One ridge regression of the step count on [1, z, z z^T] -- z the standardised features of the picked columns,
    read either as given (``sym=None``) or in the canonical strain labelling (``sym=(S, blocks)``, see `dyn_cost_order`)."""

    def __init__(self, P: int, device, sym):
        self.P, self.device, self.sym = int(P), device, sym
        self.cols = None            # int32 [n]: dyn_cost_order's encoding of the feature columns
        self.coef = None            # float32: dyn_cost_order's coefficient array at the compiled feature capacity
        self.rss = float("inf")     # in-sample residual variance of the last fit
        self.variance = 0.0         # ... and the variance of the step count itself

    def canonical(self, params: torch.Tensor) -> torch.Tensor:
        """Parameter rows with the exchangeable strain blocks relabelled: strains sorted by block 0 / block 1, largest first."""
        if self.sym is None:
            return params
        S, F = self.sym
        blk = params[:, :F * S].reshape(-1, F, S)
        rank = torch.argsort(blk[:, 0] / blk[:, 1], dim=1, descending=True, stable=True)
        out = params.clone()
        out[:, :F * S] = torch.gather(blk, 2, rank[:, None, :].expand(-1, F, -1)).reshape(-1, F * S)
        return out

    def pick_columns(self, params: torch.Tensor, attempts: torch.Tensor) -> bool:
        p = self.canonical(params.double())
        lo, hi = p.min(0).values, p.max(0).values
        varying = hi > lo + 1e-6 * torch.maximum(hi.abs(), lo.abs())
        as_log = varying & (lo > 0) & (hi < 100.0 * lo)         # rates and periods; anything that reaches zero stays linear
        cols = torch.nonzero(varying)[:, 0]
        if cols.numel() == 0:
            return False
        self._idx, self._log = cols, as_log[cols]
        if cols.numel() > _abi.MAX_COST_FEATURES:
            # screening: keep the columns whose value (or its square) moves with the step count the most
            f = self._transform(p)
            z = (f - f.mean(0)) / f.std(0).clamp_min(1e-12)
            a = (attempts.double() - attempts.double().mean()) / attempts.double().std().clamp_min(1e-12)
            score = torch.maximum((z * a[:, None]).mean(0).abs(), ((z * z - 1.0) * a[:, None]).mean(0).abs())
            keep = torch.argsort(score, descending=True)[:_abi.MAX_COST_FEATURES].sort().values
            self._idx, self._log = cols[keep], self._log[keep]
        f = self._transform(p)
        self.cols = torch.where(self._log, self._idx, -(self._idx + 1)).to(torch.int32).contiguous()
        self.centre = f.mean(0)
        self.inv_spread = 1.0 / f.std(0).clamp_min(1e-12)
        n = int(self._idx.numel())
        nq = 1 + n + n * (n + 1) // 2
        self.G = torch.zeros((nq, nq), dtype=torch.float64, device=self.device)
        self.r = torch.zeros(nq, dtype=torch.float64, device=self.device)
        self.yy = torch.zeros((), dtype=torch.float64, device=self.device)
        self.wsum = torch.zeros((), dtype=torch.float64, device=self.device)
        self._iu = torch.triu_indices(n, n, device=self.device)
        return True

    def _transform(self, canonical_params: torch.Tensor) -> torch.Tensor:
        v = canonical_params[:, self._idx].double()
        return torch.where(self._log, torch.log(v.clamp_min(1e-30)), v)

    def features(self, params: torch.Tensor) -> torch.Tensor:
        z = (self._transform(self.canonical(params)) - self.centre) * self.inv_spread
        return torch.cat([torch.ones(z.shape[0], 1, dtype=torch.float64, device=z.device), z, z[:, self._iu[0]] * z[:, self._iu[1]]], dim=1)

    def add(self, params, attempts, weight) -> None:
        keep = weight[:, None] > 0
        Q = torch.where(keep, self.features(params), torch.zeros((), dtype=torch.float64, device=params.device))   # (failed rows may hold NaN)
        Qw = Q * weight[:, None]
        y = attempts.double() * weight
        self.G += Qw.T @ Q
        self.r += Qw.T @ y
        self.yy += (y * y).sum()
        self.wsum += weight.sum()

    def fit(self) -> None:
        ridge = RIDGE * torch.diagonal(self.G).clamp_min(1e-12)
        ridge[0] = 0.0
        w = torch.linalg.solve(self.G + torch.diag(ridge), self.r)
        self.w = w
        # dyn_cost_order's layout at the compiled feature capacity NF >= n: features beyond n have zero coefficients and spread
        n, NF = int(self._idx.numel()), capacity(int(self._idx.numel()))
        quad = torch.zeros((NF, NF), dtype=torch.float64, device=self.device)
        quad[self._iu[0], self._iu[1]] = w[1 + n:]
        iu = torch.triu_indices(NF, NF, device=self.device)
        pad = torch.zeros(NF - n, dtype=torch.float64, device=self.device)
        self.coef = torch.cat([w[:1 + n], pad, quad[iu[0], iu[1]], self.centre, pad, self.inv_spread, pad]).float().contiguous()
        self.cols_padded = torch.cat([self.cols, self.cols[:1].expand(NF - n)]).contiguous()
        n = self.wsum.clamp_min(1.0)
        self.rss = float((self.yy - 2.0 * (w @ self.r) + w @ (self.G @ w)) / n)
        self.variance = float(self.yy / n - (self.r[0] / n) ** 2)          # (feature 0 is the constant: r[0] = sum of y)

    def forecast(self, params: torch.Tensor) -> torch.Tensor:
        return self.features(params) @ self.w


class CostModel:
    """Forecast of the step attempts of a trajectory from its parameter row (see the module docstring).
    ``sym=(S, blocks)``: the model treats its S strains alike and its first ``blocks * S`` parameters are [quantity][strain]
    blocks; a second regression on the canonical strain labelling is trained next to the plain one and the one with the
    smaller residual is used (correlation on unseen cfg 3 draws: 0.93 plain, 0.98 canonical; an initial state or contact
    structure that singles out a strain makes the plain one win)."""

    def __init__(self, P: int, device, sym=None):
        self.P, self.device = int(P), device
        self.variants = [_Regression(P, device, None)] + ([_Regression(P, device, sym)] if sym and sym[0] > 1 else [])
        self.best: Optional[_Regression] = None
        self.rows = 0                                 # rows that entered the normal equations
        self.fitted_rows = 0
        self.unusable = False                         # no varying parameter: nothing to forecast with
        self._picked = False

    @property
    def training(self) -> bool:
        return not self.unusable and self.rows < TRAIN_ROWS

    @property
    def ready(self) -> bool:
        return self.best is not None

    @property
    def cols(self):
        return self.best.cols if self.best is not None else self.variants[0].cols

    def observe(self, params: torch.Tensor, attempts: torch.Tensor, status: torch.Tensor) -> None:
        """Add one launch (parameter rows, n_accept + n_reject, status) to the normal equations and refit when the
        number of observed rows has doubled.  Failed solves (status != 0) stopped early: they carry no weight."""
        params, attempts, status = params[:MAX_FIT_ROWS], attempts[:MAX_FIT_ROWS], status[:MAX_FIT_ROWS]
        if not self._picked:                   # the one look at the data from the host (first training launch only)
            self._picked = True
            good = status == 0                 # (rows of failed solves may hold anything, NaN included)
            self.variants = [v for v in self.variants if v.pick_columns(params[good], attempts[good])] if bool(good.any()) else []
            if not self.variants:
                self.unusable = True
        if self.unusable:
            return
        w = (status == 0).double()
        for v in self.variants:
            v.add(params, attempts, w)
        self.rows += int(params.shape[0])
        nq = max(v.G.shape[0] for v in self.variants)
        if self.rows >= MIN_ROWS_PER_FEATURE * nq and self.rows >= 2 * self.fitted_rows:
            for v in self.variants:
                v.fit()
            best = min(self.variants, key=lambda v: v.rss)
            self.best = best if best.rss <= (1.0 - MIN_R2) * best.variance else None
            self.fitted_rows = self.rows

    def order(self, params_t: torch.Tensor, stream, deal_waves_of: int = 0) -> torch.Tensor:
        """int32 [B]: dispatch order of the rows of ``params_t`` ([B, P], float32 / float64, on the device).
        ``deal_waves_of`` = trajectories per wave when the launch is one residency round (see `dyn_cost_order`)."""
        from .engine import _DTYPES

        v = self.best
        B = params_t.shape[0]
        keys = torch.empty(B, dtype=torch.int32, device=params_t.device)
        order = torch.empty(B, dtype=torch.int32, device=params_t.device)
        S, F = v.sym if v.sym is not None else (0, 0)
        rc = _abi.lib().dyn_cost_order(params_t.data_ptr(), _DTYPES[params_t.dtype], B, self.P, int(v.cols_padded.numel()),
                                       v.cols_padded.data_ptr(), v.coef.data_ptr(), KEY_SCALE, int(S), int(F), int(deal_waves_of),
                                       keys.data_ptr(), order.data_ptr(), ctypes.c_void_p(stream.cuda_stream))
        if rc:
            raise RuntimeError(f"dyn_cost_order: {_abi.ERR_NAMES.get(rc, rc)}")
        for t in (keys, order, v.cols_padded, v.coef):
            t.record_stream(stream)
        return order

    def forecast(self, params: torch.Tensor) -> torch.Tensor:
        """Predicted step attempts (float64 [B]) -- the torch statement of what `cost_keys` evaluates; tests and probes."""
        return self.best.forecast(params)


_MODELS: Dict[tuple, CostModel] = {}


def reset() -> None:
    """Forget every trained forecast (a new prior, a new initial state or contact matrix: retrain from the next launches)."""
    _MODELS.clear()


def model_for(key: tuple, P: int, device, sym=None) -> CostModel:
    m = _MODELS.get(key)
    if m is None:
        if len(_MODELS) > 64:
            _MODELS.clear()
        m = _MODELS[key] = CostModel(P, device, sym)
    return m


def strain_symmetry(model):
    """(S, blocks) for `CostModel` if the model's leading parameters are per-strain blocks of exchangeable strains: the
    s/e/i/r/c family (beta, gamma, sigma, omega [, introduction time / scale / size]; include/dynode_hip.h), else None."""
    if model.family != 0 or model.n_strain < 2 or model.n_strain > 8:
        return None
    return model.n_strain, 2 + int(model.has_e) + int(model.has_wane) + (3 if model.has_intro else 0)
