"""Declarative ODE right-hand sides: the plugin surface for ``simulate``'s ``ode`` argument.

In the reference ``ode`` is an arbitrary JAX callable ``(t, state, params) -> grads``
(src/dynode/typing/typing.py:18-21) that diffrax traces.  A Python callable cannot run inside a
HIP kernel, so here ``ode`` is a :class:`CompartmentalODE` descriptor naming one member of the
compartmental family the fused kernel implements (include/dynode_hip.h).  The descriptors below
cover every RHS the reference ships, with the same names, the same
parameter dataclasses and the same state-tuple order, so user code written against the
reference examples runs unchanged.  Anything outside the family is rejected loudly.

A descriptor is still callable -- ``ode(t, state, p)`` evaluates the derivative with NumPy for
inspection and for the property tests -- but ``simulate`` never uses that path.
"""

from __future__ import annotations

from dataclasses import dataclass, fields
from types import SimpleNamespace
from typing import Any, Optional, Tuple

import numpy as np
import torch

from ._abi import ModelDesc


def _np(x) -> np.ndarray:
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy().astype(np.float64)
    return np.asarray(x, dtype=np.float64)


@dataclass
class AbstractODEParams:
    """Base of the parameter containers handed to an ODE (reference odes.py:25-32).

    Fields are arrays (numpy / torch / python scalars).  A leading batch axis on any field makes
    the solve batched: one trajectory per row.
    """


@dataclass
class SIR_ODEParams(AbstractODEParams):
    """examples/sir.py:70-74 and sir_age_stratified.py:103-107 (contact_matrix optional)."""

    beta: Any
    gamma: Any
    contact_matrix: Any = None


@dataclass
class SEIRS_ODEParams(AbstractODEParams):
    """examples/seirs.py:80-85."""

    beta: Any
    gamma: Any
    sigma: Any
    omega: Any


@dataclass
class SeasonalityParams:
    """examples/seirs_seasonal_forcing.py:22-26."""

    forcing_amp: Any
    forcing_phase: Any
    forcing_period: Any


@dataclass
class SEIRS_Seasonal_ODEParams(AbstractODEParams):
    """examples/seirs_seasonal_forcing.py:30-36."""

    beta: Any
    gamma: Any
    sigma: Any
    omega: Any
    seasonality_params: SeasonalityParams


@dataclass
class IntroductionParams:
    """Strains seeded from an untracked external population (``Strain.is_introduced`` and its
    ``introduction_*`` fields, reference src/dynode/config/strains.py:53-109).  Around day
    ``time`` infectious visitors worth ``percentage`` of every receiving age bin's population mix
    with it, spread in time as a normal density of standard deviation ``scale``; in the force of
    infection I_b becomes I_b + Normal(t; time, scale) * percentage * P_b (ode_model.md).
    A strain that is not introduced has percentage 0."""

    time: Any            # [S] or [B, S]   introduction_time (simulation days)
    scale: Any           # [S] or [B, S]   introduction_scale (days)
    percentage: Any      # [S] or [B, S]   introduction_percentage
    ages_mask: Any       # [S, A] of 0/1   Strain.introduction_ages_mask_vector per strain


def introduction_params(strains, initialize_date=None) -> Optional["IntroductionParams"]:
    """`IntroductionParams` from the ``Strain`` objects of a validated ``SimulationConfig`` (which has
    filled ``introduction_ages_mask_vector``); None when no strain is introduced.  A calendar
    ``introduction_time`` is converted to simulation days with ``initialize_date``."""
    import datetime

    if not any(s.is_introduced for s in strains):
        return None
    n_age = next(len(s.introduction_ages_mask_vector) for s in strains if s.introduction_ages_mask_vector is not None)

    def day(s):
        t = s.introduction_time
        if isinstance(t, datetime.date):
            if initialize_date is None:
                raise ValueError("a calendar introduction_time needs the model's initialize_date")
            return float((t - initialize_date).days)
        return t if s.is_introduced else 0.0

    live = lambda s, value, default: value if (s.is_introduced and value is not None) else default
    def stack(vals):
        """per-strain scalars -> [S] (numpy), or a tensor that keeps autograd graphs and batch axes"""
        tensors = [v for v in vals if isinstance(v, torch.Tensor)]
        if not tensors:
            return np.array([float(v) for v in vals])
        like = tensors[0]
        parts = [v.to(like) if isinstance(v, torch.Tensor) else torch.full_like(like, float(v)) for v in vals]
        return torch.stack(torch.broadcast_tensors(*parts), dim=-1)

    return IntroductionParams(
        time=stack([day(s) for s in strains]),
        scale=stack([live(s, s.introduction_scale, 1.0) for s in strains]),
        percentage=stack([live(s, s.introduction_percentage, 0.0) for s in strains]),
        ages_mask=np.array([s.introduction_ages_mask_vector if (s.is_introduced and s.introduction_ages_mask_vector is not None)
                            else [0] * n_age for s in strains], dtype=float))


@dataclass
class VaccinationParams:
    """Vaccination fluxes (ode_model.md; ``VaccinationDimension``, ``Strain.vaccine_efficacy``, the spline
    arguments of ``utils.evaluate_cubic_spline`` -- reference src/dynode/utils/splines.py:66-109).

    Compartments gain a tier axis after age: ``s[A, K]``, ``e/i/r/c[A, K, S]``.  Per day
    ``nu_{a,k}(t) * (population of age a)`` doses reach the susceptibles of tier ``k`` (at most as many
    as there are) and move them to tier ``k + 1``; ``nu`` is the cubic spline given here per (age, tier);
    the last tier keeps its people.  A tier's susceptibility to strain ``l`` is
    ``1 - vaccine_efficacy[l, k]``."""

    knot_locations: Any      # [A, K, n_knots]   n_knots <= 4
    base_equations: Any      # [A, K, 4]         a + b t + c t^2 + d t^3
    knot_coefficients: Any   # [A, K, n_knots]
    vaccine_efficacy: Any    # [S, K]            0 = no protection .. 1 = full protection


@dataclass
class SEIRS_MultiStrain_ODEParams(AbstractODEParams):
    """examples/seirs_multi_strain_age_stratified.py:177-184 (+ optional seasonality, cfg 5, and
    optional external introductions)."""

    beta: Any            # [S] or [B, S]
    gamma: Any
    sigma: Any
    omega: Any
    contact_matrix: Any  # [A, A]
    idx: Optional[SimpleNamespace] = None
    seasonality_params: Optional[SeasonalityParams] = None
    introduction_params: Optional[IntroductionParams] = None
    vaccination_params: Optional[VaccinationParams] = None


@dataclass
class Packed:
    """Everything ``dyn_solve_batch`` needs, host side, float64."""

    model: ModelDesc
    y0: np.ndarray          # [D] or [B, D]
    params: np.ndarray      # [B, P]
    contact: np.ndarray     # [A, A]
    batch: Optional[int]    # None = unbatched call
    shapes: Tuple[tuple, ...]  # per-compartment shapes without time/batch axes
    tiers: Optional[int] = None  # vaccination: tracked tiers K; the tier axis (axis 1) is padded to 2 or 4
    history_perm: Optional[np.ndarray] = None  # SEIP: kernel history slot of each reference bin (axis 1)


class CompartmentalODE:
    """One member of the kernel's RHS family, exposed under the reference example's name."""

    def __init__(self, name: str, params_type: type, compartments: Tuple[str, ...], *,
                 multi_strain: bool = False, has_e: bool = False, has_wane: bool = False,
                 has_c: bool = False, seasonal: Optional[bool] = False, normalize: bool = True,
                 contact_ndim: int = 1, n_wane: int = 1, doc: str = ""):
        self.__name__ = name
        self.__doc__ = doc
        self.params_type = params_type
        self.compartments = compartments
        self.multi_strain = multi_strain
        self.has_e, self.has_wane, self.has_c = has_e, has_wane, has_c
        self.seasonal = seasonal  # None = decided by the params object (multi-strain, cfg 5)
        self.normalize = normalize
        self.contact_ndim = contact_ndim
        self.n_wane = n_wane

    def __repr__(self) -> str:
        return f"<CompartmentalODE {self.__name__}>"

    # ------------------------------------------------------------------ packing
    def _strain_columns(self, p) -> list:
        """(name, value) of every per-strain block of the parameter vector, in kernel order:
        beta gamma (sigma) (omega) (intro_time intro_scale intro_pct)."""
        names = ["beta", "gamma"] + (["sigma"] if self.has_e else []) + (["omega"] if self.has_wane else [])
        cols = [(n, getattr(p, n)) for n in names]
        intro = getattr(p, "introduction_params", None)
        if intro is not None:
            cols += [("introduction time", intro.time), ("introduction scale", intro.scale),
                     ("introduction percentage", intro.percentage)]
        return cols

    @staticmethod
    def _intro_masks(p, A: int, S: int) -> tuple:
        intro = getattr(p, "introduction_params", None)
        if intro is None:
            return ()
        mask = np.asarray(_np(intro.ages_mask)).reshape(S, A)
        return tuple(int(sum(1 << a for a in range(A) if mask[l, a] != 0)) for l in range(S))

    def _param_matrix(self, p) -> Tuple[np.ndarray, Optional[int], bool]:
        cols, batch = [], None
        strain_rank = 1 if self.multi_strain else 0
        arrays = []
        for n, value in self._strain_columns(p):
            a = _np(value)
            if not self.multi_strain and a.ndim == 1 and a.size == 1 and strain_rank == 0:
                a = a.reshape(())  # the examples pass shape-(1,) or scalar for single-strain models
            if a.ndim == strain_rank + 1:
                batch = a.shape[0] if batch is None else batch
                if a.shape[0] != batch:
                    raise ValueError(f"inconsistent batch sizes in ode parameters ({n})")
            elif a.ndim != strain_rank:
                raise ValueError(f"parameter {n} has shape {a.shape}; expected rank {strain_rank} "
                                 f"(or {strain_rank + 1} with a leading batch axis)")
            arrays.append(a)
        S = arrays[0].shape[-1] if self.multi_strain else 1
        seas = getattr(p, "seasonality_params", None)
        seasonal = bool(self.seasonal) if self.seasonal is not None else seas is not None
        seas_arrays = []
        if seasonal:
            for n in ("forcing_amp", "forcing_phase", "forcing_period"):
                a = _np(getattr(seas, n))
                a = a.reshape(()) if a.size == 1 and a.ndim <= 1 else a
                if a.ndim == 1:
                    batch = a.shape[0] if batch is None else batch
                    if a.shape[0] != batch:
                        raise ValueError(f"inconsistent batch sizes in seasonality parameters ({n})")
                elif a.ndim != 0:
                    raise ValueError(f"seasonality parameter {n} must be a scalar or [B]")
                seas_arrays.append(a)
        B = batch or 1
        for a in arrays:
            a2 = a.reshape(-1, S) if a.ndim == strain_rank + 1 else np.broadcast_to(a.reshape(1, S), (B, S))
            cols.append(np.broadcast_to(a2, (B, S)))
        for a in seas_arrays:
            cols.append(np.broadcast_to(a.reshape(-1, 1), (B, 1)))
        return np.ascontiguousarray(np.concatenate(cols, axis=1)), batch, seasonal

    def param_tensor(self, p, device, packed=None) -> "torch.Tensor":
        """[B, P] parameter matrix built with torch ops, keeping the autograd graph of any field
        that is a tensor requiring grad (used by the differentiable solve under NUTS)."""
        strain_rank = 1 if self.multi_strain else 0
        f64 = torch.float64

        def tt(v):
            if isinstance(v, torch.Tensor):
                return v.to(device=device, dtype=f64)
            from .engine import _dev      # constants: cached device copies (no host-to-device copy per gradient)

            return _dev(np.asarray(v, dtype=np.float64), f64, device)

        arrays = []
        for _, value in self._strain_columns(p):
            a = tt(value)
            if not self.multi_strain and a.dim() == 1 and a.numel() == 1:
                a = a.reshape(())
            arrays.append(a)
        seas = getattr(p, "seasonality_params", None)
        seasonal = bool(self.seasonal) if self.seasonal is not None else seas is not None
        extra = [tt(getattr(seas, n)) for n in ("forcing_amp", "forcing_phase", "forcing_period")] if seasonal else []
        extra = [a.reshape(()) if a.numel() == 1 and a.dim() <= 1 else a for a in extra]
        B = 1
        for a in arrays:
            if a.dim() == strain_rank + 1:
                B = max(B, a.shape[0])
        for a in extra:
            if a.dim() == 1:
                B = max(B, a.shape[0])
        S = arrays[0].shape[-1] if self.multi_strain else 1
        vp = getattr(p, "vaccination_params", None)
        blocks = []
        if vp is not None:
            # sus[A, KV, S] = 1 - efficacy (padded tier slots: 1), then [base(4) knots coefs] per (age, slot);
            # the efficacy may carry a leading batch axis (one row per chain / parameter sample)
            base, knots, coefs, ve = tt(vp.base_equations), tt(vp.knot_locations), tt(vp.knot_coefficients), tt(vp.vaccine_efficacy)
            A, K = base.shape[-3], base.shape[-2]
            KV = 2 if K == 2 else 4
            ve = ve.reshape((-1,) + tuple(ve.shape[-2:]))                    # [b, S, K]
            sus = torch.cat([1.0 - ve.transpose(1, 2), ve.new_ones((ve.shape[0], KV - K, S))], dim=1)      # [b, KV, S]
            blocks.append(sus[:, None].expand(ve.shape[0], A, KV, S).reshape(ve.shape[0], -1))
            spl = torch.cat([base, knots, coefs], dim=-1)
            spl = spl.reshape((-1,) + tuple(spl.shape[-3:]))                 # [b, A, K, 4 + 2 nk]
            spl = torch.cat([spl, spl.new_zeros((spl.shape[0], A, KV - K, spl.shape[-1]))], dim=2)
            blocks.append(spl.reshape(spl.shape[0], -1))
            B = max([B] + [b.shape[0] for b in blocks])
        cols = [a.reshape(-1, S).expand(B, S) for a in arrays] + [a.reshape(-1, 1).expand(B, 1) for a in extra]
        cols += [b.expand(B, b.shape[1]) for b in blocks]
        return torch.cat(cols, dim=1)

    def wants_grad(self, p) -> bool:
        leaves = [getattr(p, f.name) for f in fields(p)]
        vp = getattr(p, "vaccination_params", None)
        if vp is not None:
            leaves += [vp.vaccine_efficacy, vp.base_equations, vp.knot_locations, vp.knot_coefficients]
        seas = getattr(p, "seasonality_params", None)
        if seas is not None:
            leaves += [seas.forcing_amp, seas.forcing_phase, seas.forcing_period]
        intro = getattr(p, "introduction_params", None)
        if intro is not None:
            leaves += [intro.time, intro.scale, intro.percentage]
        return any(isinstance(v, torch.Tensor) and v.requires_grad for v in leaves)

    @staticmethod
    def state_wants_grad(initial_state) -> bool:
        return any(isinstance(c, torch.Tensor) and c.requires_grad for c in initial_state)

    def state_tensor(self, initial_state, packed: "Packed", device) -> "torch.Tensor":
        """The flat initial state ``[D]`` / ``[B, D]`` in `pack`'s layout, built with torch ops so that compartments
        computed from a latent site (a sampled initial-infection scale, say) keep their autograd graph; the
        differentiable solve seeds the kernel's ``dy0`` planes from it (infer/autodiff.py)."""
        if packed.tiers is not None or packed.history_perm is not None:
            raise NotImplementedError(f"{self.__name__}: gradients with respect to the initial state are available for models "
                                      "without vaccination tiers / immune histories; detach the initial state")
        f64 = torch.float64
        flat = []
        for a, shape in zip(initial_state, packed.shapes):
            t = a.to(device=device, dtype=f64) if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, dtype=np.float64), device=device)
            flat.append(t.reshape(-1) if tuple(t.shape) == tuple(shape) else t.reshape(t.shape[0], -1))
        if any(f.dim() == 2 for f in flat):
            B = max(f.shape[0] for f in flat if f.dim() == 2)
            return torch.cat([f if f.dim() == 2 else f.unsqueeze(0).expand(B, f.shape[0]) for f in flat], dim=1)
        return torch.cat(flat)

    def _contact(self, p, A: int, contact_shape: tuple) -> np.ndarray:
        C = getattr(p, "contact_matrix", None)
        if C is None:
            if A != 1:
                raise ValueError(f"{self.__name__}: a contact_matrix is required for {A} bins")
            return np.ones((1, 1))
        C = _np(C)
        if self.contact_ndim == 2:
            # sir_age_risk_stratified.py:113-115,164-166: foi_kl = sum_ij C4[i,j,k,l] x_ij
            # -> flattened foi_q = sum_p M[q,p] x_p with M = C4.reshape(AR, AR).T
            if C.shape != contact_shape + contact_shape:
                raise ValueError(f"contact tensor has shape {C.shape}, expected {contact_shape + contact_shape}")
            return np.ascontiguousarray(C.reshape(A, A).T)
        if C.shape != (A, A):
            raise ValueError(f"contact_matrix has shape {C.shape}, expected {(A, A)}")
        return np.ascontiguousarray(C)

    def _param_meta(self, p):
        """(P, batch, seasonal) from SHAPES only -- no device-to-host copy of tensor-valued fields."""
        columns = self._strain_columns(p)
        strain_rank = 1 if self.multi_strain else 0
        batch, S = None, 1
        for _, value in columns:
            shape = tuple(getattr(value, "shape", np.shape(value)))
            if not self.multi_strain and len(shape) == 1 and shape[0] == 1:
                shape = ()
            if len(shape) == strain_rank + 1:
                batch = shape[0]
            if self.multi_strain:
                S = shape[-1]
        seas = getattr(p, "seasonality_params", None)
        seasonal = bool(self.seasonal) if self.seasonal is not None else seas is not None
        if seasonal:
            for n in ("forcing_amp", "forcing_phase", "forcing_period"):
                shape = tuple(getattr(getattr(seas, n), "shape", ()))
                if len(shape) == 1 and shape[0] > 1:
                    batch = shape[0]
        P = S * len(columns) + (3 if seasonal else 0)
        return P, batch, seasonal

    def pack(self, initial_state, p, with_params: bool = True) -> Packed:
        """Flatten (initial_state, params) into the kernel's layout.  ``with_params=False`` leaves
        ``Packed.params`` as a zero placeholder of the right shape (the differentiable path builds
        the parameter matrix on the device with :meth:`param_tensor`)."""
        if len(initial_state) != len(self.compartments):
            raise ValueError(f"{self.__name__} expects compartments {self.compartments}, got "
                             f"{len(initial_state)} arrays")
        if getattr(p, "vaccination_params", None) is not None:
            return self._pack_vaccination(initial_state, p, with_params)
        if with_params:
            params, pbatch, seasonal = self._param_matrix(p)
        else:
            P, pbatch, seasonal = self._param_meta(p)
            params = np.zeros((pbatch or 1, P))
        arrs = [_np(a) for a in initial_state]
        s = arrs[0]
        sbatch = None
        if s.ndim == self.contact_ndim + 1:
            sbatch = s.shape[0]
        elif s.ndim != self.contact_ndim:
            raise ValueError(f"compartment s has shape {s.shape}; expected rank {self.contact_ndim}")
        contact_shape = s.shape[1:] if sbatch is not None else s.shape
        A = int(np.prod(contact_shape))
        per_strain = len(self._strain_columns(p))       # beta, gamma (, sigma) (, omega) (, 3 introduction blocks)
        S = (params.shape[1] - (3 if seasonal else 0)) // per_strain if self.multi_strain else 1
        batch = pbatch if pbatch is not None else sbatch
        if pbatch is not None and sbatch is not None and pbatch != sbatch:
            raise ValueError(f"batch of parameters ({pbatch}) and of initial_state ({sbatch}) differ")
        flat, shapes = [], []
        for name, a in zip(self.compartments, arrs):
            tail = () if name == "s" else (((S,) if self.multi_strain else ()) + ((self.n_wane,) if (name == "r" and self.n_wane > 1) else ()))
            want = contact_shape + tail
            if a.shape == want:
                flat.append(a.reshape(-1))
            elif sbatch is not None and a.shape == (sbatch,) + want:
                flat.append(a.reshape(sbatch, -1))
            elif sbatch is None and a.ndim == len(want) + 1 and a.shape[1:] == want:
                sbatch = a.shape[0]
                flat.append(a.reshape(sbatch, -1))
            else:
                raise ValueError(f"compartment {name} has shape {a.shape}; expected {want} "
                                 f"(optionally with a leading batch axis)")
            shapes.append(want)
        if sbatch is not None:
            if batch is not None and batch != sbatch:
                raise ValueError("batch of parameters and of initial_state differ")
            batch = sbatch
            y0 = np.concatenate([f if f.ndim == 2 else np.broadcast_to(f, (sbatch, f.size)) for f in flat], axis=1)
            if params.shape[0] == 1 and sbatch > 1:
                params = np.ascontiguousarray(np.broadcast_to(params, (sbatch, params.shape[1])))
        else:
            y0 = np.concatenate(flat)
        masks = self._intro_masks(p, A, S)
        model = ModelDesc(n_age=A, n_strain=S, has_e=self.has_e, has_wane=self.has_wane, has_c=self.has_c,
                          n_wane=self.n_wane, normalize=self.normalize, seasonal=seasonal,
                          has_intro=bool(masks), intro_age_mask=masks)
        assert y0.shape[-1] == model.state_dim and params.shape[1] == model.param_dim
        return Packed(model, np.ascontiguousarray(y0), params, self._contact(p, A, contact_shape), batch,
                      tuple(shapes))

    def _pack_vaccination(self, initial_state, p, with_params: bool = True) -> Packed:
        """Vaccination tiers: the (age, tier) pairs become the groups of the kernel's contact axis
        (include/dynode_hip.h, "n_vax_tiers"), tier padded to 2 or 4 slots; the caller's age contact
        matrix is turned into the group matrix C[a][b] / P_b (P_b = population of age b in the initial
        state) and the model runs with normalize = 0."""
        if not self.multi_strain or self.contact_ndim != 1:
            raise ValueError(f"{self.__name__}: vaccination tiers are available for the multi-strain family")
        vp = p.vaccination_params
        # with_params = False (differentiable path): shapes only, no device-to-host copy of tensor-valued fields --
        # the parameter matrix is then built on the device by param_tensor
        get = _np if with_params else (lambda v: np.empty(tuple(getattr(v, "shape", np.shape(v)))))
        base, knots, coefs, ve = get(vp.base_equations), get(vp.knot_locations), get(vp.knot_coefficients), get(vp.vaccine_efficacy)
        A, K = base.shape[:2]
        nk = knots.shape[-1] if knots.ndim == 3 else 0
        if base.shape != (A, K, 4) or knots.shape != (A, K, nk) or coefs.shape != (A, K, nk) or not 2 <= K <= 4 or nk > 4:
            raise ValueError("vaccination splines must be base_equations [A, K, 4], knot_locations / knot_coefficients "
                             "[A, K, n_knots] with 2 <= K <= 4 tiers and n_knots <= 4")
        KV = 2 if K == 2 else 4
        if with_params:
            rates, pbatch, seasonal = self._param_matrix(p)
        else:
            n_rates, pbatch, seasonal = self._param_meta(p)
            rates = np.zeros((pbatch or 1, n_rates))
        S = (rates.shape[1] - (3 if seasonal else 0)) // len(self._strain_columns(p))
        if ve.shape[-2:] != (S, K) or ve.ndim not in (2, 3) or (with_params and (ve.min() < 0 or ve.max() > 1)):
            raise ValueError(f"vaccine_efficacy must have shape (strains, tiers) = {(S, K)}, optionally with a leading "
                             "batch axis, and values in [0, 1]")
        if ve.ndim == 3:
            if pbatch not in (None, ve.shape[0]):
                raise ValueError("inconsistent batch sizes in ode parameters (vaccine_efficacy)")
            pbatch = ve.shape[0]
            rates = np.broadcast_to(rates, (pbatch, rates.shape[1]))
        if not with_params:
            ve, base, knots, coefs = np.zeros_like(ve), np.zeros_like(base), np.zeros_like(knots), np.zeros_like(coefs)
        arrs = [_np(a).astype(np.float64) for a in initial_state]
        flat, shapes, pop = [], [], np.zeros(A)
        for name, a in zip(self.compartments, arrs):
            tail = () if name == "s" else ((S,) + ((self.n_wane,) if (name == "r" and self.n_wane > 1) else ()))
            if a.shape != (A, K) + tail:
                raise ValueError(f"compartment {name} has shape {a.shape}; expected {(A, K) + tail} (age, tier, ...)")
            padded = np.zeros((A, KV) + tail)
            padded[:, :K] = a
            flat.append(padded.reshape(-1))
            shapes.append((A, KV) + tail)
            if name != "c":
                pop += a.reshape(A, -1).sum(1)
        C = _np(p.contact_matrix)
        if C.shape != (A, A):
            raise ValueError(f"contact_matrix has shape {C.shape}, expected {(A, A)}")
        inv_pop = np.where(pop > 0, 1.0 / np.where(pop > 0, pop, 1.0), 0.0)   # an empty age group infects nobody
        Cg = np.repeat(np.repeat(C * inv_pop[None, :], KV, axis=0), KV, axis=1)
        ve3 = ve.reshape((-1, S, K))
        sus = np.ones((ve3.shape[0], A, KV, S))
        sus[:, :, :K, :] = 1.0 - np.swapaxes(ve3, 1, 2)[:, None, :, :]
        spl = np.zeros((A, KV, 4 + 2 * nk))
        spl[:, :K, :4], spl[:, :K, 4:4 + nk], spl[:, :K, 4 + nk:] = base, knots, coefs
        B = rates.shape[0]
        params = np.concatenate([rates, np.broadcast_to(sus.reshape(sus.shape[0], -1), (B, sus[0].size)),
                                 np.broadcast_to(spl.reshape(1, -1), (B, spl.size))], axis=1)
        masks = self._intro_masks(p, A, S)
        if masks:
            raise ValueError("introduced strains and vaccination tiers cannot be combined yet")
        model = ModelDesc(n_age=A * KV, n_strain=S, has_e=self.has_e, has_wane=self.has_wane, has_c=self.has_c,
                          n_wane=self.n_wane, normalize=False, seasonal=seasonal, n_vax_tiers=K, n_vax_knots=nk)
        assert params.shape[1] == model.param_dim
        return Packed(model, np.concatenate(flat), np.ascontiguousarray(params), np.ascontiguousarray(Cg), pbatch,
                      tuple(shapes), tiers=K)

    # ------------------------------------------------------------------ host evaluation
    def __call__(self, t, state, p):
        """f(t, state, p) with NumPy, unbatched -- for inspection/tests, never used by simulate."""
        pk = self.pack(state, p)
        if pk.batch is not None:
            raise ValueError("the host evaluation of an ODE descriptor is unbatched")
        m, y, q, C = pk.model, pk.y0, pk.params[0], pk.contact
        A, S, W = m.n_age, m.n_strain, m.n_wane
        pos = 0
        s = y[pos:pos + A]; pos += A
        e = None
        if m.has_e:
            e = y[pos:pos + A * S].reshape(A, S); pos += A * S
        i = y[pos:pos + A * S].reshape(A, S); pos += A * S
        r = y[pos:pos + A * S * W].reshape(A, S, W); pos += A * S * W
        beta, gamma = q[:S], q[S:2 * S]
        k = 2
        sigma = q[k * S:(k + 1) * S] if m.has_e else None
        k += int(m.has_e)
        omega = q[k * S:(k + 1) * S] if m.has_wane else None
        k += int(m.has_wane)
        pos = k * S
        intro = None
        if m.has_intro:
            intro = q[pos:pos + 3 * S].reshape(3, S); pos += 3 * S
        if m.seasonal:
            amp, phase, period = q[pos:pos + 3]; pos += 3
            beta = beta * (1.0 + amp * np.sin(2 * np.pi * t / period + phase))
        N = s + i.sum(1) + r.sum((1, 2)) + (e.sum(1) if e is not None else 0.0)
        x = i / N[:, None] if m.normalize else i
        if intro is not None:               # external introductions: a Gaussian pulse of infectious contacts
            when, scale, pct = intro
            mask = np.array([[(int(m.intro_age_mask[l]) >> b) & 1 for l in range(S)] for b in range(A)], dtype=float)
            pulse = pct * np.exp(-0.5 * ((t - when) / scale) ** 2) / (scale * np.sqrt(2 * np.pi))
            x = x + mask * pulse[None, :] * (1.0 if m.normalize else N[:, None])
        foi = beta[None, :] * (C @ x)
        doses = None
        if m.n_vax_tiers > 1:               # groups = (age, tier): susceptibility per group, doses move s up a tier
            KV, nk = m.vax_lanes, m.n_vax_knots
            foi = foi * q[pos:pos + A * S].reshape(A, S); pos += A * S
            c = q[pos:pos + A * (4 + 2 * nk)].reshape(A, 4 + 2 * nk)
            nu = c[:, 0] + t * (c[:, 1] + t * (c[:, 2] + t * c[:, 3]))
            nu = nu + (c[:, 4 + nk:] * np.maximum(t - c[:, 4:4 + nk], 0.0) ** 3).sum(1)
            tier = np.arange(A) % KV
            people = np.repeat(N.reshape(-1, KV).sum(1), KV)
            doses = np.where(tier >= m.n_vax_tiers - 1, 0.0, np.minimum(np.maximum(nu, 0.0) * people, np.maximum(s, 0.0)))
        flux = foi * s[:, None]
        g_i = gamma[None, :] * i
        ds = -flux.sum(1)
        out = []
        if e is not None:
            s_e = sigma[None, :] * e
            de, di = flux - s_e, s_e - g_i
        else:
            de, di = None, flux - g_i
        dr = np.zeros_like(r)
        if m.has_wane:
            rate = W * omega[None, :, None] * r
            dr[:, :, 0] = g_i - rate[:, :, 0]
            dr[:, :, 1:] = rate[:, :, :-1] - rate[:, :, 1:]
            ds = ds + rate[:, :, -1].sum(1)
        else:
            dr[:, :, 0] = g_i
        if doses is not None:
            ds = ds - doses + np.where(tier == 0, 0.0, np.roll(doses, 1))
        parts = [ds] + ([de] if de is not None else []) + [di, dr] + ([flux] if m.has_c else [])
        for arr, shape in zip(parts, pk.shapes):
            arr = np.asarray(arr).reshape(shape)
            out.append(arr[:, :pk.tiers] if pk.tiers is not None else arr)
        return tuple(out)


# ---------------------------------------------------------------------- the shipped RHS family
sir_ode = CompartmentalODE(
    "sir_ode", SIR_ODEParams, ("s", "i", "r"),
    doc="SIR with optional age stratification: examples/sir.py:78-84, sir_age_stratified.py:127-142.")

sir_ode_unnormalised = CompartmentalODE(
    "sir_ode", SIR_ODEParams, ("s", "i", "r"), normalize=False,
    doc="beta*s*i without the /N: the reference's tests/test_simulation/test_odes.py:17-28.")

sir_age_risk_ode = CompartmentalODE(
    "sir_ode", SIR_ODEParams, ("s", "i", "r"), contact_ndim=2,
    doc="Age x risk SIR with a 4-D contact tensor: examples/sir_age_risk_stratified.py:157-173.")

seirs_ode = CompartmentalODE(
    "seirs_ode", SEIRS_ODEParams, ("s", "e", "i", "r"), has_e=True, has_wane=True,
    doc="SEIRS: examples/seirs.py:88-95.")

seirs_ode_seasonal = CompartmentalODE(
    "seirs_ode_seasonal", SEIRS_Seasonal_ODEParams, ("s", "e", "i", "r"), has_e=True, has_wane=True,
    seasonal=True, doc="SEIRS with sinusoidal beta: examples/seirs_seasonal_forcing.py:40-55.")

seirs_multi_strain_ode = CompartmentalODE(
    "seirs_multi_strain_ode", SEIRS_MultiStrain_ODEParams, ("s", "e", "i", "r", "c"),
    multi_strain=True, has_e=True, has_wane=True, has_c=True, seasonal=None,
    doc="Age x strain SEIRS + cumulative incidence: examples/seirs_multi_strain_age_stratified.py:213-243; "
        "seasonal when p.seasonality_params is set (BASELINE cfg 5).")
