"""The SEIP model of the reference's ode_model.md behind the ODE-descriptor interface.

The reference documents its production model -- susceptible / exposed / infectious / partially immune,
stratified by age, immune history, vaccination count and waning state (ode_model.md:15-53, 70-105, 176-211)
-- and ships the configuration classes for it (``FullStratifiedImmuneHistoryDimension``,
``VaccinationDimension``, ``WaneDimension``/``WaneBin``, ``Strain.vaccine_efficacy``,
``TransmissionParams.strain_interactions``), but no right-hand side.  Here it is a member of the kernel
family (``family = 1``; include/dynode_hip.h "SEIP", csrc/seip_kernel.hpp):

    state        s[A, H, K1, M1]    e, i, c[A, H, K1, L]        H = 2^L immune histories
    simulate(seip_ode, days, (s, e, i, c), SEIP_ODEParams(...), SolverParams())

The history axis follows the reference's bin order (``none, a, b, c, a_b, a_c, b_c, a_b_c``:
config/dimension.py:152-171); the kernel indexes histories by bit set and the front-end permutes.
"""

from __future__ import annotations

from dataclasses import dataclass
from itertools import combinations
from types import SimpleNamespace
from typing import Any, Optional

import numpy as np

from ._abi import ModelDesc
from .rhs import (AbstractODEParams, CompartmentalODE, IntroductionParams, Packed, SeasonalityParams, VaccinationParams,
                  _np)


def history_masks(n_strains: int) -> np.ndarray:
    """Bit set of every bin of ``FullStratifiedImmuneHistoryDimension`` in the reference's order: subsets
    by size, then by strain order (strain q = bit q)."""
    masks = [0] + [sum(1 << q for q in sub) for size in range(1, n_strains + 1) for sub in combinations(range(n_strains), size)]
    return np.asarray(masks, dtype=np.int64)


def protection_table_torch(crossimmunity, vaccine_efficacy, wane_protection, min_homologous_immunity=0.0):
    """:func:`protection_table` with torch ops (same formulas, same index order): a cross-immunity or vaccine efficacy that
    is a tensor requiring grad keeps its autograd graph, so the susceptibility table can be differentiated -- the solve
    itself through `engine._replayed_tangents`."""
    import torch

    f64 = torch.float64
    dev = next((v.device for v in (crossimmunity, vaccine_efficacy, wane_protection, min_homologous_immunity)
                if isinstance(v, torch.Tensor)), torch.device("cpu"))
    chi, ve, prot, floor = (v.to(device=dev, dtype=f64) if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v, dtype=float), device=dev)
                            for v in (crossimmunity, vaccine_efficacy, wane_protection, min_homologous_immunity))
    L, K1, M1 = chi.shape[-1], ve.shape[-1], prot.shape[-1]
    lead = torch.broadcast_shapes(chi.shape[:-2], ve.shape[:-2], prot.shape[:-1], floor.shape)
    rows = []
    for j in range(1 << L):
        past = [q for q in range(L) if (j >> q) & 1]
        cols = []
        for l in range(L):
            escape = torch.ones(lead, dtype=f64, device=dev)
            for q in past:
                escape = escape * (1.0 - chi[..., l, q])
            initial = 1.0 - (1.0 - ve[..., l, :]) * escape[..., None]                          # [..., K1]
            wib = initial[..., :, None] * prot[..., None, :]                                    # [..., K1, M1]
            wim = (1.0 - wib) * floor[..., None, None] if (j >> l) & 1 else torch.zeros_like(wib)
            cols.append((1.0 - (wib + wim)).expand(lead + (K1, M1)))
        rows.append(torch.stack(cols, dim=-1))                                                   # [..., K1, M1, L]
    return torch.stack(rows, dim=-4)                                                             # [..., H, K1, M1, L]


def protection_table(crossimmunity, vaccine_efficacy, wane_protection, min_homologous_immunity=0.0) -> np.ndarray:
    """Susceptibility ``1 - WI`` of ode_model.md:185-211, indexed [history bit set, doses, waning state, strain].

    For a susceptible with past strains ``j``, ``k`` doses, in waning state ``m``, challenged by strain ``l``:
    initial immunity ``II = 1 - (1 - ve[l, k]) * prod_{q in j} (1 - chi[l, q])`` (cross-immunity
    ``strain_interactions`` and ``Strain.vaccine_efficacy``), waned ``WIB = II * protection[m]``
    (``WaneBin.base_protection``), homologous floor ``WIM = (1 - WIB) * M_HI`` if ``l`` is in ``j``;
    ``WI = WIB + WIM``."""
    chi, ve, prot = (np.asarray(v, dtype=float) for v in (crossimmunity, vaccine_efficacy, wane_protection))
    floor = np.asarray(min_homologous_immunity, dtype=float)
    L, K1, M1 = chi.shape[-1], ve.shape[-1], prot.shape[-1]
    # leading axes (one table per parameter sample) broadcast: chi[..., L, L], ve[..., L, K1], prot[..., M1], floor[...]
    lead = np.broadcast_shapes(chi.shape[:-2], ve.shape[:-2], prot.shape[:-1], floor.shape)
    sus = np.empty(lead + (1 << L, K1, M1, L))
    for j in range(1 << L):
        past = [q for q in range(L) if (j >> q) & 1]
        for l in range(L):
            escape = np.prod([1.0 - chi[..., l, q] for q in past], axis=0) if past else 1.0
            initial = 1.0 - (1.0 - ve[..., l, :]) * np.asarray(escape)[..., None]            # [..., K1]
            wib = initial[..., :, None] * prot[..., None, :]                                  # [..., K1, M1]
            wim = (1.0 - wib) * (floor[..., None, None] if (j >> l) & 1 else 0.0)
            sus[..., j, :, :, l] = 1.0 - (wib + wim)
    return sus


@dataclass
class SEIP_ODEParams(AbstractODEParams):
    """Parameters of :data:`seip_ode`.  Rates may carry a leading batch axis."""

    beta: Any                   # [L] or [B, L]   transmission rate per strain
    gamma: Any                  # [L]             recovery rate
    sigma: Any                  # [L]             exposed -> infectious
    waning_rates: Any           # [M1]            1 / WaneBin.waiting_time; the last state keeps its people
    contact_matrix: Any         # [A, A]          lambda_a = beta * sum_b C[a, b] * I_b / P_b
    susceptibility: Any         # [H, K1, M1, L]  by history BIT SET (protection_table), or [B, ...]
    vaccination_params: Optional[VaccinationParams] = None   # splines [A, K1, ...]; vaccine_efficacy is unused here
    seasonality_params: Optional[SeasonalityParams] = None
    seasonal_vaccination_tau: Optional[float] = None         # tau = 182.5 - days to the season change; None = off
    population: Any = None      # [A]; default: the initial state's population per age
    introduction_params: Optional[IntroductionParams] = None   # externally introduced strains (rhs.introduction_params)
    idx: Optional[SimpleNamespace] = None


class SEIPODE(CompartmentalODE):
    """Descriptor of the SEIP right-hand side (compartments s, e, i, c)."""

    def __init__(self):
        super().__init__("seip_ode", SEIP_ODEParams, ("s", "e", "i", "c"), multi_strain=True, has_e=True, has_wane=True,
                         has_c=True, seasonal=None, normalize=False,
                         doc="SEIP with immune histories, vaccination tiers and waning states: ode_model.md:15-53.")

    _GRAD_FIELDS = ("beta", "gamma", "sigma", "waning_rates", "susceptibility")

    def wants_grad(self, p) -> bool:
        """Rates or the susceptibility table given as tensors that require grad: the solve is then differentiated along
        them (central differences of replayed solves on the primal's step sequence, `engine._replayed_tangents`)."""
        import torch

        return any(isinstance(getattr(p, f), torch.Tensor) and getattr(p, f).requires_grad for f in self._GRAD_FIELDS)

    def param_tensor(self, p, device, packed: Optional[Packed] = None):
        """[B, P] parameter matrix with the autograd graph of the tensor-valued fields: the columns of `pack` (host
        constants) with the blocks of beta / gamma / sigma / waning_rates / susceptibility replaced by the tensors."""
        import torch

        if packed is None:
            raise ValueError("seip_ode.param_tensor needs the packed call (simulate passes it)")
        base = torch.as_tensor(packed.params, dtype=torch.float64, device=device)
        B = base.shape[0]
        A, L, H, K1, M1, _ = packed.model.seip_dims
        m = packed.model
        sus_at = 3 * L + M1 + (3 * L if m.has_intro else 0) + (3 if m.seasonal else 0) + (1 if m.seasonal_vax else 0) + A
        blocks = {"beta": (0, L), "gamma": (L, L), "sigma": (2 * L, L), "waning_rates": (3 * L, M1), "susceptibility": (sus_at, H * K1 * M1 * L)}
        pieces, pos = [], 0
        for name in self._GRAD_FIELDS:
            start, width = blocks[name]
            v = getattr(p, name)
            if isinstance(v, torch.Tensor) and v.requires_grad:
                pieces.append(base[:, pos:start])
                t = v.to(device=device, dtype=torch.float64)
                t = t.reshape(-1, width) if name != "susceptibility" or t.dim() == 5 else t.reshape(1, width)
                pieces.append(t.expand(B, width) if t.shape[0] == 1 else t)
                pos = start + width
        pieces.append(base[:, pos:])
        out = torch.cat(pieces, dim=1)
        assert out.shape == base.shape
        return out

    def pack(self, initial_state, p, with_params: bool = True) -> Packed:
        if len(initial_state) != 4:
            raise ValueError(f"seip_ode expects compartments {self.compartments}, got {len(initial_state)} arrays")
        s, e, i, c = (_np(a) for a in initial_state)
        if s.ndim != 4 or e.ndim != 4:
            raise ValueError("seip_ode state: s[age, history, doses, waning state], e / i / c[age, history, doses, strain]")
        A, H, K1, M1 = s.shape
        L = e.shape[3]
        if H != 1 << L or e.shape != (A, H, K1, L) or i.shape != e.shape or c.shape != e.shape:
            raise ValueError(f"seip_ode state shapes {s.shape}, {e.shape}, {i.shape}, {c.shape}: expected s{(A, 1 << L, K1, M1)} "
                             f"and e, i, c{(A, 1 << L, K1, L)} with 2^strains immune histories")
        masks = history_masks(L)                    # reference bin r holds bit set masks[r]
        inv = np.argsort(masks)                     # kernel slot j holds reference bin inv[j]
        rates, batch = [], None
        for name, width in (("beta", L), ("gamma", L), ("sigma", L), ("waning_rates", M1)):
            a = _np(getattr(p, name))
            if a.shape[-1:] != (width,) or a.ndim not in (1, 2):
                raise ValueError(f"{name} must have shape ({width},) or (batch, {width}), got {a.shape}")
            if a.ndim == 2:
                if batch not in (None, a.shape[0]):
                    raise ValueError(f"inconsistent batch sizes in ode parameters ({name})")
                batch = a.shape[0]
            rates.append(a)
        sus = _np(p.susceptibility)
        if sus.shape[-4:] != (H, K1, M1, L) or sus.ndim not in (4, 5) or sus.min() < 0:
            raise ValueError(f"susceptibility must have shape {(H, K1, M1, L)} (optionally batched) with values >= 0")
        if sus.ndim == 5:
            if batch not in (None, sus.shape[0]):
                raise ValueError("inconsistent batch sizes in ode parameters (susceptibility)")
            batch = sus.shape[0]
        B = batch or 1
        cols = [np.broadcast_to(a.reshape(-1, a.shape[-1]), (B, a.shape[-1])) for a in rates]
        intro, masks_intro = p.introduction_params, ()
        if intro is not None:
            for name in ("time", "scale", "percentage"):
                a = _np(getattr(intro, name))
                if a.shape != (L,):
                    raise ValueError(f"introduction {name} must have shape {(L,)}")
                cols.append(np.broadcast_to(a, (B, L)))
            masks_intro = self._intro_masks(p, A, L)
        seas = p.seasonality_params
        if seas is not None:
            for n in ("forcing_amp", "forcing_phase", "forcing_period"):      # a scalar, or one value per trajectory
                a = _np(getattr(seas, n)).reshape(-1)
                if a.size not in (1, B):
                    raise ValueError(f"seasonality_params.{n} has {a.size} values; expected a scalar or one per trajectory ({B})")
                cols.append(np.broadcast_to(a.reshape(-1, 1), (B, 1)))
        if p.seasonal_vaccination_tau is not None:
            cols.append(np.full((B, 1), float(p.seasonal_vaccination_tau)))
        pop = (s.sum((1, 2, 3)) + e.sum((1, 2, 3)) + i.sum((1, 2, 3))) if p.population is None else _np(p.population)
        if pop.shape != (A,):
            raise ValueError(f"population must have shape {(A,)}")
        cols.append(np.broadcast_to(pop, (B, A)))
        cols.append(np.broadcast_to(sus.reshape(-1, H * K1 * M1 * L), (B, H * K1 * M1 * L)))
        vp = p.vaccination_params
        nk = 0
        if vp is not None:
            base, knots, coefs = _np(vp.base_equations), _np(vp.knot_locations), _np(vp.knot_coefficients)
            nk = knots.shape[-1] if knots.ndim == 3 else 0
            if base.shape != (A, K1, 4) or knots.shape != (A, K1, nk) or coefs.shape != (A, K1, nk) or nk > 4:
                raise ValueError(f"vaccination splines: base_equations {(A, K1, 4)}, knot_locations / knot_coefficients "
                                 f"{(A, K1)} + (n_knots <= 4,)")
            spl = np.concatenate([base, knots, coefs], axis=2)
        else:
            spl = np.zeros((A, K1, 4))
        cols.append(np.broadcast_to(spl.reshape(1, -1), (B, spl.size)))
        C = _np(p.contact_matrix)
        if C.shape != (A, A):
            raise ValueError(f"contact_matrix has shape {C.shape}, expected {(A, A)}")
        inv_pop = np.where(pop > 0, 1.0 / np.where(pop > 0, pop, 1.0), 0.0)   # an empty age group infects nobody
        model = ModelDesc(n_age=A, n_strain=L, has_e=True, has_wane=True, has_c=True, n_wane=M1, normalize=False,
                          seasonal=seas is not None, n_vax_tiers=K1, n_vax_knots=nk, family=1,
                          seasonal_vax=p.seasonal_vaccination_tau is not None, has_intro=intro is not None,
                          intro_age_mask=masks_intro)
        params = np.ascontiguousarray(np.concatenate(cols, axis=1))
        assert params.shape[1] == model.param_dim
        y0 = np.concatenate([a[:, inv].reshape(-1) for a in (s, e, i, c)])
        return Packed(model, y0, params, np.ascontiguousarray(C * inv_pop[None, :]), batch,
                      ((A, H, K1, M1),) + ((A, H, K1, L),) * 3, history_perm=masks)

    def __call__(self, t, state, p):
        """f(t, state, p) with NumPy, unbatched (inspection and tests; simulate never uses it)."""
        pk = self.pack(state, p)
        if pk.batch is not None:
            raise ValueError("the host evaluation of an ODE descriptor is unbatched")
        m, q, C = pk.model, pk.params[0], pk.contact
        A, L, H, K1, M1, nk = m.seip_dims
        K = K1 - 1
        s, e, i, _ = (pk.y0[a:b].reshape(shape) for (a, b), shape in zip(_bounds(pk.shapes), pk.shapes))
        beta, gamma, sigma, omega = q[:L], q[L:2 * L], q[2 * L:3 * L], q[3 * L:3 * L + M1]
        pos = 3 * L + M1
        season = phi = visitors = None
        if m.has_intro:
            when, scale, pct = q[pos:pos + 3 * L].reshape(3, L); pos += 3 * L
            mask = np.array([[(int(m.intro_age_mask[l]) >> b) & 1 for l in range(L)] for b in range(A)], dtype=float)
            visitors = mask * (pct * np.exp(-0.5 * ((t - when) / scale) ** 2) / (scale * np.sqrt(2 * np.pi)))[None, :]
        if m.seasonal:
            season = 1.0 + q[pos] * np.sin(2 * np.pi * t / q[pos + 2] + q[pos + 1]); pos += 3
        if m.seasonal_vax:
            phi = np.sin(2 * np.pi * (t + q[pos]) / 730.0) ** 1000; pos += 1
        pop = q[pos:pos + A]; pos += A
        sus = q[pos:pos + H * K1 * M1 * L].reshape(H, K1, M1, L); pos += H * K1 * M1 * L
        spl = q[pos:].reshape(A, K1, 4 + 2 * nk)
        infectious = i.sum((1, 2)) + (visitors * pop[:, None] if visitors is not None else 0.0)
        lam = beta * (season if season is not None else 1.0) * (C @ infectious)
        infect = lam[:, None, None, None, :] * sus[None] * s[..., None]
        ds, inflow = -infect.sum(-1), infect.sum(3)
        wane = omega * s
        wane[..., -1] = 0.0
        ds -= wane
        ds[..., 1:] += wane[..., :-1]
        nu = spl[..., 0] + t * (spl[..., 1] + t * (spl[..., 2] + t * spl[..., 3]))
        nu = nu + (spl[..., 4 + nk:] * np.maximum(t - spl[..., 4:4 + nk], 0.0) ** 3).sum(-1)
        tot = s.sum((1, 3))
        rate = np.where(tot > 0, np.minimum(np.maximum(nu, 0.0) * pop[:, None] / np.where(tot > 0, tot, 1.0), 1.0), 0.0)
        vax = rate[:, None, :, None] * s
        vax[:, :, K, 0] = 0.0
        ds -= vax
        ds[:, :, 1:, 0] += vax[:, :, :K].sum(-1)
        ds[:, :, K, 0] += vax[:, :, K].sum(-1)
        s_e, g_i = sigma * e, gamma * i
        de, di, dc = inflow - s_e, s_e - g_i, inflow.copy()
        for l in range(L):
            for j in range(H):
                ds[:, j | (1 << l), :, 0] += g_i[:, j, :, l]
        if phi is not None and K > 0:
            for arr, darr in ((s, ds), (e, de), (i, di)):
                darr[:, :, K] -= phi * arr[:, :, K]
                darr[:, :, K - 1] += phi * arr[:, :, K]
        return tuple(d[:, pk.history_perm] for d in (ds, de, di, dc))


def params_from_config(config, vaccination_params: Optional[VaccinationParams] = None, min_homologous_immunity: float = 0.0,
                       seasonality_params: Optional[SeasonalityParams] = None, seasonal_vaccination_tau: Optional[float] = None,
                       compartment: str = "s") -> SEIP_ODEParams:
    """:class:`SEIP_ODEParams` from a ``SimulationConfig`` whose ``s`` compartment is stratified by
    (age, immune history, vaccination, waning) -- the mapping examples/seip_immune_history.py spells out:

    * ``beta = r0 / infectious_period``, ``gamma = 1 / infectious_period``, ``sigma = 1 / exposed_to_infectious`` per strain;
    * waning rates ``1 / WaneBin.waiting_time`` (``inf`` -> 0), protections ``WaneBin.base_protection``;
    * cross-immunity ``strain_interactions[challenger][past strain]`` and ``Strain.vaccine_efficacy[doses]`` folded into the
      susceptibility table by :func:`protection_table` (ode_model.md:185-211);
    * externally introduced strains (``Strain.is_introduced``) through ``rhs.introduction_params``.
    Values must be resolved numbers (call ``sample_then_resolve`` first when the config carries priors)."""
    import math

    from .rhs import introduction_params

    tp = config.parameters.transmission_params
    strains = tp.strains
    names = [s.strain_name for s in strains]
    dims = config.get_compartment(compartment).dimensions
    if len(dims) != 4:
        raise ValueError(f"compartment {compartment!r} must be stratified by (age, immune history, vaccination, waning)")
    n_tiers, wane_bins = len(dims[2]), dims[3].bins
    if not all(hasattr(b, "waiting_time") and hasattr(b, "base_protection") for b in wane_bins):
        raise ValueError(f"compartment {compartment!r} must be stratified by (age, immune history, vaccination, waning): "
                         "its last dimension has no WaneBin bins")
    if len(dims[1]) != 1 << len(strains):
        raise ValueError(f"the immune-history dimension has {len(dims[1])} bins; the SEIP model needs all 2^{len(strains)} subsets "
                         "(FullStratifiedImmuneHistoryDimension)")
    r0 = np.array([float(s.r0) for s in strains])
    t_inf = np.array([float(s.infectious_period) for s in strains])
    t_lat = np.array([float(s.exposed_to_infectious) for s in strains])
    chi = np.array([[float(tp.strain_interactions[a][b]) for b in names] for a in names])
    ve = np.array([[float((s.vaccine_efficacy or {}).get(k, 0.0)) for k in range(n_tiers)] for s in strains])
    sus = protection_table(chi, ve, [float(b.base_protection) for b in wane_bins], min_homologous_immunity)
    init = getattr(config.initializer, "initialize_date", None)
    return SEIP_ODEParams(
        beta=r0 / t_inf, gamma=1.0 / t_inf, sigma=1.0 / t_lat,
        waning_rates=np.array([0.0 if math.isinf(b.waiting_time) else 1.0 / float(b.waiting_time) for b in wane_bins]),
        contact_matrix=tp.contact_matrix, susceptibility=sus, vaccination_params=vaccination_params,
        seasonality_params=seasonality_params, seasonal_vaccination_tau=seasonal_vaccination_tau,
        introduction_params=introduction_params(strains, init), idx=config.idx)


def _bounds(shapes):
    pos = 0
    for shape in shapes:
        n = int(np.prod(shape))
        yield pos, pos + n
        pos += n


seip_ode = SEIPODE()
