"""Synthetic epidemic ensembles for the BASELINE.json configs (recipes: SURVEY.md 8d).

Pure numpy, deterministic per seed; used by bench.py, the smoke test and the parity tests.
Every generator returns a :class:`Workload` whose arrays are float64 on the host; callers
cast to the solve dtype.  Literal constants cite the reference example they come from.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from ._abi import ModelDesc


@dataclass
class Workload:
    name: str
    model: ModelDesc
    y0: np.ndarray        # [D] or [B, D]
    params: np.ndarray    # [B, P]
    contact: np.ndarray   # [A, A]
    t1: float
    save_ts: np.ndarray   # [n_save]
    population: float     # scale of the state, for norm-wise error reporting

    @property
    def B(self) -> int:
        return self.params.shape[0]

    @property
    def n_save(self) -> int:
        return self.save_ts.shape[0]

    def bytes_per_trajectory(self, itemsize: int = 4, d_saved: int | None = None) -> int:
        """Algorithmic HBM bytes per trajectory: w * (P + [y0 if batched] + n_save * D_saved)."""
        d = self.model.state_dim if d_saved is None else d_saved
        y0 = self.model.state_dim if self.y0.ndim == 2 else 0
        return itemsize * (self.model.param_dim + y0 + self.n_save * d)


def save_grid(t1: float, step: int = 1) -> np.ndarray:
    """linspace(0, T, T//step + 1) -- reference odes.py:177-180 (not arange)."""
    if step <= 0:
        step = 1
    return np.linspace(0.0, t1, int(t1 // step) + 1)


def contact_matrix(rng: np.random.Generator, A: int) -> np.ndarray:
    """Symmetric positive matrix normalised by its spectral radius (sir_age_stratified.py:81-85)."""
    M = rng.uniform(0.05, 1.0, (A, A))
    C = 0.5 * (M + M.T) + 2.0 * np.eye(A)
    return C / np.max(np.real(np.linalg.eigvals(C)))


def _trunc_normal(rng, loc, scale, low, high, size):
    out = rng.normal(loc, scale, size)
    bad = (out < low) | (out > high)
    while bad.any():
        out[bad] = rng.normal(loc, scale, int(bad.sum()))
        bad = (out < low) | (out > high)
    return out


def sir_literal() -> Workload:
    """cfg 1(i): examples/sir.py literal -- 1 bin, y0=(0.9,0.1,0), r0=2, T_inf=7 (sir.py:34,43,90-91)."""
    model = ModelDesc(n_age=1)
    return Workload("sir_literal", model, np.array([0.9, 0.1, 0.0]),
                    np.array([[2.0 / 7.0, 1.0 / 7.0]]), np.array([[1.0]]), 150.0,
                    save_grid(150.0), 1.0)


def sir_two_age_literal(t1: float = 100.0, r0: float = 2.0, infectious_period: float = 7.0) -> Workload:
    """cfg 1(ii): the 2-age model of examples/sir_age_stratified.py:46-66,70,81-85,118-119."""
    model = ModelDesc(n_age=2)
    demo = np.array([0.75, 0.25])
    y0 = np.concatenate([1000 * 0.99 * demo, 1000 * 0.01 * demo, np.zeros(2)])
    C = np.array([[0.7, 0.3], [0.3, 0.7]])
    C = C / np.max(np.real(np.linalg.eigvals(C)))
    params = np.array([[r0 / infectious_period, 1.0 / infectious_period]])
    return Workload("sir_two_age_literal", model, y0, params, C, t1, save_grid(t1), 1000.0)


def sir_age_stratified(B: int = 4096, seed: int = 0, A: int = 8, t1: float = 365.0) -> Workload:
    """cfg 2: A-age SIR, priors of examples/sir_infer_parameters.py:50-56, shared y0 and C."""
    rng = np.random.default_rng(seed)
    w = rng.dirichlet(5.0 * np.ones(A))
    C = contact_matrix(rng, A)
    r0 = 1.5 + rng.beta(0.5, 0.5, B)
    t_inf = _trunc_normal(rng, 8.0, 2.0, 2.0, 15.0, B)
    params = np.stack([r0 / t_inf, 1.0 / t_inf], axis=1)
    y0 = np.concatenate([0.99 * 1000 * w, 0.01 * 1000 * w, np.zeros(A)])
    return Workload("sir_age_stratified", ModelDesc(n_age=A), y0, params, C, t1, save_grid(t1),
                    1000.0)


def seirs_multi_strain(B: int = 16384, seed: int = 1, A: int = 8, S: int = 4, W: int = 1,
                       seasonal: bool = False, t1: float = 365.0) -> Workload:
    """cfg 3 (and cfg 5 with seasonal=True): A-age x S-strain SEIRS + cumulative incidence.

    RHS/initializer pattern: examples/seirs_multi_strain_age_stratified.py:146-172,213-243;
    parameter ranges bracket the literals at :46-49; seasonal literals from
    examples/seirs_seasonal_forcing.py:61-63.  W > 1 is the build-defined Erlang waning chain.
    """
    rng = np.random.default_rng(seed)
    w = rng.dirichlet(5.0 * np.ones(A))
    C = contact_matrix(rng, A)
    r0 = rng.uniform(1.8, 2.8, (B, S))
    t_inf = rng.uniform(5.0, 9.0, (B, S))
    t_lat = rng.uniform(2.0, 4.0, (B, S))
    t_wane = rng.uniform(50.0, 90.0, (B, S))
    cols = [r0 / t_inf, 1.0 / t_inf, 1.0 / t_lat, 1.0 / t_wane]
    if seasonal:
        cols += [rng.uniform(0.0, 0.4, (B, 1)), rng.uniform(0.0, 2 * np.pi, (B, 1)),
                 np.full((B, 1), 365.0)]
    params = np.concatenate(cols, axis=1)
    model = ModelDesc(n_age=A, n_strain=S, has_e=True, has_wane=True, has_c=True, n_wane=W,
                      seasonal=seasonal)
    D = model.state_dim
    y0 = np.zeros((B, D))
    y0[:, :A] = 1000 * 0.99 * w
    dom = r0 / r0.sum(axis=1, keepdims=True)
    off_i = A + A * S
    y0[:, off_i:off_i + A * S] = (1000 * 0.01 * w[None, :, None] * dom[:, None, :]).reshape(B, -1)
    name = "seirs_seasonal_forcing_multi_strain" if seasonal else "seirs_multi_strain_age_stratified"
    return Workload(name, model, y0, params, C, t1, save_grid(t1), 1000.0)


def seip_protection_table(*args, **kwargs) -> np.ndarray:
    """``dynode_amd.seip.protection_table`` (ode_model.md:185-211)."""
    from .seip import protection_table

    return protection_table(*args, **kwargs)


def seip(B: int = 4096, seed: int = 7, A: int = 8, L: int = 2, K1: int = 3, M1: int = 4, n_knots: int = 2,
         seasonal: bool = False, seasonal_vax: bool = False, t1: float = 365.0, intro: bool = False) -> Workload:
    """SEIP ensemble (ode_model.md; include/dynode_hip.h "SEIP"): A ages x 2^L immune histories x K1 vaccination
    tiers x M1 waning states, L strains.  Rates as cfg 3; everyone starts unexposed and unvaccinated in the
    last waning state; doses start between day 20 and 120 at 0.2-1 % of the age group per day; cross-immunity
    0.4-0.9, vaccine efficacy rising with doses, protection falling over the waning states."""
    rng = np.random.default_rng(seed)
    H = 1 << L
    masks = tuple(int(v) for v in rng.integers(1, 1 << A, L)) if intro else ()   # ages that receive each strain's visitors
    model = ModelDesc(n_age=A, n_strain=L, has_e=True, has_wane=True, has_c=True, n_wane=M1, normalize=False,
                      seasonal=seasonal, n_vax_tiers=K1, n_vax_knots=n_knots, family=1, seasonal_vax=seasonal_vax,
                      has_intro=intro, intro_age_mask=masks)
    w = rng.dirichlet(5.0 * np.ones(A))
    pop = 1000.0 * w
    C = contact_matrix(rng, A) / pop[None, :]                          # lambda_a = beta sum_b C_ab I_b / P_b
    r0 = rng.uniform(1.8, 2.8, (B, L))
    t_inf, t_lat = rng.uniform(5.0, 9.0, (B, L)), rng.uniform(2.0, 4.0, (B, L))
    omega = 1.0 / rng.uniform(20.0, 60.0, (B, M1))
    cols = [r0 / t_inf, 1.0 / t_inf, 1.0 / t_lat, omega]
    if intro:     # visitors around day 20-100, spread over 3-10 days, worth 0.1-1 % of the receiving age groups
        cols += [rng.uniform(20.0, 100.0, (B, L)), rng.uniform(3.0, 10.0, (B, L)), rng.uniform(0.001, 0.01, (B, L))]
    if seasonal:
        cols += [rng.uniform(0.0, 0.4, (B, 1)), rng.uniform(0.0, 2 * np.pi, (B, 1)), np.full((B, 1), 365.0)]
    if seasonal_vax:
        cols.append(rng.uniform(100.0, 260.0, (B, 1)))                 # tau = 182.5 - days to the season change
    cols.append(np.broadcast_to(pop, (B, A)))
    prot = np.linspace(1.0, 0.0, M1) if M1 > 1 else np.ones(1)
    sus = np.empty((B, H * K1 * M1 * L))
    for b in range(B):
        chi = rng.uniform(0.4, 0.9, (L, L))
        np.fill_diagonal(chi, 1.0)
        ve = np.sort(rng.uniform(0.0, 0.7, (L, K1)), axis=1)
        ve[:, 0] = 0.0
        sus[b] = seip_protection_table(chi, ve, prot, rng.uniform(0.0, 0.3)).ravel()
    cols.append(sus)
    spl = np.zeros((B, A, K1, 4 + 2 * n_knots))
    if n_knots:
        start = rng.uniform(20.0, 120.0, (B, A, K1, 1)) + 30.0 * np.arange(K1)[None, None, :, None]
        ramp = rng.uniform(10.0, 30.0, (B, A, K1, 1))
        plateau = rng.uniform(0.002, 0.01, (B, A, K1))
        spl[..., 4:4 + n_knots] = start + ramp * np.arange(n_knots)[None, None, None, :]
        coef = plateau / (ramp[..., 0] ** 3 * 6.0)
        spl[..., 4 + n_knots] = coef                                   # cubic rise from the first knot ...
        if n_knots > 1:
            spl[..., 4 + n_knots + 1] = -coef                          # ... turned into a quadratic one at the second
    else:
        spl[..., 0] = rng.uniform(0.0, 0.004, (B, A, K1))
    cols.append(spl.reshape(B, -1))
    params = np.concatenate(cols, axis=1)
    assert params.shape[1] == model.param_dim
    s0 = np.zeros((B, A, H, K1, M1))
    i0 = np.zeros((B, A, H, K1, L))
    dom = r0 / r0.sum(axis=1, keepdims=True)
    s0[:, :, 0, 0, M1 - 1] = 0.99 * pop
    i0[:, :, 0, 0, :] = 0.01 * pop[None, :, None] * dom[:, None, :]
    zeros = np.zeros_like(i0).reshape(B, -1)
    y0 = np.concatenate([s0.reshape(B, -1), zeros, i0.reshape(B, -1), zeros], axis=1)
    return Workload("seip", model, y0, params, C, t1, save_grid(t1), 1000.0)


WORKLOADS = {
    "cfg2": lambda B=4096, seed=0: sir_age_stratified(B, seed),
    # BASELINE.json cfg 3 as worded: 8 age x 4 strain x 8 immunity bins (Erlang waning chain, D = 360)
    "cfg3": lambda B=16384, seed=1: seirs_multi_strain(B, seed, W=8),
    # the reference's own RHS at that size (seirs_multi_strain_age_stratified.py:213-243, no bins axis): D = 136
    "cfg3d136": lambda B=16384, seed=1: seirs_multi_strain(B, seed),
    "seip": lambda B=4096, seed=7: seip(B, seed),
    "seip3": lambda B=4096, seed=7: seip(B, seed, A=4, L=3),      # three strains: tiers dealt over two lanes
    # the north star's age x strain x immune-history sizes: lane groups of 128 / 256 = workgroups of 2 / 4 waves per trajectory
    "seip83": lambda B=4096, seed=7: seip(B, seed, A=8, L=3),     # 8 ages x 3 strains (8 histories): D = 2496
    "seip84": lambda B=2048, seed=7: seip(B, seed, A=8, L=4),     # 8 ages x 4 strains (16 histories): D = 6144
    "cfg5": lambda B=8192, seed=5: seirs_multi_strain(B, seed, seasonal=True),
}
