"""Shared test helpers: oracle adapters and an independent NumPy twin of the RHS family.

The NumPy twin is written from the reference formulas (vectorised, einsum), independently of
the C oracle's loops, so that oracle-vs-twin agreement pins the oracle's RHS.
"""

from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def omodel(m) -> "O.Model":
    """dynode_amd ModelDesc (or anything with the same fields) -> oracle Model."""
    return O.Model(m.n_age, m.n_strain, bool(m.has_e), bool(m.has_wane), bool(m.has_c), m.n_wane,
                   bool(m.normalize), bool(m.seasonal), bool(getattr(m, "has_intro", False)),
                   tuple(getattr(m, "intro_age_mask", ())), int(getattr(m, "n_vax_tiers", 0)),
                   int(getattr(m, "n_vax_knots", 0)))


def split_state(m, y):
    """Flat state -> dict of compartment arrays shaped like the reference ([A], [A,S], [A,S,W])."""
    A, S, W = m.n_age, m.n_strain, m.n_wane
    out, pos = {}, 0
    out["s"] = y[pos:pos + A]; pos += A
    if m.has_e:
        out["e"] = y[pos:pos + A * S].reshape(A, S); pos += A * S
    out["i"] = y[pos:pos + A * S].reshape(A, S); pos += A * S
    out["r"] = y[pos:pos + A * S * W].reshape(A, S, W); pos += A * S * W
    if m.has_c:
        out["c"] = y[pos:pos + A * S].reshape(A, S); pos += A * S
    assert pos == y.size
    return out


def split_params(m, p):
    S = m.n_strain
    out, pos = {}, 0
    out["beta"] = p[pos:pos + S]; pos += S
    out["gamma"] = p[pos:pos + S]; pos += S
    if m.has_e:
        out["sigma"] = p[pos:pos + S]; pos += S
    if m.has_wane:
        out["omega"] = p[pos:pos + S]; pos += S
    if getattr(m, "has_intro", False):
        out["intro_time"] = p[pos:pos + S]; pos += S
        out["intro_scale"] = p[pos:pos + S]; pos += S
        out["intro_pct"] = p[pos:pos + S]; pos += S
    if m.seasonal:
        out["amp"], out["phase"], out["period"] = p[pos:pos + 3]; pos += 3
    if getattr(m, "n_vax_tiers", 0) > 1:
        A, nk = m.n_age, m.n_vax_knots
        out["sus"] = p[pos:pos + A * S].reshape(A, S); pos += A * S
        out["spline"] = p[pos:pos + A * (4 + 2 * nk)].reshape(A, 4 + 2 * nk); pos += A * (4 + 2 * nk)
    assert pos == p.size
    return out


def rhs_numpy(m, t, y, p, C):
    """Vectorised restatement of the reference RHS family.

    seirs_multi_strain_age_stratified.py:213-243 generalised; with S=1, no e, no waning it is
    sir_age_stratified.py:127-142 (`beta * sum((C * i) / N, axis=1)`), with A=1 it is
    sir.py:78-84 / seirs.py:88-95; seasonal multiplier seirs_seasonal_forcing.py:40-55.
    """
    st, pr = split_state(m, np.asarray(y, dtype=np.float64)), split_params(m, np.asarray(p, dtype=np.float64))
    s, i, r = st["s"], st["i"], st["r"]
    e = st.get("e")
    W = m.n_wane
    N = s + i.sum(1) + r.sum((1, 2)) + (e.sum(1) if e is not None else 0.0)
    x = i / N[:, None] if m.normalize else i
    if getattr(m, "has_intro", False):
        # externally introduced strains (ode_model.md): I_b + Normal(t; time, scale) * pct * P_b for the masked ages
        mask = np.array([[(int(m.intro_age_mask[l]) >> b) & 1 for l in range(m.n_strain)] for b in range(m.n_age)], dtype=float)
        pulse = pr["intro_pct"] * np.exp(-0.5 * ((t - pr["intro_time"]) / pr["intro_scale"]) ** 2) / (pr["intro_scale"] * np.sqrt(2 * np.pi))
        x = x + mask * pulse[None, :] * (1.0 if m.normalize else N[:, None])
    beta = pr["beta"]
    if m.seasonal:
        beta = beta * (1.0 + pr["amp"] * np.sin(2 * np.pi * t / pr["period"] + pr["phase"]))
    foi = beta[None, :] * np.einsum("ab,bl->al", np.asarray(C, dtype=np.float64), x)
    if "sus" in pr:
        foi = foi * pr["sus"]
    flux = foi * s[:, None]
    g_i = pr["gamma"][None, :] * i
    ds = -flux.sum(1)
    if e is not None:
        s_e = pr["sigma"][None, :] * e
        de, di = flux - s_e, s_e - g_i
    else:
        de, di = None, flux - g_i
    dr = np.zeros_like(r)
    if m.has_wane:
        rate = W * pr["omega"][None, :, None] * r  # outflow of every stage
        dr[:, :, 0] = g_i - rate[:, :, 0]
        dr[:, :, 1:] = rate[:, :, :-1] - rate[:, :, 1:]
        ds = ds + rate[:, :, -1].sum(1)
    else:
        dr[:, :, 0] = g_i
    if "spline" in pr:                 # vaccination: tier k -> k + 1 within each age (groups = age * KV + tier)
        nk, KV = m.n_vax_knots, (2 if m.n_vax_tiers <= 2 else 4)
        c = pr["spline"]
        nu = c[:, 0] + c[:, 1] * t + c[:, 2] * t**2 + c[:, 3] * t**3
        if nk:
            lag = np.maximum(t - c[:, 4:4 + nk], 0.0)
            nu = nu + (c[:, 4 + nk:4 + 2 * nk] * lag**3).sum(1)
        n_age = np.repeat(N.reshape(-1, KV).sum(1), KV)
        tier = np.arange(m.n_age) % KV
        leave = np.where(tier >= m.n_vax_tiers - 1, 0.0, np.minimum(np.maximum(nu, 0.0) * n_age, np.maximum(s, 0.0)))
        arrive = np.where(tier == 0, 0.0, np.roll(leave, 1))
        ds = ds + arrive - leave
    parts = [ds]
    if de is not None:
        parts.append(de.ravel())
    parts += [di.ravel(), dr.ravel()]
    if m.has_c:
        parts.append(flux.ravel())
    return np.concatenate(parts)


def ground_truth(m, y0, p, C, t1, save_ts, rtol=1e-12, atol=1e-12):
    """fp64 reference trajectory from scipy's DOP853 (independent integrator)."""
    from scipy.integrate import solve_ivp

    sol = solve_ivp(lambda t, y: rhs_numpy(m, t, y, p, C), (0.0, float(t1)), np.asarray(y0, float),
                    method="DOP853", t_eval=np.asarray(save_ts, float), rtol=rtol, atol=atol)
    assert sol.success
    return sol.y.T  # [n_save, D]
