"""Vaccination tiers through the front-end: VaccinationParams -> pack -> kernel parameter layout.

CPU: the packing, the host evaluation against the independent NumPy twin, the oracle against SciPy, and the
limits that tie the tiered model back to the plain one.  GPU: `simulate` on the example model.
"""

import numpy as np
import pytest

import helpers as H
from helpers import O

from dynode_amd.rhs import SEIRS_MultiStrain_ODEParams, VaccinationParams, seirs_multi_strain_ode
from examples import seirs_vaccination as ex


def _packed(cfg=None, params=None):
    cfg = cfg or ex.get_config()
    params = params or ex.get_odeparams(cfg)
    return cfg, params, seirs_multi_strain_ode.pack(cfg.initializer.get_initial_state(cfg), params)


def test_pack_layout_groups_contact_and_padding():
    cfg, p, pk = _packed()
    m = pk.model
    assert (m.n_age, m.n_strain, m.n_vax_tiers, m.n_vax_knots, m.vax_lanes, m.normalize) == (12, 2, 3, 3, 4, False)
    assert pk.tiers == 3 and pk.params.shape == (1, m.param_dim) and pk.contact.shape == (12, 12)
    pr = H.split_params(m, pk.params[0])
    sus = pr["sus"].reshape(3, 4, 2)
    want = 1.0 - np.array([[0.0, 0.45, 0.7], [0.0, 0.25, 0.5]]).T
    assert np.array_equal(sus[:, :3], np.broadcast_to(want, (3, 3, 2))) and np.all(sus[:, 3] == 1.0)
    spl = pr["spline"].reshape(3, 4, 10)
    vp = p.vaccination_params
    assert np.array_equal(spl[:, :3, :4], vp.base_equations) and np.array_equal(spl[:, :3, 4:7], vp.knot_locations)
    assert np.array_equal(spl[:, :3, 7:], vp.knot_coefficients) and np.all(spl[:, 3] == 0)
    # group contact = C[a][b] / P_b on every (tier, tier) block
    pop = 100_000 * np.array([0.22, 0.61, 0.17])
    C = np.asarray(p.contact_matrix)
    assert np.allclose(pk.contact.reshape(3, 4, 3, 4), (C / pop[None, :])[:, None, :, None], rtol=1e-15)
    # state: the padded tier slot is empty
    s = pk.y0[:12].reshape(3, 4)
    assert np.all(s[:, 1:] == 0) and np.allclose(s[:, 0], 0.9995 * pop)


def test_pack_rejects_bad_shapes():
    cfg, p, _ = _packed()
    vp = p.vaccination_params
    state = cfg.initializer.get_initial_state(cfg)
    bad = SEIRS_MultiStrain_ODEParams(**{**p.__dict__, "vaccination_params": VaccinationParams(
        vp.knot_locations, vp.base_equations, vp.knot_coefficients, np.full((2, 3), 1.5))})
    with pytest.raises(ValueError, match="vaccine_efficacy"):
        seirs_multi_strain_ode.pack(state, bad)
    bad = SEIRS_MultiStrain_ODEParams(**{**p.__dict__, "vaccination_params": VaccinationParams(
        vp.knot_locations[:, :2], vp.base_equations, vp.knot_coefficients, vp.vaccine_efficacy)})
    with pytest.raises(ValueError, match="splines"):
        seirs_multi_strain_ode.pack(state, bad)
    with pytest.raises(ValueError, match="expected"):
        seirs_multi_strain_ode.pack(tuple(a[:, :2] for a in state), p)


def test_spline_in_the_rhs_is_the_utils_spline():
    """The dose rate the right-hand side uses is utils.evaluate_cubic_spline of the same arrays
    (reference src/dynode/utils/splines.py:66-109)."""
    from dynode_amd import utils

    cfg, p, pk = _packed()
    vp = p.vaccination_params
    state = cfg.initializer.get_initial_state(cfg)
    for t in (0.0, 31.0, 45.5, 70.0, 140.0):
        ds = seirs_multi_strain_ode(t, state, p)[0]
        nu = np.asarray(utils.evaluate_cubic_spline(t, vp.knot_locations, vp.base_equations, vp.knot_coefficients))
        pop = np.array([a.reshape(3, -1).sum(1) for a in state[:4]]).sum(0)
        doses = np.minimum(nu[:, 0] * pop, state[0][:, 0])
        # at t = 0 nobody has a dose yet: tier 1 only gains what tier 0 loses to vaccination
        assert np.allclose(ds[:, 1], doses, rtol=1e-12, atol=1e-12)


def test_host_evaluation_matches_the_numpy_twin_and_the_oracle():
    cfg, p, pk = _packed()
    rng = np.random.default_rng(3)
    m = pk.model
    y = pk.y0 + rng.uniform(0, 50, pk.y0.size) * (np.arange(pk.y0.size) % 4 != 3 if True else 1)
    # rebuild a front-end state from the flat one (padded tier slots dropped)
    parts, pos = [], 0
    for shape in pk.shapes:
        n = int(np.prod(shape))
        parts.append(y[pos:pos + n].reshape(shape)[:, :3]); pos += n
    # the group contact matrix depends on the age populations of THIS state
    pk2 = seirs_multi_strain_ode.pack(tuple(parts), p)
    for t in (0.0, 33.0, 47.0, 90.0):
        got = np.concatenate([np.pad(g, [(0, 0), (0, 1)] + [(0, 0)] * (g.ndim - 2)).ravel()
                              for g in seirs_multi_strain_ode(t, tuple(parts), p)])
        twin = H.rhs_numpy(pk2.model, t, pk2.y0, pk2.params[0], pk2.contact)
        orc = O.rhs(H.omodel(pk2.model), t, pk2.y0, pk2.params[0], pk2.contact)
        scale = np.abs(twin).max()
        assert np.abs(got - twin).max() < 1e-12 * scale and np.abs(orc - twin).max() < 1e-12 * scale


def test_oracle_solution_vs_scipy_and_conservation():
    cfg, p, pk = _packed()
    m, ts = pk.model, np.arange(0.0, 201.0, 10.0)
    want = H.ground_truth(m, pk.y0, pk.params[0], pk.contact, 200.0, ts, rtol=1e-10, atol=1e-6)
    ys, st, na, nr = O.solve(H.omodel(m), pk.y0, pk.params, pk.contact, 200.0, ts, dtype=np.float64, rtol=1e-9, atol=1e-6)
    assert st[0] == 0 and np.abs(ys[0] - want).max() < 2e-3          # of 1e5 people
    G, S = m.n_age, m.n_strain
    people = ys[0][:, :G + 3 * G * S]
    per_group = people[:, :G] + people[:, G:].reshape(len(ts), 3, G, S).sum((1, 3))
    by_age = per_group.reshape(len(ts), 3, 4).sum(-1)
    assert np.abs(by_age - 100_000 * np.array([0.22, 0.61, 0.17])).max() < 1e-6
    tiers = per_group.reshape(len(ts), 3, 4).sum(1)
    assert np.all(tiers[:, 3] == 0)                                   # the padded slot stays empty
    assert tiers[2, 1] == 0 and tiers[5, 1] > 0 and tiers[7, 2] > 0   # first doses from day 30, second from day 58
    assert np.all(np.diff(tiers[:, 2]) > -1e-9 - 0.02 * tiers[:-1, 2])


def test_without_doses_and_without_protection_it_is_the_plain_model():
    """nu = 0 and efficacy = 0: summing the tier axis away gives the un-tiered model's trajectory
    (examples/seirs_multi_strain_age_stratified.py's right-hand side with P-normalised contacts)."""
    cfg = ex.get_config(efficacy=({0: 0.0, 1: 0.0, 2: 0.0}, {0: 0.0, 1: 0.0, 2: 0.0}))
    p = ex.get_odeparams(cfg)
    vp = p.vaccination_params
    p.vaccination_params = VaccinationParams(vp.knot_locations, vp.base_equations, 0 * vp.knot_coefficients, vp.vaccine_efficacy)
    state = cfg.initializer.get_initial_state(cfg)
    pk = seirs_multi_strain_ode.pack(state, p)
    ts = np.arange(0.0, 151.0, 25.0)
    ys, st, _, _ = O.solve(H.omodel(pk.model), pk.y0, pk.params, pk.contact, 150.0, ts, dtype=np.float64, rtol=1e-10, atol=1e-8)
    q = SEIRS_MultiStrain_ODEParams(beta=p.beta, gamma=p.gamma, sigma=p.sigma, omega=p.omega, contact_matrix=p.contact_matrix, idx=p.idx)
    pk0 = seirs_multi_strain_ode.pack(tuple(a.sum(1) for a in state), q)
    ys0, st0, _, _ = O.solve(H.omodel(pk0.model), pk0.y0, pk0.params, pk0.contact, 150.0, ts, dtype=np.float64, rtol=1e-10, atol=1e-8)
    assert st[0] == 0 and st0[0] == 0
    pos = pos0 = 0
    for shape, shape0 in zip(pk.shapes, pk0.shapes):
        n, n0 = int(np.prod(shape)), int(np.prod(shape0))
        tiered = ys[0][:, pos:pos + n].reshape((len(ts),) + shape).sum(2)
        flat = ys0[0][:, pos0:pos0 + n0].reshape((len(ts),) + shape0)
        assert np.abs(tiered - flat).max() < 1e-4
        pos, pos0 = pos + n, pos0 + n0


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_simulate_vaccination_example_matches_oracle_and_protects():
    import torch

    cfg = ex.get_config()
    sol = ex.run_simulation(cfg, tf=300)
    s, e, i, r, c = (a.cpu().numpy() for a in sol.ys)
    assert s.shape == (301, 3, 3) and c.shape == (301, 3, 3, 2) and sol.ys[0].dtype == torch.float32
    _, p, pk = _packed(cfg)
    ts = np.arange(0.0, 301.0)
    want, st, _, _ = O.solve(H.omodel(pk.model), pk.y0, pk.params, pk.contact, 300.0, ts, dtype=np.float32)
    assert st[0] == 0
    pos = 0
    for got, shape in zip((s, e, i, r, c), pk.shapes):
        n = int(np.prod(shape))
        ref = want[0][:, pos:pos + n].reshape((301,) + shape)[:, :, :3]
        # min(doses, susceptibles) has a kink where a tier runs empty: single accept/reject decisions differ
        # between the two fp32 runs there, so they agree to a few solver tolerances (rtol 1e-5), not to rounding
        assert np.abs(got - ref).max() < 2e-4 * 100_000, shape
        pos += n
    by_age = s.sum(2) + (e + i + r).sum((2, 3))
    assert np.abs(by_age - by_age[0]).max() < 1.0                    # fp32, 1e5 people
    assert s[20, :, 1:].max() == 0 and s[60, :, 1].min() > 0 and s[120, :, 2].min() > 0
    # protection: with the same dose schedule but no efficacy more people are infected
    naive = ex.run_simulation(ex.get_config(efficacy=({0: 0, 1: 0, 2: 0}, {0: 0, 1: 0, 2: 0})), tf=300)
    assert naive.ys[4][-1].sum() > 1.02 * sol.ys[4][-1].sum()


@pytest.mark.gpu
def test_fused_observation_likelihood_for_a_vaccinated_model():
    """simulate(..., observe=...) with tiers: the caller's observation array has the tracked tiers only; the padded
    tier slots of the kernel layout are filled and their constant contribution removed on the host.  Value and
    gradient (with respect to the vaccine efficacy) equal scoring the saved trajectory."""
    import torch
    from dynode_amd import PoissonObservation, SolverParams, simulate
    from dynode_amd.rhs import SEIRS_MultiStrain_ODEParams, VaccinationParams

    cfg = ex.get_config()
    p0 = ex.get_odeparams(cfg)
    vp0 = p0.vaccination_params
    y0 = cfg.initializer.get_initial_state(cfg)
    sp = SolverParams()
    rng = np.random.default_rng(0)
    base = ex.run_simulation(cfg, tf=120).ys[cfg.idx.c].cpu().numpy()
    obs = torch.as_tensor(rng.poisson(np.clip(np.diff(base, axis=0), 0, None)).astype(np.float64))     # (120, 3, 3, 2)

    def both(ve):
        q = SEIRS_MultiStrain_ODEParams(beta=p0.beta, gamma=p0.gamma, sigma=p0.sigma, omega=p0.omega, contact_matrix=p0.contact_matrix,
                                        vaccination_params=VaccinationParams(vp0.knot_locations, vp0.base_equations, vp0.knot_coefficients, ve))
        fused = simulate(seirs_multi_strain_ode, 120, y0, q, sp, dtype=torch.float64,
                         observe=PoissonObservation(compartment=cfg.idx.c, data=obs, increments=True, floor=1e-6)).log_likelihood
        sol = simulate(seirs_multi_strain_ode, 120, y0, q, sp, dtype=torch.float64)
        rate = torch.clamp(torch.diff(sol.ys[cfg.idx.c], dim=0), min=1e-6)
        o = obs.to(rate.device)
        plain = (o * torch.log(rate) - rate - torch.lgamma(o + 1.0)).sum()
        return fused.reshape(()), plain

    ve = torch.tensor(np.asarray(vp0.vaccine_efficacy, dtype=float) + np.array([[0.05, 0, 0], [0.05, 0, 0]]), dtype=torch.float64,
                      device="cuda", requires_grad=True)
    fused, plain = both(ve)
    assert abs(float(fused.detach()) - float(plain.detach())) < 1e-7 * abs(float(plain.detach()))
    (g_f,) = torch.autograd.grad(fused, ve)
    (g_p,) = torch.autograd.grad(plain, ve)
    assert torch.allclose(g_f, g_p, rtol=1e-6, atol=1e-8)
    with pytest.raises(ValueError, match="observations have shape"):
        simulate(seirs_multi_strain_ode, 120, y0, p0, sp, observe=PoissonObservation(compartment=cfg.idx.c, data=obs[:, :, :2], increments=True))
