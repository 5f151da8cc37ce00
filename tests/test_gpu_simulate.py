"""The reference's own simulate-path tests, run through dynode_amd.simulate on the GPU.

Each test names the reference test it re-expresses (files under /root/reference/tests/), so they
read like the reference's suite: same models (our examples/ mirror theirs), same assertions, same
tolerances -- plus parity of `simulate` against the oracle and the batched extension.
"""

import numpy as np
import pytest
import torch
from scipy.optimize import root_scalar

import helpers as H
from dynode_amd import Dopri5, SolverError, SolverParams, rhs, simulate
from dynode_amd.simulation import odes
from examples import seirs as ex_seirs
from examples import seirs_multi_strain_age_stratified as ex_ms
from examples import sir as ex_sir
from examples import sir_age_risk_stratified as ex_risk
from examples import sir_age_stratified as ex_age

pytestmark = pytest.mark.gpu
O = H.O


def _np(sol):
    return [a.cpu().numpy() for a in sol.ys]


@pytest.fixture
def test_ode():
    """tests/test_simulation/test_odes.py:11-42: beta*s*i (no /N), y0 = (99, 1, 0), r0 = 2, T_inf = 7."""
    y0 = (np.array([99.0]), np.array([1.0]), np.array([0.0]))
    return rhs.sir_ode_unnormalised, y0, rhs.SIR_ODEParams(beta=2.0 / 7.0, gamma=1 / 7.0)


def test_simulation_expected_shapes(test_ode):
    """tests/test_simulation/test_odes.py:45-60."""
    ode, y0, p = test_ode
    for days in [50, 100, 200, 300.0]:
        sol = simulate(ode, duration_days=days, initial_state=y0, ode_parameters=p, solver_parameters=SolverParams())
        assert all(sol.ys[c].shape == (days + 1, 1) for c in range(3))
        assert sol.ts.shape == (int(days) + 1,) and float(sol.ts[-1]) == days


def test_first_timestep_is_the_initial_state(test_ode):
    """tests/test_simulation/test_odes.py:63-74."""
    ode, y0, p = test_ode
    sol = simulate(ode, 100, y0, p, SolverParams())
    for c in range(3):
        assert float(sol.ys[c][0, 0]) == float(y0[c][0])


@pytest.mark.parametrize("save_step", [1, 2, 3, 7])
def test_save_step_shapes(test_ode, save_step):
    """tests/test_simulation/test_odes.py:77-92."""
    ode, y0, p = test_ode
    sol = simulate(ode, 100, y0, p, SolverParams(), save_step=save_step)
    assert all(sol.ys[c].shape == (int(100 / save_step) + 1, 1) for c in range(3))


@pytest.mark.parametrize("sub", [(0,), (1,), (2,), (0, 1), (0, 2), (1, 2), (0, 1, 2)])
def test_sub_save_indices(test_ode, sub):
    """tests/test_simulation/test_odes.py:95-120: unsaved compartments have shape (101, 0)."""
    ode, y0, p = test_ode
    full = simulate(ode, 100, y0, p, SolverParams())
    sol = simulate(ode, 100, y0, p, SolverParams(), sub_save_indices=sub)
    for c in range(3):
        if c in sub:
            assert torch.equal(sol.ys[c], full.ys[c])
        else:
            assert sol.ys[c].shape == (101, 0)


@pytest.mark.parametrize("s0,i0,r0", [(0.99, 0.01, 0.0), (0.95, 0.05, 0.0), (0.90, 0.10, 0.0), (0.80, 0.20, 0.0)])
def test_final_epidemic_size_matches_theory(s0, i0, r0):
    """tests/test_sir_dynamics/test_sir.py:18-65."""
    cfg = ex_sir.get_config()
    sol = simulate(rhs.sir_ode, 300, cfg.initializer.get_initial_state(s_0=s0, i_0=i0, r_0=r0), ex_sir.get_odeparams(cfg),
                   cfg.parameters.solver_params)
    r0_param = cfg.parameters.transmission_params.strains[0].r0
    s_inf = root_scalar(lambda x: x - s0 * np.exp(-r0_param * (1 - x)), bracket=[0.0, s0], method="bisect", xtol=1e-8).root
    assert float(sol.ys[2].squeeze()[-1]) == pytest.approx(1 - s_inf, abs=2e-2)


@pytest.mark.parametrize("s0,i0,r0", [(0.99, 0.01, 0.0), (0.95, 0.05, 0.0), (0.90, 0.10, 0.0), (0.80, 0.20, 0.0),
                                      (0.8, 0.0, 0.2), (0.75, 0.1, 0.15)])
def test_sir_mass_conservation(s0, i0, r0):
    """tests/test_sir_dynamics/test_sir.py:68-100."""
    cfg = ex_sir.get_config()
    sol = simulate(rhs.sir_ode, 120, cfg.initializer.get_initial_state(s_0=s0, i_0=i0, r_0=r0), ex_sir.get_odeparams(cfg),
                   cfg.parameters.solver_params)
    s, i, r = [a.squeeze() for a in _np(sol)]
    total = s + i + r
    assert np.allclose(total, total[0], atol=1e-6)


@pytest.mark.parametrize("r0,ti,tl,tw", [(2.0, 7.0, 3.0, 60.0), (3.0, 5.0, 2.0, 100.0)])
def test_seirs_endemic_equilibrium(r0, ti, tl, tw):
    """tests/test_seirs_dynamics/test_seirs.py:8-65."""
    cfg = ex_seirs.get_config(r_0=r0, infectious_period=ti, latent_period=tl, waning_period=tw)
    p = ex_seirs.get_seirs_odeparams(cfg)
    sol = simulate(rhs.seirs_ode, 1000, cfg.initializer.get_initial_state(), p, cfg.parameters.solver_params)
    s, e, i, r = [a.squeeze() for a in _np(sol)]
    beta, gamma, sigma, omega = float(p.beta), float(p.gamma), float(p.sigma), float(p.omega)
    s_star = gamma / beta
    i_star = (1 - s_star) / (1 + gamma / sigma + gamma / omega)
    assert s[-1] == pytest.approx(s_star, rel=1e-2) and i[-1] == pytest.approx(i_star, rel=1e-2)
    assert e[-1] == pytest.approx(gamma * i_star / sigma, rel=1e-2) and r[-1] == pytest.approx(gamma * i_star / omega, rel=1e-2)
    assert all(x[-100:].std() < 1e-4 for x in (s, e, i, r))


@pytest.mark.parametrize("r_0, infectious_period, latent_period, waning_period",
                         [(2.0, 7.0, 3.0, 60.0), (3.0, 5.0, 2.0, 100.0)])
def test_seasonal_seirs_keeps_oscillating(r_0, infectious_period, latent_period, waning_period):
    """tests/test_seirs_seasonality_dynamics/test_seirs_seasonality_dynamics.py:19-42, through the same
    module and names the reference's test imports (examples.seirs_seasonal_forcing)."""
    from examples.seirs_seasonal_forcing import get_config, get_seirs_odeparams, seirs_ode_seasonal

    config = get_config(r_0=r_0, infectious_period=infectious_period, latent_period=latent_period,
                        waning_period=waning_period)
    sol = simulate(ode=seirs_ode_seasonal, duration_days=1000, initial_state=config.initializer.get_initial_state(),
                   ode_parameters=get_seirs_odeparams(config), solver_parameters=config.parameters.solver_params)
    s, e, i, r = [a.squeeze() for a in _np(sol)]
    assert s[-100:].std() > 1e-4 and e[-100:].std() > 1e-4 and i[-100:].std() > 1e-4 and r[-100:].std() > 1e-4


# ------------------------------------------------------------------ the examples, vs the oracle
def _oracle(ode, y0, p, t1, sp=None, dtype=np.float32, **kw):
    pk = ode.pack(y0, p)
    ts = odes.build_saveat(0.0, t1).ts
    ys, st, na, nr = O.solve(H.omodel(pk.model), pk.y0, pk.params, pk.contact, t1, ts, dtype=dtype, **kw)
    return ys, pk


@pytest.mark.parametrize("name", ["sir", "age", "seirs", "multi", "risk"])
def test_examples_match_oracle(name):
    if name == "sir":
        cfg = ex_sir.get_config(); ode, y0, p = rhs.sir_ode, cfg.initializer.get_initial_state(), ex_sir.get_odeparams(cfg)
    elif name == "age":
        cfg = ex_age.get_config(); ode, y0, p = rhs.sir_ode, cfg.initializer.get_initial_state(), ex_age.get_odeparams(cfg)
    elif name == "seirs":
        cfg = ex_seirs.get_config(); ode, y0, p = rhs.seirs_ode, cfg.initializer.get_initial_state(), ex_seirs.get_seirs_odeparams(cfg)
    elif name == "multi":
        cfg = ex_ms.get_config(); ode, y0, p = rhs.seirs_multi_strain_ode, cfg.initializer.get_initial_state(cfg), ex_ms.get_odeparams(cfg)
    else:
        cfg = ex_risk.get_config(); ode, y0, p = rhs.sir_age_risk_ode, cfg.initializer.get_initial_state(), ex_risk.get_odeparams(cfg)
    sol = simulate(ode, 150, y0, p, cfg.parameters.solver_params)
    want, pk = _oracle(ode, y0, p, 150)
    got = np.concatenate([a.reshape(151, -1) for a in _np(sol)], axis=1)
    scale = np.abs(want).max()
    assert np.abs(got - want[0]).max() / scale < 1e-5
    assert [tuple(a.shape[1:]) for a in sol.ys] == [tuple(s) for s in pk.shapes]
    assert int(sol.result) == 0 and int(sol.stats["num_steps"]) == int(sol.stats["num_accepted_steps"]) + int(sol.stats["num_rejected_steps"])


def test_multi_strain_example_uses_idx_like_the_reference():
    cfg = ex_ms.get_config()
    sol = simulate(rhs.seirs_multi_strain_ode, 200, cfg.initializer.get_initial_state(cfg), ex_ms.get_odeparams(cfg),
                   cfg.parameters.solver_params)
    c = sol.ys[cfg.idx.c]
    assert c.shape == (201, 2, 3)
    by_strain = c.sum(dim=cfg.idx.c.age + 1)          # callers add 1 for the time axis (sir_infer_parameters.py:69-79)
    assert by_strain.shape == (201, 3) and bool((by_strain[1:] >= by_strain[:-1] - 1e-3).all())


def test_x64_and_dopri5_options():
    cfg = ex_age.get_config()
    y0, p = cfg.initializer.get_initial_state(), ex_age.get_odeparams(cfg)
    sol64 = simulate(rhs.sir_ode, 100, y0, p, cfg.parameters.solver_params, dtype=torch.float64)
    want, _ = _oracle(rhs.sir_ode, y0, p, 100, dtype=np.float64)
    got = np.concatenate([a.reshape(101, -1) for a in _np(sol64)], axis=1)
    assert sol64.ys[0].dtype == torch.float64 and np.abs(got - want[0]).max() < 1e-9
    odes.enable_x64(True)
    try:
        assert simulate(rhs.sir_ode, 10, y0, p, cfg.parameters.solver_params).ys[0].dtype == torch.float64
    finally:
        odes.enable_x64(False)
    sp = SolverParams(solver_method=Dopri5())
    sol = simulate(rhs.sir_ode, 100, (np.array([0.9]), np.array([0.1]), np.array([0.0])),
                   rhs.SIR_ODEParams(beta=2 / 7, gamma=1 / 7), sp)
    want, _ = _oracle(rhs.sir_ode, (np.array([0.9]), np.array([0.1]), np.array([0.0])),
                      rhs.SIR_ODEParams(beta=2 / 7, gamma=1 / 7), 100, method="dopri5")
    assert np.abs(np.concatenate([a.reshape(101, -1) for a in _np(sol)], 1) - want[0]).max() < 1e-5


def test_max_steps_raises_like_the_reference():
    """params.py:51-55: 'maximum number of steps ... before raising an error'."""
    cfg = ex_sir.get_config()
    sp = SolverParams(max_steps=5)
    with pytest.raises(SolverError, match="max_steps"):
        simulate(rhs.sir_ode, 300, cfg.initializer.get_initial_state(), ex_sir.get_odeparams(cfg), sp)
    sol = simulate(rhs.sir_ode, 300, cfg.initializer.get_initial_state(), ex_sir.get_odeparams(cfg), sp, throw=False)
    assert int(sol.result) == 1 and bool(torch.isinf(sol.ys[0][-1]).all())


def test_batched_simulate_equals_a_loop_of_single_calls():
    cfg = ex_age.get_config()
    y0 = cfg.initializer.get_initial_state()
    rng = np.random.default_rng(0)
    r0, ti = rng.uniform(1.5, 2.5, 9), rng.uniform(4, 10, 9)
    C = cfg.parameters.transmission_params.contact_matrix
    batched = simulate(rhs.sir_ode, 100, y0, rhs.SIR_ODEParams(beta=r0 / ti, gamma=1 / ti, contact_matrix=C),
                       cfg.parameters.solver_params)
    assert batched.ys[0].shape == (9, 101, 2) and batched.result.shape == (9,)
    for b in range(9):
        one = simulate(rhs.sir_ode, 100, y0, rhs.SIR_ODEParams(beta=np.array(r0[b] / ti[b]), gamma=np.array(1 / ti[b]),
                                                                contact_matrix=C), cfg.parameters.solver_params)
        for c in range(3):
            assert torch.equal(one.ys[c], batched.ys[c][b])
