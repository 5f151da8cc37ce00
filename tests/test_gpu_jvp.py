"""Forward-mode tangents of the solve (dyn_solve_batch_jvp), the gradient-solve under NUTS.

Checker: finite differences of the fp64 ORACLE primal.  With ConstantStepSize the discrete solve
is a smooth map of the parameters, so central differences match the kernel's tangents to FD
accuracy -- this pins "tangent == exact derivative of the computed trajectory".  With the adaptive
controller at tight tolerances both converge to the true sensitivity.
"""

import numpy as np
import pytest
import torch

import helpers as H
from dynode_amd import ModelDesc, synthetic
from dynode_amd.engine import solve_batch

pytestmark = pytest.mark.gpu
O = H.O


def fd_oracle(m, y0, p, C, t1, ts, dp, dy0=None, eps=1e-6, **kw):
    """Central difference of the oracle along direction (dp, dy0), fp64."""
    dy0 = np.zeros_like(y0) if dy0 is None else dy0
    up, _, _, _ = O.solve(H.omodel(m), y0 + eps * dy0, p + eps * dp, C, t1, ts, dtype=np.float64, **kw)
    dn, _, _, _ = O.solve(H.omodel(m), y0 - eps * dy0, p - eps * dp, C, t1, ts, dtype=np.float64, **kw)
    return (up - dn) / (2 * eps)


CASES = [
    (ModelDesc(n_age=2), 2), (ModelDesc(n_age=1), 2), (ModelDesc(n_age=8), 2),
    (ModelDesc(n_age=1, has_e=True, has_wane=True), 4), (ModelDesc(n_age=1, has_e=True, has_wane=True), 1),
    (ModelDesc(n_age=2, n_strain=3, has_e=True, has_wane=True, has_c=True), 1),
    (ModelDesc(n_age=1, has_e=True, has_wane=True, seasonal=True), 4),
    # externally introduced strains: tangents with respect to introduction time / scale / percentage too
    (ModelDesc(n_age=3, n_strain=2, has_e=True, has_wane=True, has_c=True, has_intro=True, intro_age_mask=(0b001, 0b110)), 2),
    (ModelDesc(n_age=1, has_e=True, has_wane=True, has_intro=True, intro_age_mask=(1,)), 2),
    (ModelDesc(n_age=8, has_intro=True, intro_age_mask=(0b00111100,)), 2),
    (ModelDesc(n_age=8, normalize=False, has_intro=True, intro_age_mask=(0b00111100,)), 2),
]


def _workload(m, B, seed):
    from test_gpu_parity import intro_workload, random_workload
    if m.has_intro:
        y0, p, C, t1, ts = intro_workload(m, B, seed, t1=80.0)
        S, at = m.n_strain, m.n_strain * (2 + int(m.has_e) + int(m.has_wane))
        p[:, at:at + S] = np.random.default_rng(seed).uniform(15.0, 60.0, (B, S))     # arrivals inside the 80 days
        if not m.normalize:
            p[:, :S] /= 1000.0                                                         # mass-action beta for N = 1000
        return y0, p, C, t1, ts
    return random_workload(m, B, seed, t1=80.0)


@pytest.mark.parametrize("m,nd", CASES, ids=lambda v: str(v) if isinstance(v, int) else f"A{v.n_age}S{v.n_strain}e{int(v.has_e)}s{int(v.seasonal)}i{int(v.has_intro)}n{int(v.normalize)}")
def test_tangents_equal_derivative_of_the_discrete_solve(m, nd):
    B = 11
    y0, p, C, t1, ts = _workload(m, B, seed=7 + nd)
    rng = np.random.default_rng(0)
    dp = rng.normal(size=(B, nd, m.param_dim)) * 0.1 * np.abs(p)[:, None, :]
    dy0 = rng.normal(size=(B, nd, m.state_dim))
    r = solve_batch(m, y0, p, C, t1, ts, dtype=torch.float64, constant_dt=0.25, dparams=dp, dy0=dy0)
    torch.cuda.synchronize()
    want_y, _, _, _ = O.solve(H.omodel(m), y0, p, C, t1, ts, dtype=np.float64, constant_dt=0.25)
    assert np.abs(r.ys.cpu().numpy() - want_y).max() / 1000 < 1e-11           # primal untouched by the tangents
    dys = r.dys.cpu().numpy()
    assert dys.shape == (B, len(ts), nd, m.state_dim)
    for j in range(nd):
        # (seeding the absent strain's initial infections is an exponentially amplified direction: the
        # difference quotient's eps^2 truncation error needs a smaller step there)
        want = fd_oracle(m, y0, p, C, t1, ts, dp[:, j], dy0[:, j], eps=1e-7 if m.has_intro else 1e-6, constant_dt=0.25)
        scale = np.abs(want).max() + 1e-12
        assert np.abs(dys[:, :, j] - want).max() / scale < 2e-6, (j, np.abs(dys[:, :, j] - want).max() / scale)


def test_adaptive_tangents_converge_to_the_true_sensitivity():
    m = ModelDesc(n_age=2)
    wl = synthetic.sir_two_age_literal(t1=100.0)
    p = np.repeat(wl.params, 3, axis=0) * np.array([[1.0], [1.1], [0.9]])
    dp = np.zeros((3, 2, 2)); dp[:, 0, 0] = 1.0; dp[:, 1, 1] = 1.0             # d/dbeta, d/dgamma
    r = solve_batch(m, wl.y0, p, wl.contact, 100.0, wl.save_ts, dtype=torch.float64, rtol=1e-10, atol=1e-10, dparams=dp)
    torch.cuda.synchronize()
    for j in range(2):
        want = fd_oracle(m, wl.y0, p, wl.contact, 100.0, wl.save_ts, dp[:, j], eps=1e-5, rtol=1e-12, atol=1e-12)
        scale = np.abs(want).max()
        assert np.abs(r.dys.cpu().numpy()[:, :, j] - want).max() / scale < 1e-6
    # default tolerances, fp32: tangents stay within solver-tolerance distance of the truth
    r32 = solve_batch(m, wl.y0, p, wl.contact, 100.0, wl.save_ts, dtype=torch.float32, dparams=dp)
    torch.cuda.synchronize()
    ref = r.dys.cpu().numpy()
    assert np.abs(r32.dys.cpu().numpy() - ref).max() / np.abs(ref).max() < 2e-3
    plain = solve_batch(m, wl.y0, p, wl.contact, 100.0, wl.save_ts, dtype=torch.float32).ys
    assert float((r32.ys - plain).abs().max()) / 1000 < 1e-5     # same algorithm, separately compiled kernels


def test_jvp_sub_save_and_failures(monkeypatch):
    m = ModelDesc(n_age=2)
    wl = synthetic.sir_two_age_literal(t1=50.0)
    dp = np.ones((1, 2, 2))
    full = solve_batch(m, wl.y0, wl.params, wl.contact, 50.0, wl.save_ts, dtype=torch.float64, dparams=dp)
    sub = solve_batch(m, wl.y0, wl.params, wl.contact, 50.0, wl.save_ts, dtype=torch.float64, dparams=dp,
                      save_mask=(0, 0, 1))
    assert sub.dys.shape == (1, 51, 2, 2) and torch.equal(sub.dys, full.dys[..., 4:])
    bad = solve_batch(m, wl.y0, wl.params, wl.contact, 50.0, wl.save_ts, dtype=torch.float64, dparams=dp, max_steps=3)
    torch.cuda.synchronize()
    assert int(bad.status[0]) == 1 and bool(torch.isinf(bad.dys[0, -1]).all())
    from dynode_amd.engine import SolveError
    monkeypatch.setenv("DYNODE_HIP_JIT", "0")        # (with on-demand builds enabled the 3-direction kernel would be compiled)
    with pytest.raises(SolveError, match="UNSUPPORTED"):
        solve_batch(ModelDesc(n_age=8, n_strain=4, has_e=True, has_wane=True, has_c=True), np.zeros(136),
                    np.ones((1, 16)), np.eye(8), 10.0, [0.0, 10.0], dparams=np.ones((1, 3, 16)))   # no 3-direction kernel


def test_gradient_of_a_multi_strain_model_through_simulate():
    """Directions are tiled over the compiled tangent kernels: 12 parameters of the 2-age x 3-strain
    example (1 direction per launch) and 16 of the 8 x 4 model (2 per launch, strains split over lanes)."""
    from dynode_amd import rhs, simulate
    from examples import seirs_multi_strain_age_stratified as ex

    for cfg in (ex.get_config(),
                ex.get_config(r0s=(2.0, 2.5, 1.8, 2.2), infectious_periods=(7.0, 6.0, 8.0, 7.5),
                              latent_periods=(3.0, 2.5, 4.0, 3.0), waning_periods=(60.0, 80.0, 50.0, 70.0),
                              contact_matrix=np.eye(8) * 0.5 + 0.5 / 8, age_names=tuple(f"a{k}" for k in range(8)),
                              age_demographics=tuple([0.125] * 8))):
        p = ex.get_odeparams(cfg)
        y0 = cfg.initializer.get_initial_state(cfg)
        beta = torch.tensor(p.beta, dtype=torch.float64, device="cuda", requires_grad=True)

        def loss(b):
            q = rhs.SEIRS_MultiStrain_ODEParams(beta=b, gamma=p.gamma, sigma=p.sigma, omega=p.omega,
                                                contact_matrix=p.contact_matrix)
            sol = simulate(rhs.seirs_multi_strain_ode, 60, y0, q, cfg.parameters.solver_params, dtype=torch.float64)
            return sol.ys[cfg.idx.c][-1].sum()          # cumulative incidence at day 60

        val = loss(beta)
        (grad,) = torch.autograd.grad(val, beta)
        eps = 1e-6
        fd = torch.stack([(loss(beta.detach() + eps * torch.eye(len(beta), dtype=torch.float64, device="cuda")[j])
                           - loss(beta.detach() - eps * torch.eye(len(beta), dtype=torch.float64, device="cuda")[j])) / (2 * eps)
                          for j in range(len(beta))])
        assert torch.allclose(grad, fd.detach(), rtol=5e-4), (grad, fd)


# ------------------------------------------------------------------ fused observation likelihood
LL_CASES = [
    # model, directions, observed compartment (index into the state tuple), increments?
    (ModelDesc(n_age=2), 2, 2, True), (ModelDesc(n_age=2), 2, 1, False), (ModelDesc(n_age=1), 2, 0, False),
    (ModelDesc(n_age=8), 2, 2, True), (ModelDesc(n_age=1, has_e=True, has_wane=True), 4, 1, False),
    (ModelDesc(n_age=2, n_strain=3, has_e=True, has_wane=True, has_c=True), 1, 4, True),
    (ModelDesc(n_age=2, n_strain=3, has_e=True, has_wane=True, has_c=True), 1, 2, False),
    (ModelDesc(n_age=8, n_strain=4, has_e=True, has_wane=True, has_c=True), 2, 4, True),
]


@pytest.mark.parametrize("replicas", ["default", "0"])
@pytest.mark.parametrize("m,nd,comp,inc", LL_CASES, ids=lambda v: str(v) if not isinstance(v, ModelDesc) else f"A{v.n_age}S{v.n_strain}e{int(v.has_e)}")
def test_fused_poisson_likelihood_equals_scoring_the_saved_trajectory(m, nd, comp, inc, replicas, hints):
    """dyn_solve_batch_loglik == Poisson log-likelihood (and its tangents) computed from the output of
    dyn_solve_batch_jvp, in float64, for every compartment kind, both observation modes, and both
    kernel paths (in-order accumulation / LDS table of replicated trajectories)."""
    from dynode_amd.engine import solve_batch_loglik

    if replicas != "default":
        hints(replicas_log2=int(replicas))
    B = 37
    y0, p, C, t1, ts = _workload(m, B, seed=11)
    rng = np.random.default_rng(1)
    dp = rng.normal(size=(B, nd, m.param_dim)) * 0.1 * np.abs(p)[:, None, :]
    dy0 = rng.normal(size=(B, nd, m.state_dim))
    ref = solve_batch(m, y0, p, C, t1, ts, dtype=torch.float64, dparams=dp, dy0=dy0)
    off = np.concatenate([[0], np.cumsum(m.compartment_sizes)])
    v = ref.ys[:, :, off[comp]:off[comp + 1]]                      # [B, n_save, size]
    dv = ref.dys[:, :, :, off[comp]:off[comp + 1]]                 # [B, n_save, nd, size]
    if inc:
        v, dv = v[:, 1:] - v[:, :-1], dv[:, 1:] - dv[:, :-1]
    floor = max(float(torch.quantile(v.flatten(), 0.1)), 0.05)     # a tenth of the rates sit on the floor
    obs = torch.as_tensor(rng.poisson(np.clip(v[0].cpu().numpy(), 0.0, 50.0) + 0.5).astype(np.float64), device="cuda")
    rate = torch.clamp(v, min=floor)
    want = (obs * torch.log(rate) - rate).sum((1, 2))
    coef = torch.where(v >= floor, obs / rate - 1.0, torch.zeros_like(rate))
    dwant = (coef[:, :, None, :] * dv).sum((1, 3))
    assert int((v < floor).sum()) > 0
    lp, dlp, st, na, nr = solve_batch_loglik(m, y0, p, C, t1, ts, obs, comp, dparams=dp, dy0=dy0, increments=inc, floor=floor,
                                             dtype=torch.float64)
    torch.cuda.synchronize()
    assert int(st.max()) == 0 and torch.equal(na, ref.n_accept) and torch.equal(nr, ref.n_reject)
    assert torch.allclose(lp, want, rtol=1e-12, atol=1e-9), float((lp - want).abs().max())
    assert torch.allclose(dlp, dwant, rtol=1e-10, atol=1e-8), float((dlp - dwant).abs().max())
    # fp32 kernels: same quantity at fp32 accuracy
    lp32, dlp32, st32, _, _ = solve_batch_loglik(m, y0, p, C, t1, ts, obs, comp, dparams=dp, dy0=dy0, increments=inc, floor=floor,
                                                 dtype=torch.float32)
    assert int(st32.max()) == 0
    assert torch.allclose(lp32, want, rtol=2e-4, atol=2e-2), float((lp32 - want).abs().max())


def test_fused_likelihood_argument_checks_and_failed_solves():
    from dynode_amd.engine import SolveError, solve_batch_loglik

    m = ModelDesc(n_age=2)
    y0, p, C, t1, ts = _workload(m, 5, seed=3)
    dp = np.zeros((5, 2, m.param_dim))
    obs = np.ones((len(ts) - 1, 2))
    with pytest.raises(ValueError):
        solve_batch_loglik(m, y0, p, C, t1, ts, obs[:-1], 2, dparams=dp)            # one row short
    with pytest.raises(ValueError):
        solve_batch_loglik(m, y0, p, C, t1, ts, obs, 3, dparams=dp)                 # no such compartment
    with pytest.raises(SolveError):
        solve_batch_loglik(m, y0, p, C, t1, ts, obs, 2, dparams=dp, floor=0.0)      # the floor must be positive
    lp, dlp, st, _, _ = solve_batch_loglik(m, y0, p, C, t1, ts, obs, 2, dparams=dp, max_steps=3)
    assert bool((st == 1).all()) and bool(torch.isinf(lp).all()) and bool((lp < 0).all()) and bool((dlp == 0).all())


def test_gradient_with_respect_to_the_introduction_time_through_simulate():
    """When did the newcomer arrive?  d(cumulative newcomer infections at day 150) / d(introduction_time)
    and d/d(introduction_percentage) by autograd through `simulate` (Strain fields as tensors) against
    central differences of the same front end."""
    from dynode_amd import simulate
    from dynode_amd.rhs import seirs_multi_strain_ode
    from dynode_amd.simulation import odes
    from examples import seirs_introduced_strain as ex

    def outcome(time, pct):
        cfg = ex.get_config(introduction_time=time, introduction_percentage=pct)
        cfg.parameters.solver_params.ode_solver_rel_tolerance = 1e-9
        cfg.parameters.solver_params.ode_solver_abs_tolerance = 1e-9
        sol = simulate(ode=seirs_multi_strain_ode, duration_days=150, initial_state=cfg.initializer.get_initial_state(cfg),
                       ode_parameters=ex.get_odeparams(cfg), solver_parameters=cfg.parameters.solver_params)
        return sol.ys[cfg.idx.c][..., -1, :, 1].sum(-1)

    odes.enable_x64(True)
    try:
        time = torch.tensor(60.0, dtype=torch.float64, device="cuda", requires_grad=True)
        pct = torch.tensor(0.005, dtype=torch.float64, device="cuda", requires_grad=True)
        out = outcome(time, pct)
        g_time, g_pct = torch.autograd.grad(out.sum(), (time, pct))
        fd_time = (outcome(60.0 + 1e-3, 0.005) - outcome(60.0 - 1e-3, 0.005)) / 2e-3
        fd_pct = (outcome(60.0, 0.005 * (1 + 1e-4)) - outcome(60.0, 0.005 * (1 - 1e-4))) / (2e-4 * 0.005)
    finally:
        odes.enable_x64(False)
    assert float(g_time) < 0 < float(g_pct)                      # arriving later means fewer cases by day 150
    assert abs(float(g_time) / float(fd_time) - 1) < 1e-4 and abs(float(g_pct) / float(fd_pct) - 1) < 1e-4


def test_tangent_kernels_agree_across_direction_counts_on_random_calls():
    """Randomized consistency sweep (the generator of tests/test_gpu_parity.py: irregular save grids,
    discontinuity points, sub-save masks, both methods): in float64 the primal of a tangent solve equals
    the plain solve, and a direction gives the same tangent whether it runs alone (1-direction kernel)
    or as either slot of the 2-direction kernel."""
    import ctypes

    from dynode_amd import _abi
    from test_gpu_parity import fuzz_case

    def has(m, method, nd):
        o = _abi.SolverOptsC(method={"tsit5": 0, "dopri5": 1}[method], dtype=1, rtol=1e-5, atol=1e-6, max_steps=10**6,
                             constant_dt=0.0, jump_ts=None, n_jump=0)
        return bool(_abi.lib().dyn_is_supported_jvp(ctypes.byref(m.c()), ctypes.byref(o), nd))

    ran = 0
    for seed in range(400):
        case = fuzz_case(3000 + seed)
        if case is None:
            continue
        m, y0, p, C, t1, ts, kw = case
        if not (has(m, kw["method"], 1) and has(m, kw["method"], 2)):
            continue
        rng = np.random.default_rng(seed)
        B = p.shape[0]
        dp = rng.normal(size=(B, 2, m.param_dim)) * 0.1 * np.abs(p)[:, None, :]
        dy0 = rng.normal(size=(B, 2, m.state_dim)) * (np.abs(np.broadcast_to(y0, (B, m.state_dim)))[:, None, :] > 0)
        plain = solve_batch(m, y0, p, C, t1, ts, dtype=torch.float64, **kw)
        both = solve_batch(m, y0, p, C, t1, ts, dtype=torch.float64, dparams=dp, dy0=dy0, **kw)
        one = [solve_batch(m, y0, p, C, t1, ts, dtype=torch.float64, dparams=dp[:, j:j + 1], dy0=dy0[:, j:j + 1], **kw) for j in (0, 1)]
        fin = torch.isfinite(plain.ys)
        scale = float(plain.ys[fin].abs().max()) if bool(fin.any()) else 1.0
        assert torch.equal(both.status, plain.status) and torch.equal(both.n_accept, plain.n_accept)
        assert float((both.ys[fin] - plain.ys[fin]).abs().max()) <= 1e-12 * scale if bool(fin.any()) else True
        for j in (0, 1):
            a, b = both.dys[:, :, j], one[j].dys[:, :, 0]
            ok = torch.isfinite(a) & torch.isfinite(b)
            assert torch.equal(torch.isfinite(a), torch.isfinite(b))
            if bool(ok.any()):
                tscale = float(b[ok].abs().max()) + 1e-30
                assert float((a[ok] - b[ok]).abs().max()) <= 1e-10 * tscale, (seed, m, kw, j)
        ran += 1
    assert ran >= 40


def test_fused_likelihood_on_random_calls():
    """The fused Poisson likelihood against scoring the saved tangent solve, on the randomized calls of
    the parity sweep (irregular save grids, discontinuity points, constant steps, both methods)."""
    import ctypes

    from dynode_amd import _abi
    from dynode_amd.engine import solve_batch_loglik
    from test_gpu_parity import fuzz_case

    ran = 0
    for seed in range(500):
        case = fuzz_case(9000 + seed)
        if case is None:
            continue
        m, y0, p, C, t1, ts, kw = case
        kw = {k: v for k, v in kw.items() if k != "save_mask"}
        o = _abi.SolverOptsC(method={"tsit5": 0, "dopri5": 1}[kw["method"]], dtype=1, rtol=1e-5, atol=1e-6, max_steps=10**6,
                             constant_dt=0.0, jump_ts=None, n_jump=0)
        if len(ts) < 3 or not _abi.lib().dyn_is_supported_jvp(ctypes.byref(m.c()), ctypes.byref(o), 1):
            continue
        rng = np.random.default_rng(seed)
        B = p.shape[0]
        dp = rng.normal(size=(B, 1, m.param_dim)) * 0.1 * np.abs(p)[:, None, :]
        comp = int(rng.integers(len(m.compartment_names)))
        inc = bool(rng.integers(2))
        ref = solve_batch(m, y0, p, C, t1, ts, dtype=torch.float64, dparams=dp, **kw)
        off = np.concatenate([[0], np.cumsum(m.compartment_sizes)])
        v, dv = ref.ys[:, :, off[comp]:off[comp + 1]], ref.dys[:, :, 0, off[comp]:off[comp + 1]]
        if inc:
            v, dv = v[:, 1:] - v[:, :-1], dv[:, 1:] - dv[:, :-1]
        floor = max(float(torch.quantile(v.flatten()[:200000], 0.1)), 1e-3)
        obs = torch.as_tensor(rng.uniform(0.0, 30.0, tuple(v.shape[1:])), device="cuda")
        rate = torch.clamp(v, min=floor)
        want = (obs * torch.log(rate) - rate).sum((1, 2))
        dwant = (torch.where(v >= floor, obs / rate - 1.0, torch.zeros_like(rate)) * dv).sum((1, 2))
        lp, dlp, st, _, _ = solve_batch_loglik(m, y0, p, C, t1, ts, obs, comp, dparams=dp, increments=inc, floor=floor,
                                               dtype=torch.float64, **kw)
        assert torch.equal(st, ref.status) and int(st.max()) == 0
        assert torch.allclose(lp, want, rtol=1e-11, atol=1e-8), (seed, m, kw, comp, inc, float((lp - want).abs().max()))
        assert torch.allclose(dlp[:, 0], dwant, rtol=1e-9, atol=1e-7 * (1.0 + float(dwant.abs().max()))), (seed, m, kw, comp, inc)
        ran += 1
    assert ran >= 40


VAX_CASES = [
    (2, ModelDesc(n_age=4, normalize=False, n_vax_tiers=2, n_vax_knots=2), 2),
    (2, ModelDesc(n_age=8, normalize=False, n_vax_tiers=3, n_vax_knots=1), 1),
    (4, ModelDesc(n_age=8, n_strain=2, has_e=True, has_wane=True, has_c=True, normalize=False, n_vax_tiers=2, n_vax_knots=3), 2),
]


@pytest.mark.parametrize("ages,m,nd", VAX_CASES, ids=lambda v: str(v) if isinstance(v, int) else f"G{v.n_age}S{v.n_strain}K{v.n_vax_tiers}")
@pytest.mark.parametrize("dose_scale", [1.0, 6.0], ids=["doses_last", "tier_runs_empty"])
def test_vaccination_tangents_equal_derivative_of_the_discrete_solve(ages, m, nd, dose_scale):
    """Seeds on every column of a vaccinated model's parameter row -- rates, susceptibilities (1 - efficacy), spline
    base / knots / coefficients -- and on the initial state, against central differences of the fp64 oracle.  With
    dose_scale = 6 the unvaccinated tier runs empty during the run, so both branches of min(doses, s) are taken."""
    from test_gpu_parity import vax_workload

    B = 9
    y0, p, C, t1, ts, pop = vax_workload(ages, m, B, seed=3 + nd, t1=90.0)
    G, S, nk = m.n_age, m.n_strain, m.n_vax_knots
    at = m.param_dim - G * (4 + 2 * nk)
    spl = p[:, at:].reshape(B, G, 4 + 2 * nk)
    spl[:, :, :2] *= dose_scale
    rng = np.random.default_rng(1)
    dp = rng.normal(size=(B, nd, m.param_dim)) * 0.05 * np.abs(p)[:, None, :]
    dy0 = rng.normal(size=(B, nd, m.state_dim)) * (y0 > 0)[:, None, :]         # keep empty tiers empty (no negative people)
    r = solve_batch(m, y0, p, C, t1, ts, dtype=torch.float64, constant_dt=0.25, dparams=dp, dy0=dy0)
    want_y, st, _, _ = O.solve(H.omodel(m), y0, p, C, t1, ts, dtype=np.float64, constant_dt=0.25)
    assert int(r.status.max()) == 0 and np.abs(r.ys.cpu().numpy() - want_y).max() / 1000 < 1e-11
    if dose_scale > 1:
        KV = m.vax_lanes
        assert (want_y[:, -1, :G].reshape(B, -1, KV)[:, :, 0] < 1e-2 * pop).any()     # tier 0 did run empty somewhere
    dys = r.dys.cpu().numpy()
    for j in range(nd):
        want = fd_oracle(m, y0, p, C, t1, ts, dp[:, j], dy0[:, j], eps=1e-6, constant_dt=0.25)
        scale = np.abs(want).max() + 1e-12
        err = np.abs(dys[:, :, j] - want).max(axis=(1, 2)) / scale
        # a trajectory whose difference quotient straddles the instant a tier runs empty sees the kink of min():
        # its quotient is off by O(1) of the jump in slope -- allow one such trajectory per direction
        assert np.sort(err)[-2] < 1e-5 and err.max() < 5e-2, err


def test_gradient_with_respect_to_vaccine_efficacy_through_simulate():
    """autograd through simulate for a vaccinated model (examples/seirs_vaccination.py): d(infections by day 150) /
    d(vaccine efficacy, beta) against central differences of the same float64 solve."""
    from dynode_amd import SolverParams, simulate
    from dynode_amd.rhs import SEIRS_MultiStrain_ODEParams, VaccinationParams, seirs_multi_strain_ode
    from examples import seirs_vaccination as ex

    cfg = ex.get_config()
    p0 = ex.get_odeparams(cfg)
    y0 = cfg.initializer.get_initial_state(cfg)
    vp0 = p0.vaccination_params
    sp = SolverParams(constant_step_size=0.25)
    dev = "cuda"

    def loss(ve, beta):
        vp = VaccinationParams(vp0.knot_locations, vp0.base_equations, vp0.knot_coefficients, ve)
        q = SEIRS_MultiStrain_ODEParams(beta=beta, gamma=p0.gamma, sigma=p0.sigma, omega=p0.omega,
                                        contact_matrix=p0.contact_matrix, vaccination_params=vp)
        sol = simulate(seirs_multi_strain_ode, 150, y0, q, sp, dtype=torch.float64)
        assert sol.ys[cfg.idx.c].shape == (151, 3, 3, 2)
        return sol.ys[cfg.idx.c][-1].sum()

    ve_np = np.asarray(vp0.vaccine_efficacy, dtype=float).copy()
    ve_np[:, 0] = 0.05                                      # off the boundary of [0, 1]: the difference quotient stays valid
    ve = torch.tensor(ve_np, dtype=torch.float64, device=dev, requires_grad=True)
    beta = torch.tensor(np.asarray(p0.beta), dtype=torch.float64, device=dev, requires_grad=True)
    val = loss(ve, beta)
    g_ve, g_beta = torch.autograd.grad(val, (ve, beta))
    assert g_ve.shape == (2, 3) and float(g_ve[:, 0].abs().max()) > 0 and bool((g_ve[:, 1:] < 0).all())   # protection prevents infections
    eps = 1e-6
    with torch.no_grad():
        for l in range(2):
            for k in range(3):
                d = torch.zeros_like(ve); d[l, k] = eps
                fd = (loss(ve + d, beta) - loss(ve - d, beta)) / (2 * eps)
                assert abs(float(g_ve[l, k]) - float(fd)) < 2e-4 * abs(float(fd)) + 1e-6, (l, k, float(g_ve[l, k]), float(fd))
            d = torch.zeros_like(beta); d[l] = eps
            fd = (loss(ve, beta + d) - loss(ve, beta - d)) / (2 * eps)
            assert abs(float(g_beta[l]) - float(fd)) < 2e-4 * abs(float(fd)), (l, float(g_beta[l]), float(fd))


def test_gradient_with_respect_to_the_initial_state_through_simulate():
    """A compartment of the initial state that is computed from a tensor requiring grad (a sampled initial-infection
    scale is the common case) is differentiated through the kernel's dy0 planes: value and gradient of a loss that
    depends on BOTH a rate and the seeding, checked against central differences of the same solve -- plain solve and
    fused Poisson likelihood, batched (one seeding per chain) and shared."""
    from dynode_amd import PoissonObservation, rhs, simulate
    from dynode_amd.config import SolverParams

    dev = "cuda"
    f64 = torch.float64
    sp = SolverParams(constant_step_size=0.25)             # smooth discrete map: differences are meaningful
    pop = torch.tensor([600.0, 400.0], dtype=f64, device=dev)
    C = np.array([[0.7, 0.3], [0.3, 0.7]])
    obs = torch.tensor(np.random.default_rng(3).uniform(0.5, 6.0, (30, 2)), dtype=f64)

    def run(theta, fused):
        # theta [..., 2]: infected share of the population at t0 and beta; trailing axis = the two unknowns
        share, beta = theta[..., 0], theta[..., 1]
        i0 = share[..., None] * pop
        s0 = pop - i0
        r0 = torch.zeros_like(i0)
        p = rhs.SIR_ODEParams(beta=beta, gamma=torch.full_like(beta, 1.0 / 7.0), contact_matrix=C)
        if fused:
            sol = simulate(rhs.sir_ode, 30, (s0, i0, r0), p, sp, dtype=f64,
                           observe=PoissonObservation(compartment=2, data=obs, increments=True, floor=1e-6))
            return sol.log_likelihood
        sol = simulate(rhs.sir_ode, 30, (s0, i0, r0), p, sp, dtype=f64)
        return sol.ys[2][..., -1, :].sum(-1) + 0.5 * sol.ys[1][..., 10, :].sum(-1)

    for fused in (False, True):
        for theta0 in (torch.tensor([0.01, 0.30], dtype=f64, device=dev),                                  # unbatched
                       torch.tensor([[0.01, 0.30], [0.03, 0.25], [0.002, 0.40]], dtype=f64, device=dev)):   # one row per chain
            theta = theta0.clone().requires_grad_(True)
            val = run(theta, fused)
            (grad,) = torch.autograd.grad(val.sum(), theta)
            fd = torch.zeros_like(theta0)
            for j, eps in ((0, 1e-7), (1, 1e-6)):
                d = torch.zeros_like(theta0)
                d[..., j] = eps
                fd[..., j] = (run(theta0 + d, fused) - run(theta0 - d, fused)) / (2 * eps)
            assert torch.isfinite(grad).all() and float(grad[..., 0].abs().min()) > 0.0       # the seeding gradient is there
            assert torch.allclose(grad, fd, rtol=2e-4, atol=1e-6), (fused, grad, fd)


def test_initial_state_gradient_reaches_the_sampler_coordinates():
    """Inside a sampler potential the rows of [params | y0] depend on the chain's own latent row: the tangent solve is
    seeded along the latent coordinates (one launch) and the potential's gradient includes the seeding site."""
    from dynode_amd import rhs, simulate
    from dynode_amd.config import SolverParams
    from dynode_amd.infer import distributions as dist
    from dynode_amd.infer import handlers
    from dynode_amd.infer.inference import Potential
    from dynode_amd.simulation import odes

    C = np.array([[0.7, 0.3], [0.3, 0.7]])
    data = torch.tensor(np.random.default_rng(4).uniform(0.5, 6.0, (30, 2)), dtype=torch.float64)

    def model(obs_data):
        share = handlers.sample("i0_share", dist.Uniform(0.001, 0.05))
        beta = handlers.sample("beta", dist.Uniform(0.15, 0.6))
        pop = torch.tensor([600.0, 400.0], dtype=torch.float64, device=share.device)
        i0 = share[..., None] * pop
        p = rhs.SIR_ODEParams(beta=beta, gamma=torch.full_like(beta, 1.0 / 7.0), contact_matrix=C)
        sol = simulate(rhs.sir_ode, 30, (pop - i0, i0, torch.zeros_like(i0)), p, SolverParams(constant_step_size=0.25))
        inc = torch.clamp(torch.diff(sol.ys[2], dim=-2), min=1e-6)
        handlers.sample("obs", dist.Poisson(inc), obs=obs_data)

    odes.enable_x64(True)
    try:
        pot = Potential(model, dict(obs_data=data), 0, torch.device("cuda"))
        z = torch.tensor([[0.2, -0.4], [-1.0, 0.7]], dtype=torch.float64, device="cuda")
        u, g = pot.potential_and_grad(z)
        for d in range(2):
            dz = torch.zeros_like(z)
            dz[:, d] = 1e-5
            fd = (pot.potential_and_grad(z + dz)[0] - pot.potential_and_grad(z - dz)[0]) / 2e-5
            assert torch.allclose(g[:, d], fd, rtol=2e-4, atol=1e-5), (d, g[:, d], fd)
    finally:
        odes.enable_x64(False)


# ------------------------------------------------------------------ SEIP family: gradients by replaying the primal's steps
def _seip_case(B=3, dtype=np.float64, **shape):
    shape = shape or dict(A=2, L=2, K1=2, M1=2, n_knots=1)
    wl = synthetic.seip(B=B, seed=17, t1=90.0, **shape)
    return wl, synthetic.save_grid(90.0, 3)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_seip_replay_of_the_own_schedule_reproduces_the_adaptive_solve(dtype):
    """dyn_solve_batch_record writes down the accepted steps; replaying them with the same parameters takes the same
    stages from the same states: bitwise the same trajectory, no rejected step, and a follower of ANOTHER row's schedule
    takes that row's step count."""
    wl, ts = _seip_case(B=5, A=4, L=2, K1=3, M1=4, n_knots=2)
    m = wl.model
    base = solve_batch(m, wl.y0, wl.params, wl.contact, 90.0, ts, dtype=dtype, record_steps=512)
    steps, count = base.schedule
    torch.cuda.synchronize()
    assert int(base.status.max()) == 0 and torch.equal(count, base.n_accept) and int(count.min()) > 10
    t = steps[0, :int(count[0])]
    assert float(t[0, 0]) == 0.0 and float(t[-1, 1]) == 90.0 and bool((t[1:, 0] == t[:-1, 1]).all())   # contiguous cover of [0, 90]
    again = solve_batch(m, wl.y0, wl.params, wl.contact, 90.0, ts, dtype=dtype, replay=(steps, count, None))
    assert torch.equal(again.ys, base.ys) and torch.equal(again.n_accept, base.n_accept) and int(again.n_reject.max()) == 0
    # every row on row 2's schedule: same number of steps everywhere, row 2 itself unchanged
    lead = torch.full((5,), 2, dtype=torch.int64)
    tied = solve_batch(m, wl.y0, wl.params, wl.contact, 90.0, ts, dtype=dtype, replay=(steps, count, lead))
    assert int(tied.status.max()) == 0 and bool((tied.n_accept == count[2]).all()) and torch.equal(tied.ys[2], base.ys[2])
    assert not torch.equal(tied.ys[0], base.ys[0])
    # a schedule that did not fit is reported, and followers of it fail instead of stopping short
    small = solve_batch(m, wl.y0, wl.params, wl.contact, 90.0, ts, dtype=dtype, record_steps=8)
    assert int(small.schedule[1].max()) == -1
    bad = solve_batch(m, wl.y0, wl.params, wl.contact, 90.0, ts, dtype=dtype, replay=small.schedule + (None,))
    assert bool((bad.status == 1).all()) and bool(torch.isinf(bad.ys[:, -1]).all())


def test_seip_tangents_equal_differences_of_the_oracle():
    """solve_batch(dparams=..., dy0=...) on the SEIP family (central differences of replayed solves) against central
    differences of the float64 ORACLE under the same constant step: rates, a susceptibility entry, a spline coefficient,
    the initial state."""
    wl, ts = _seip_case()
    m = wl.model
    B, P, D = wl.B, m.param_dim, m.state_dim
    A, L, H, K1, M1, nk = m.seip_dims
    sus_at = 3 * L + M1 + A
    rng = np.random.default_rng(5)
    dp = np.zeros((B, 4, P))
    dp[:, 0, 0] = 1.0                                         # beta of strain 0
    dp[:, 1, L + 1] = 1.0                                     # gamma of strain 1
    dp[:, 2, sus_at + 5] = 1.0                                # one susceptibility entry
    dp[:, 3, :3 * L] = rng.normal(size=(B, 3 * L))            # a mixed direction over all rates ...
    dy = np.zeros((B, 4, D))
    dy[:, 3] = rng.uniform(0.0, 1.0, (B, D)) * (wl.y0 > 0)    # ... and the occupied part of the initial state
    got = solve_batch(m, wl.y0, wl.params, wl.contact, 90.0, ts, dtype=torch.float64, constant_dt=0.5, dparams=dp, dy0=dy)
    import helpers as Hh

    def orc(p, y0):
        ys, st, _, _ = Hh.O.solve(Hh.omodel(m), y0, p, wl.contact, 90.0, ts, dtype=np.float64, n_threads=8, constant_dt=0.5)
        assert st.max() == 0
        return ys

    base = orc(wl.params, wl.y0)
    assert np.abs(got.ys.cpu().numpy() - base).max() / 1000.0 < 1e-11
    for k in range(4):
        eps = 1e-6
        want = (orc(wl.params + eps * dp[:, k], wl.y0 + eps * dy[:, k]) - orc(wl.params - eps * dp[:, k], wl.y0 - eps * dy[:, k])) / (2 * eps)
        have = got.dys[:, :, k].cpu().numpy()
        scale = np.abs(want).max()
        assert scale > 0.1 and np.abs(have - want).max() / scale < 2e-6, (k, np.abs(have - want).max() / scale)


def test_seip_adaptive_tangents_and_autograd_through_simulate():
    """With the step-size controller ON the tangents are those of the recorded step sequence; at tight tolerances that is
    the sensitivity of the ODE itself (oracle differences at rtol 1e-11)."""
    import helpers as Hh

    wl, ts = _seip_case(B=2)
    m = wl.model
    P = m.param_dim
    dp = np.zeros((2, 1, P))
    dp[:, 0, 1] = 1.0                                          # beta of strain 1
    got = solve_batch(m, wl.y0, wl.params, wl.contact, 90.0, ts, dtype=torch.float64, rtol=1e-9, atol=1e-9, dparams=dp)

    def orc(p):
        ys, st, _, _ = Hh.O.solve(Hh.omodel(m), wl.y0, p, wl.contact, 90.0, ts, dtype=np.float64, n_threads=8, rtol=1e-11, atol=1e-11)
        assert st.max() == 0
        return ys

    want = (orc(wl.params + 1e-6 * dp[:, 0]) - orc(wl.params - 1e-6 * dp[:, 0])) / 2e-6
    have = got.dys[:, :, 0].cpu().numpy()
    assert np.abs(have - want).max() / np.abs(want).max() < 2e-5
    # float32, default tolerances: the same sensitivities to a few parts in a thousand of their scale
    g32 = solve_batch(m, wl.y0, wl.params, wl.contact, 90.0, ts, dtype=torch.float32, dparams=dp).dys[:, :, 0].cpu().numpy()
    assert np.abs(g32 - want).max() / np.abs(want).max() < 5e-3
