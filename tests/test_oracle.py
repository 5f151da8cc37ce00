"""Pin the CPU oracle (the checker) before it is trusted.

No diffrax output exists in this pipeline ("parity unpinned"), so the oracle is
pinned by (1) an independent NumPy RHS twin, (2) the reference's OWN analytic tests,
re-expressed against the oracle, (3) fp64 scipy-DOP853 ground truth in tests/golden/,
(4) closed forms.  Reference test files are cited per test.
"""

import numpy as np
import pytest
from scipy.optimize import root_scalar

import helpers as H
from dynode_amd import ModelDesc, synthetic

O = H.O
SIR = ModelDesc(n_age=1)
SEIRS = ModelDesc(n_age=1, has_e=True, has_wane=True)


def solve(m, y0, p, C, t1, ts=None, **kw):
    ts = synthetic.save_grid(t1) if ts is None else ts
    kw.setdefault("dtype", np.float32)  # the reference's effective dtype
    ys, st, na, nr = O.solve(H.omodel(m), y0, np.atleast_2d(p), C, t1, ts, **kw)
    return ys, st, na, nr


# ------------------------------------------------------------------ RHS vs independent twin
@pytest.mark.parametrize("m", [
    ModelDesc(n_age=1), ModelDesc(n_age=1, normalize=False), ModelDesc(n_age=5),
    ModelDesc(n_age=1, has_e=True, has_wane=True, seasonal=True),
    ModelDesc(n_age=2, n_strain=3, has_e=True, has_wane=True, has_c=True),
    ModelDesc(n_age=8, n_strain=4, has_e=True, has_wane=True, has_c=True, seasonal=True),
    ModelDesc(n_age=8, n_strain=4, has_e=True, has_wane=True, has_c=True, n_wane=8),
    ModelDesc(n_age=3, n_strain=2, has_wane=True, n_wane=2),
    ModelDesc(n_age=3, n_strain=2, has_e=True, has_wane=True, has_c=True, has_intro=True, intro_age_mask=(0b000, 0b110)),
    ModelDesc(n_age=8, n_strain=4, has_e=True, has_wane=True, has_c=True, seasonal=True, has_intro=True,
              intro_age_mask=(0, 0xff, 0x0f, 0x81)),
    ModelDesc(n_age=2, normalize=False, has_intro=True, intro_age_mask=(0b01,)),
])
def test_rhs_matches_numpy_twin(m):
    rng = np.random.default_rng(3)
    for _ in range(5):
        y = rng.uniform(0.1, 50.0, m.state_dim)
        p = rng.uniform(0.05, 0.5, m.param_dim)
        if m.seasonal:
            p[-3:] = [0.3, 1.1, 365.0]
        if m.has_intro:                      # introduction day near the evaluation time, scale in days, a few per cent
            S, at = m.n_strain, m.param_dim - (3 if m.seasonal else 0) - 3 * m.n_strain
            p[at:at + S] = rng.uniform(10.0, 25.0, S)
            p[at + S:at + 2 * S] = rng.uniform(2.0, 8.0, S)
            p[at + 2 * S:at + 3 * S] = rng.uniform(0.0, 0.05, S)
        C = rng.uniform(0.1, 1.0, (m.n_age, m.n_age))  # asymmetric: catches a transposed C
        got = O.rhs(H.omodel(m), 17.5, y, p, C)
        want = H.rhs_numpy(m, 17.5, y, p, C)
        np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-13)


def test_rhs_reduces_to_reference_scalar_sir_and_seirs():
    # examples/sir.py:78-84 and examples/seirs.py:88-95 written out literally
    s, i, r, beta, gamma = 0.7, 0.2, 0.1, 2 / 7, 1 / 7
    N = s + i + r
    want = [-beta * s * i / N, beta * s * i / N - gamma * i, gamma * i]
    np.testing.assert_allclose(O.rhs(H.omodel(SIR), 0.0, [s, i, r], [beta, gamma], [[1.0]]), want, rtol=1e-15)
    s, e, i, r, sigma, omega = 0.6, 0.1, 0.2, 0.1, 1 / 3, 1 / 60
    N = s + e + i + r
    want = [-beta * s * i / N + omega * r, beta * s * i / N - sigma * e, sigma * e - gamma * i,
            gamma * i - omega * r]
    np.testing.assert_allclose(O.rhs(H.omodel(SEIRS), 0.0, [s, e, i, r], [beta, gamma, sigma, omega], [[1.0]]),
                               want, rtol=1e-15)


def test_age_risk_contact_tensor_flattening():
    # examples/sir_age_risk_stratified.py:113-115,164-166: foi_kl = beta*einsum("ijkl,ij->kl", C4, i/N)
    # with C4 = einsum("ij,kl->ikjl", C_age, C_risk).  Flattened (A*R) it is foi_q = sum_p M[q,p] x_p
    # with M = C4.reshape(AR, AR).T -- the transpose the host front-end applies.
    rng = np.random.default_rng(0)
    A, R = 3, 2
    Ca, Cr = rng.uniform(0.1, 1, (A, A)), rng.uniform(0.1, 1, (R, R))
    C4 = np.einsum("ij,kl->ikjl", Ca, Cr)
    s, i, r = (rng.uniform(1, 9, (A, R)) for _ in range(3))
    beta, gamma = 0.3, 0.1
    foi = beta * np.einsum("ijkl,ij->kl", C4, i / (s + i + r))
    want = np.concatenate([(-s * foi).ravel(), (s * foi - gamma * i).ravel(), (gamma * i).ravel()])
    m = ModelDesc(n_age=A * R)
    M = C4.reshape(A * R, A * R).T
    got = O.rhs(H.omodel(m), 0.0, np.concatenate([s.ravel(), i.ravel(), r.ravel()]), [beta, gamma], M)
    np.testing.assert_allclose(got, want, rtol=1e-13)


# ------------------------------------------------------------------ the reference's own tests
@pytest.mark.parametrize("s0,i0,r0", [(0.99, 0.01, 0.0), (0.95, 0.05, 0.0), (0.90, 0.10, 0.0), (0.80, 0.20, 0.0)])
def test_final_epidemic_size_matches_theory(s0, i0, r0):
    """reference tests/test_sir_dynamics/test_sir.py:18-65 (abs=2e-2, 300 days, r0=2, T_inf=7)."""
    ys, st, _, _ = solve(SIR, [s0, i0, r0], [2 / 7, 1 / 7], [[1.0]], 300)
    s_inf = root_scalar(lambda x: x - s0 * np.exp(-2.0 * (1 - x)), bracket=[0.0, s0], method="bisect", xtol=1e-8).root
    assert st[0] == 0
    assert ys[0, -1, 2] == pytest.approx(1 - s_inf, abs=2e-2)


@pytest.mark.parametrize("s0,i0,r0", [(0.99, 0.01, 0.0), (0.95, 0.05, 0.0), (0.90, 0.10, 0.0), (0.80, 0.20, 0.0),
                                      (0.8, 0.0, 0.2), (0.75, 0.1, 0.15)])
def test_sir_mass_conservation(s0, i0, r0):
    """reference tests/test_sir_dynamics/test_sir.py:68-100 (atol=1e-6, 120 days)."""
    ys, _, _, _ = solve(SIR, [s0, i0, r0], [2 / 7, 1 / 7], [[1.0]], 120)
    total = ys[0].sum(axis=1)
    assert np.allclose(total, total[0], atol=1e-6)


@pytest.mark.parametrize("r0,ti,tl,tw", [(2.0, 7.0, 3.0, 60.0), (3.0, 5.0, 2.0, 100.0)])
def test_seirs_endemic_equilibrium(r0, ti, tl, tw):
    """reference tests/test_seirs_dynamics/test_seirs.py:8-65 (rel=1e-2; last-100-day std < 1e-4)."""
    beta, gamma, sigma, omega = r0 / ti, 1 / ti, 1 / tl, 1 / tw
    ys, st, _, _ = solve(SEIRS, [0.99, 0.0, 0.01, 0.0], [beta, gamma, sigma, omega], [[1.0]], 1000)
    s_star = gamma / beta
    i_star = (1 - s_star) / (1 + gamma / sigma + gamma / omega)
    want = [s_star, gamma * i_star / sigma, i_star, gamma * i_star / omega]
    assert st[0] == 0
    np.testing.assert_allclose(ys[0, -1], want, rtol=1e-2)
    assert np.all(ys[0, -100:].std(axis=0) < 1e-4)


def test_seasonal_seirs_keeps_oscillating():
    """reference tests/test_seirs_seasonality_dynamics/...py:19-42 (last-100-day std > 1e-4)."""
    m = ModelDesc(n_age=1, has_e=True, has_wane=True, seasonal=True)
    ys, _, _, _ = solve(m, [0.99, 0.0, 0.01, 0.0], [2 / 7, 1 / 7, 1 / 3, 1 / 60, 0.2, 0.0, 365.0], [[1.0]], 1500)
    assert np.all(ys[0, -100:].std(axis=0) > 1e-4)


UNNORM = ModelDesc(n_age=1, normalize=False)  # tests/test_simulation/test_odes.py:17-28: beta*s*i


@pytest.mark.parametrize("days", [50, 100, 200, 300.0])
def test_expected_shapes_and_first_row(days):
    """reference tests/test_simulation/test_odes.py:45-74: (days+1) rows; ys[0] == initial state exactly."""
    ys, st, _, _ = solve(UNNORM, [99.0, 1.0, 0.0], [2 / 7, 1 / 7], [[1.0]], days)
    assert ys.shape == (1, int(days) + 1, 3) and st[0] == 0
    assert np.array_equal(ys[0, 0], np.array([99.0, 1.0, 0.0], dtype=np.float32))


@pytest.mark.parametrize("save_step", [1, 2, 3, 7])
def test_save_step_grid(save_step):
    """reference tests/test_simulation/test_odes.py:77-92 + odes.py:177-180: linspace(0,T,T//step+1)."""
    ts = synthetic.save_grid(100, save_step)
    assert ts.shape == (int(100 / save_step) + 1,) and ts[-1] == 100.0
    if save_step == 3:
        assert ts[1] == pytest.approx(100 / 33)  # NOT integer days
    ys, _, _, _ = solve(UNNORM, [99.0, 1.0, 0.0], [2 / 7, 1 / 7], [[1.0]], 100, ts=ts)
    dense, _, _, _ = solve(UNNORM, [99.0, 1.0, 0.0], [2 / 7, 1 / 7], [[1.0]], 100, ts=np.array([0.0, ts[1], 100.0]))
    assert ys.shape == (1, ts.size, 3)
    np.testing.assert_array_equal(ys[0, [0, 1, -1]], dense[0])  # same steps, same interpolant


@pytest.mark.parametrize("mask", [(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)])
def test_sub_save_indices(mask):
    """reference tests/test_simulation/test_odes.py:95-120: unsaved compartments come back empty."""
    full, _, _, _ = solve(UNNORM, [99.0, 1.0, 0.0], [2 / 7, 1 / 7], [[1.0]], 100)
    sub, _, _, _ = solve(UNNORM, [99.0, 1.0, 0.0], [2 / 7, 1 / 7], [[1.0]], 100, save_mask=mask)
    keep = [j for j in range(3) if mask[j]]
    assert sub.shape == (1, 101, len(keep))
    np.testing.assert_array_equal(sub, full[:, :, keep])


# ------------------------------------------------------------------ golden ground truth
GT = np.load(H.GOLDEN + "/ground_truth.npz")


def _case(name):
    f = [int(v) for v in GT[f"{name}/model"]]
    m = ModelDesc(f[0], f[1], bool(f[2]), bool(f[3]), bool(f[4]), f[5], bool(f[6]), bool(f[7]))
    return (m, GT[f"{name}/y0"], GT[f"{name}/params"], GT[f"{name}/contact"], float(GT[f"{name}/t1"]),
            GT[f"{name}/ts"], GT[f"{name}/ys"])


@pytest.mark.parametrize("name", [str(n) for n in GT["names"]])
@pytest.mark.parametrize("method", ["tsit5", "dopri5"])
def test_oracle_vs_scipy_ground_truth(name, method):
    m, y0, p, C, t1, ts, want = _case(name)
    scale = np.abs(want).max()
    # tight tolerances, fp64: the restated stepper + interpolant converge to the true solution
    ys, st, _, _ = solve(m, y0, p, C, t1, ts=ts, dtype=np.float64, rtol=1e-10, atol=1e-10 * scale, method=method)
    assert st[0] == 0
    assert np.abs(ys[0] - want).max() / scale < 2e-8
    # reference defaults (rtol 1e-5, atol 1e-6): global error stays at solver-tolerance level
    for dt, bound in ((np.float64, 2e-4), (np.float32, 2e-4)):
        ys, st, na, nr = solve(m, y0, p, C, t1, ts=ts, dtype=dt, method=method)
        assert st[0] == 0
        assert np.abs(ys[0] - want).max() / scale < bound, (name, method, dt)


def test_introduced_strain_vs_scipy_ground_truth_and_limits():
    """External introductions (ode_model.md term; Strain.is_introduced): the oracle against SciPy on
    the independent NumPy RHS, and the two limits -- zero percentage is the plain model, and a strain
    that starts at zero only appears once its visitors arrive."""
    plain = ModelDesc(n_age=3, n_strain=2, has_e=True, has_wane=True, has_c=True)
    m = ModelDesc(n_age=3, n_strain=2, has_e=True, has_wane=True, has_c=True, has_intro=True, intro_age_mask=(0, 0b010))
    rng = np.random.default_rng(5)
    C = synthetic.contact_matrix(rng, 3)
    y0 = np.zeros(m.state_dim)
    y0[:3] = [21780.0, 60390.0, 16830.0]
    y0[3 + 6:3 + 12:2] = [220.0, 610.0, 170.0]                  # resident strain infectious, newcomer absent
    base = np.array([1.8 / 7, 2.6 / 6, 1 / 7, 1 / 6, 1 / 3, 1 / 2.5, 1 / 120, 1 / 120])
    p = np.concatenate([base, [0.0, 60.0], [1.0, 5.0], [0.0, 0.005]])
    ts = synthetic.save_grid(200.0)
    want = H.ground_truth(m, y0, p, C, 200.0, ts)
    scale = np.abs(want).max()
    for method in ("tsit5", "dopri5"):
        ys, st, _, _ = solve(m, y0, p, C, 200.0, ts=ts, dtype=np.float64, rtol=1e-10, atol=1e-10 * scale, method=method)
        assert st[0] == 0 and np.abs(ys[0] - want).max() / scale < 2e-8
        ys32, st, _, _ = solve(m, y0, p, C, 200.0, ts=ts, method=method)
        assert st[0] == 0 and np.abs(ys32[0] - want).max() / scale < 2e-4
    newcomer_i = want[:, 3 + 6 + 1:3 + 12:2].sum(1)
    assert newcomer_i[:40].max() < 1e-6 * scale and newcomer_i[80] > 5.0 and newcomer_i.max() > 5e2
    p0 = p.copy(); p0[-2:] = 0.0                                # nobody arrives: identical to the plain model
    a, _, _, _ = solve(m, y0, p0, C, 200.0, ts=ts, dtype=np.float64)
    b, _, na, nr = solve(plain, y0, base, C, 200.0, ts=ts, dtype=np.float64)
    assert np.array_equal(a, b)


def test_blown_up_trial_steps_are_rejected_and_nan_inputs_fail_at_once():
    """diffeqsolve turns a NaN error estimate into an infinite one, i.e. a rejected step shrunk by
    factormin; only a non-finite starting state / derivative (e.g. NaN parameters) fails the solve."""
    from test_gpu_parity import fuzz_case      # the randomized sweep's generator (pure NumPy)

    m, y0, p, C, t1, ts, kw = fuzz_case(22857)  # Dopri5 proposes ~120 days right after a discontinuity point
    ys, st, na, nr = O.solve(H.omodel(m), y0, p, C, t1, ts, dtype=np.float64, **kw)
    assert st.max() == 0 and np.isfinite(ys).all() and (na[9], nr[9]) == (35, 5)
    bad = p.copy()
    bad[3, 0] = np.nan
    ys, st, na, nr = O.solve(H.omodel(m), y0, bad, C, t1, ts, dtype=np.float64, **kw)
    assert st[3] == 2 and na[3] == 0 and nr[3] == 0 and np.isinf(ys[3]).all()
    assert (np.delete(st, 3) == 0).all() and np.isfinite(np.delete(ys, 3, axis=0)).all()


def test_linear_decay_closed_form():
    """beta = 0: i(t) = i0 exp(-gamma t), r = r0 + i0 (1 - exp(-gamma t)) -- pins stepper + interpolant."""
    g = 0.2
    ts = np.linspace(0, 30, 61)
    for method in ("tsit5", "dopri5"):
        ys, _, _, _ = solve(SIR, [0.5, 0.4, 0.1], [0.0, g], [[1.0]], 30, ts=ts, dtype=np.float64, rtol=1e-11,
                            atol=1e-13, method=method)
        np.testing.assert_allclose(ys[0, :, 1], 0.4 * np.exp(-g * ts), rtol=2e-9)
        np.testing.assert_allclose(ys[0, :, 2], 0.1 + 0.4 * (1 - np.exp(-g * ts)), rtol=2e-9)
        assert np.all(ys[0, :, 0] == 0.5)


# ------------------------------------------------------------------ controller / failure semantics
def test_max_steps_is_reported_not_truncated():
    """params.py:51-55: exhausting max_steps is an error state; unreached rows stay +inf."""
    ys, st, na, nr = solve(SIR, [0.99, 0.01, 0.0], [2 / 7, 1 / 7], [[1.0]], 300, max_steps=5)
    assert st[0] == 1 and na[0] + nr[0] == 5
    assert np.isinf(ys[0, -1]).all() and np.isfinite(ys[0, 0]).all()


def test_constant_step_size_mode():
    """odes.py:115-118: ConstantStepSize with dt0 = constant_step_size, every step accepted."""
    ys, st, na, nr = solve(SIR, [0.99, 0.01, 0.0], [2 / 7, 1 / 7], [[1.0]], 100, constant_dt=0.5, dtype=np.float64)
    assert st[0] == 0 and nr[0] == 0 and na[0] == 200
    ref, _, _, _ = solve(SIR, [0.99, 0.01, 0.0], [2 / 7, 1 / 7], [[1.0]], 100, dtype=np.float64, rtol=1e-10, atol=1e-12)
    assert np.abs(ys - ref).max() < 1e-6


def test_discontinuity_points_are_hit_and_harmless_for_smooth_rhs():
    base, _, na0, _ = solve(SEIRS, [0.99, 0, 0.01, 0], [2 / 7, 1 / 7, 1 / 3, 1 / 60], [[1.0]], 200, dtype=np.float64)
    ys, st, na, _ = solve(SEIRS, [0.99, 0, 0.01, 0], [2 / 7, 1 / 7, 1 / 3, 1 / 60], [[1.0]], 200, dtype=np.float64,
                          jump_ts=[50.0, 120.5])
    assert st[0] == 0 and na[0] >= na0[0]
    assert np.abs(ys - base).max() < 5e-5


def test_a_discontinuity_point_a_step_ends_on_does_not_hide_the_later_ones():
    """Constant steps of 0.25 end exactly on t = 30: that point needs no clipping -- and must not keep the index of the next
    point to come, or every later point would be ignored (round 4; the stepper only moved the index when a clipped step
    landed).  With it out of the way the point at 45.1 clips a step: one more step than without points, the same as with that
    point alone."""
    args = (SEIRS, [0.99, 0, 0.01, 0], [2 / 7, 1 / 7, 1 / 3, 1 / 60], [[1.0]], 100)
    plain, _, na0, _ = solve(*args, dtype=np.float64, constant_dt=0.25)
    both, st, na, nr = solve(*args, dtype=np.float64, constant_dt=0.25, jump_ts=[30.0, 45.1])
    only, _, na1, _ = solve(*args, dtype=np.float64, constant_dt=0.25, jump_ts=[45.1])
    assert st[0] == 0 and nr[0] == 0 and na0[0] == 400
    assert na[0] == na1[0] and na[0] > na0[0]
    np.testing.assert_array_equal(both, only)
    assert np.abs(both - plain).max() < 1e-6


def test_batch_is_independent_of_threads_and_order():
    wl = synthetic.seirs_multi_strain(48, seed=7)
    a, _, na, _ = O.solve(H.omodel(wl.model), wl.y0, wl.params, wl.contact, 120, synthetic.save_grid(120), n_threads=1)
    b, _, nb, _ = O.solve(H.omodel(wl.model), wl.y0, wl.params, wl.contact, 120, synthetic.save_grid(120), n_threads=4)
    np.testing.assert_array_equal(a, b)
    perm = np.random.default_rng(0).permutation(48)
    c, _, _, _ = O.solve(H.omodel(wl.model), wl.y0[perm], wl.params[perm], wl.contact, 120, synthetic.save_grid(120))
    np.testing.assert_array_equal(a[perm], c)


def test_oracle_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """`-fsanitize=address,undefined` on the CPU restatement.  `make -C oracle asan` builds the oracle with
    both sanitizers; tests/c_oracle/asan_driver.c drives it through the shapes the parity suite leans on (2-age SIR at 365
    days, 8 x 4 multi-strain with seasonal forcing / the waning chain, sub-saved compartments, discontinuity points, a constant
    step, a vaccinated and a SEIP model; both precisions, both methods, ragged batches, n_save = 1).  Any out-of-bounds access
    or undefined operation aborts the driver."""
    import os
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("gcc not installed")
    odir = os.path.join(H.ROOT, "oracle")
    subprocess.run(["make", "-C", odir, "asan", "-s"], check=True)
    exe = str(tmp_path / "asan_driver")
    build = subprocess.run(["gcc", "-std=c11", "-O1", "-g", "-fsanitize=address,undefined", "-I", odir,
                            os.path.join(H.ROOT, "tests", "c_oracle", "asan_driver.c"), "-L", odir, "-ldynode_oracle_asan", "-lm", "-fopenmp",
                            f"-Wl,-rpath,{odir}", "-o", exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"))
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "bad=0" in run.stdout and "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
