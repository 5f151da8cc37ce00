"""Pin every Runge-Kutta constant compiled into the oracle by the RK order conditions.

diffrax (the third-party home of the reference's stepper, pyproject.toml:12) is absent, so
the tableaux are restated from the published papers; a wrong digit cannot satisfy the order
conditions to 1e-14.  The HIP kernel carries the same constants (solve_kernel.hpp); the fp64
GPU-vs-oracle test (1e-12) transfers this pin to the device code.
"""

import numpy as np
import pytest

import helpers as H

O = H.O


def _conditions(A, c, w, order, theta=1.0):
    Ac = A @ c
    th = theta
    out = [w.sum() - th, w @ c - th**2 / 2, w @ c**2 - th**3 / 3, w @ Ac - th**3 / 6,
           w @ c**3 - th**4 / 4, w @ (c * Ac) - th**4 / 8, w @ (A @ c**2) - th**4 / 12,
           w @ (A @ Ac) - th**4 / 24]
    if order >= 5:
        out += [w @ c**4 - 1 / 5, w @ (c**2 * Ac) - 1 / 10, w @ (c * (A @ c**2)) - 1 / 15,
                w @ (c * (A @ Ac)) - 1 / 30, w @ (Ac * Ac) - 1 / 20, w @ (A @ c**3) - 1 / 20,
                w @ (A @ (c * Ac)) - 1 / 40, w @ (A @ (A @ c**2)) - 1 / 60,
                w @ (A @ (A @ Ac)) - 1 / 120]
    return np.array(out)


@pytest.mark.parametrize("method", ["tsit5", "dopri5"])
def test_tableau_order_conditions(method):
    c, A, berr, _ = O.tableau(method)
    assert np.allclose(A.sum(1), c, atol=1e-15)          # row sums
    b = A[6]                                              # FSAL: last row == b
    assert np.abs(_conditions(A, c, b, 5)).max() < 5e-15       # 5th order solution
    assert np.abs(_conditions(A, c, b - berr, 4)).max() < 5e-15  # 4th order embedded
    assert abs(berr.sum()) < 1e-15


def test_tsit5_dense_output_is_fourth_order_and_ends_on_b():
    c, A, _, _ = O.tableau("tsit5")
    assert np.allclose(O.tsit5_dense_weights(1.0), A[6], atol=1e-14)
    assert np.all(O.tsit5_dense_weights(0.0) == 0.0)      # ts[0] == t0 returns y0 exactly
    for th in (0.1, 0.3, 0.5, 0.77):
        w = O.tsit5_dense_weights(th)
        assert np.abs(_conditions(A, c, w, 4, theta=th)).max() < 5e-15


def test_dopri5_midpoint_weights_are_fourth_order():
    c, A, _, cmid = O.tableau("dopri5")
    assert np.abs(_conditions(A, c, cmid, 4, theta=0.5)).max() < 5e-15


def test_tsit5_dense_polynomial_in_the_kernel_equals_the_factored_weights():
    """solve_kernel.hpp evaluates the dense output as a polynomial in theta (Tab<0>::bp, formed once per accepted step); the
    table must be the expansion of the interpolant's factored weights the oracle (and diffrax) evaluate per row."""
    import os
    import re

    src = open(os.path.join(os.path.dirname(__file__), "..", "dynode_amd", "csrc", "solve_kernel.hpp")).read()
    body = re.search(r"static constexpr double bp\[7\]\[3\] = \{(.*?)\};", src, re.S).group(1)
    bp = np.array([float(x) for x in re.findall(r"-?\d+\.\d+(?:e-?\d+)?", body)]).reshape(7, 3)
    for th in np.linspace(0.0, 1.0, 41):
        w = bp @ np.array([th**2, th**3, th**4])
        w[0] += th                                   # the only theta^1 term: y'(t_prev) = f_1
        assert np.abs(w - O.tsit5_dense_weights(th)).max() < 3e-14   # coefficients up to 88: cancellation at the 1e-14 level in float64
    _, A, _, _ = O.tableau("tsit5")
    assert np.abs(bp.sum(1) + np.eye(7)[0] - A[6]).max() < 3e-14      # theta = 1: the solution weights


def test_seasonal_sine_of_the_float32_kernels():
    """`Mth<float>::sin` (csrc/solve_kernel.hpp): x - n pi by three Cody-Waite FMAs, the odd Taylor polynomial to x^13, the sign
    from n's parity.  The same arithmetic in NumPy (an FMA = the product and sum in float64, rounded once) against float64:
    within 1.3e-7 everywhere a seasonal argument w t + phase can be (years of a yearly cycle and far beyond)."""
    f32 = np.float32

    def fma(a, b, c):
        return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(f32) if np.isscalar(b) else \
            (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)

    def kernel_sin(x):
        x = x.astype(f32)
        n = np.rint(x * f32(0.318309886183790671538)).astype(f32)
        r = fma(n, f32(-3.140625), x)
        r = fma(n, f32(-9.67502593994140625e-4), r)
        r = fma(n, f32(-1.509957990978376432e-7), r)
        r2 = (r * r).astype(f32)
        p = np.full_like(x, f32(1.0 / 6227020800.0))
        for c in (-1.0 / 39916800.0, 1.0 / 362880.0, -1.0 / 5040.0, 1.0 / 120.0, -1.0 / 6.0):
            p = fma(p, r2, np.full_like(x, f32(c)))
        s = fma(r, (p * r2).astype(f32), r)
        return np.where(n.astype(np.int64) & 1, -s, s).astype(f32)

    rng = np.random.default_rng(0)
    for lo, hi in ((-7.0, 7.0), (0.0, 3000.0), (-2e4, 2e4)):
        x = rng.uniform(lo, hi, 400_000).astype(f32)
        assert np.abs(kernel_sin(x).astype(np.float64) - np.sin(x.astype(np.float64))).max() < 1.3e-7
    edge = np.array([0.0, -0.0, np.pi, -np.pi, np.pi / 2, 1e-30, 2e5], dtype=f32)
    assert np.abs(kernel_sin(edge).astype(np.float64) - np.sin(edge.astype(np.float64))).max() < 2e-7
