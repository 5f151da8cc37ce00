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
