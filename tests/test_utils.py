"""dynode_amd.utils against the closed forms and literals of the reference's
tests/test_utils/test_utils.py (host-side helpers: key handling, splines, simulation days)."""

import datetime

import numpy as np
import pytest
import torch

from dynode_amd import utils
from dynode_amd.infer import distributions as dist

TIMES = list(range(-2, 15)) + [100]


def test_drop_keys_with_substring():
    d = {"a": np.ones((5, 20)), "b1234": np.ones((5, 20))}
    out = utils.drop_keys_with_substring(d, drop_s="b1")
    assert out is d and "b1234" not in out and "a" in out


def test_base_equation_is_the_cubic():
    for t in TIMES:                                          # 5 + t + 2 t^2 + 3 t^3
        assert utils.base_equation(t, np.array([5, 1, 2, 3])) == 5 + t + 2 * t**2 + 3 * t**3


@pytest.mark.parametrize("coef", [(1, 1, 1), (1, 2, 3)])
def test_conditional_knots_switch_on_after_their_knot(coef):
    knots = np.array([0, 5, 10])
    for t in TIMES:
        want = sum(c * (t - k) ** 3 * (t > k) for c, k in zip(coef, knots))
        assert utils.conditional_knots(t, knots, np.array(coef)) == want


def test_cubic_spline_scalar_and_stacked():
    knots, base, kc = np.array([0, 5, 10]), np.array([1, 2, 3, 4]), np.array([5, 6, 7])
    for t in TIMES:
        want = 1 + 2 * t + 3 * t**2 + 4 * t**3 + sum(c * (t - k) ** 3 * (t > k) for c, k in zip(kc, knots))
        assert utils.evaluate_cubic_spline(t, knots, base, kc) == want
    # two (age x dose) rows at once, torch tensors accepted
    bases = torch.tensor([[1, 2, 3, 4], [1, 2, 3, 4]])
    locs = torch.tensor([[0, 2, 4, 6], [0, 2, 4, 6]])
    coefs = torch.tensor([[1, 1, 1, 1], [1, 2, 3, -4]])
    for t in range(-5, 5):
        got = utils.evaluate_cubic_spline(t, locs, bases, coefs).flatten()
        cubic = 1 + 2 * t + 3 * t**2 + 4 * t**3
        for row in range(2):
            want = cubic + sum(int(c) * (t - int(k)) ** 3 * (t > int(k)) for c, k in zip(coefs[row], locs[row]))
            assert got[row] == want


def test_identify_distribution_indexes():
    parameters = {"test": [0, dist.Normal(), 2], "example": dist.Normal(), "no-sample": 5,
                  "grid": [[1.0, dist.Uniform(0, 1)], [dist.Beta(2, 2), 3.0]]}
    idx = utils.identify_distribution_indexes(parameters)
    assert idx["test_1"] == {"sample_name": "test", "sample_idx": (1,)}
    assert idx["example"] == {"sample_name": "example", "sample_idx": None}
    assert "no-sample" not in idx
    assert idx["grid_0_1"]["sample_idx"] == (0, 1) and idx["grid_1_0"]["sample_idx"] == (1, 0) and len(idx) == 4


@pytest.mark.parametrize("make", [np.ones, torch.ones], ids=["numpy", "torch"])
def test_flatten_list_parameters(make):
    flat = utils.flatten_list_parameters({"test": make((4, 20, 5)), "scalar_site": make((4, 20))})
    assert "test" not in flat and flat["scalar_site"].shape == (4, 20)
    assert all(tuple(flat[f"test_{i}"].shape) == (4, 20) for i in range(5))
    flat = utils.flatten_list_parameters({"test": make((4, 20, 5, 2))})
    assert sorted(flat) == sorted(f"test_{i}_{j}" for i in range(5) for j in range(2))
    assert all(tuple(v.shape) == (4, 20) for v in flat.values())


def test_vectorize_objects():
    class Obj:
        def __init__(self, value):
            self.value = value

    objs = [Obj(i) for i in range(5)]
    assert utils.vectorize_objects(objs, "value") == [0, 1, 2, 3, 4]
    assert utils.vectorize_objects(objs, "value", filter=lambda o: o.value > 2) == [3, 4]
    with pytest.raises(AttributeError):
        utils.vectorize_objects(objs, "non_existing")


def test_simulation_days():
    """(the reference's epiweek helpers, datetime_utils.py:36-61,91-106, are out of scope: SURVEY section 2)"""
    init = datetime.date(2022, 10, 15)
    assert utils.sim_day_to_date(21, init) == init + datetime.timedelta(days=21)
    assert utils.date_to_sim_day(datetime.date(2022, 11, 5), init) == 21
