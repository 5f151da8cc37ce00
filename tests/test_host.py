"""Host logic of the drop-in surface, CPU only (no compute calls).

Mirrors the reference's own tests where they pin the boundary: tests/test_config/*,
tests/test_infer/test_sample.py, tests/test_simulation/test_odes.py (error behaviour),
tests/test_age_risk_groups/test_age_risk_groups.py (contact tensor literals).
"""

import numpy as np
import pytest
import torch
from pydantic import ValidationError

import helpers as H
from dynode_amd import (Bin, Compartment, DeterministicParameter, Dimension, Dopri5, Params, SimulationConfig,
                        SolverParams, Strain, TransmissionParams, Tsit5, rhs, simulate)
from dynode_amd.config import AgeBin
from dynode_amd.infer import distributions as dist
from dynode_amd.infer import handlers, resolve_deterministic, sample_distributions, sample_then_resolve
from dynode_amd.simulation import build_saveat
from examples import seirs as ex_seirs
from examples import seirs_multi_strain_age_stratified as ex_ms
from examples import sir as ex_sir
from examples import sir_age_risk_stratified as ex_risk
from examples import sir_age_stratified as ex_age

O = H.O


# ------------------------------------------------------------------ config / idx
def test_idx_namespace_is_recursive_ints():
    """reference tests/test_config/test_simulation_config.py:42-48, test_compartment.py:21-22."""
    cfg = ex_ms.get_config()
    idx = cfg.idx
    assert (idx.s, idx.e, idx.i, idx.r, idx.c) == (0, 1, 2, 3, 4)
    assert idx.e.age == 0 and idx.e.strain == 1          # axis numbers WITHOUT the time axis
    assert idx.e.age.old == 1 and idx.e.strain.C == 2
    assert isinstance(idx.c, int) and cfg.get_compartment("e").shape == (2, 3)
    with pytest.raises(AssertionError):
        cfg.get_compartment("nope")


def test_config_validators():
    d = Dimension(name="age", bins=[Bin(name="a"), Bin(name="b")])
    with pytest.raises(ValidationError):
        Bin(name="1bad")
    with pytest.raises(ValidationError):
        Bin(name="has space")
    with pytest.raises(ValidationError):
        Dimension(name="age", bins=[])
    with pytest.raises(ValidationError):
        Dimension(name="age", bins=[Bin(name="a"), Bin(name="a")])
    with pytest.raises(ValidationError):
        Dimension(name="age", bins=[AgeBin(0, 4), AgeBin(6, 9)])      # gap
    assert Dimension(name="age", bins=[AgeBin(0, 4), AgeBin(5, 9)]).idx.a5_9 == 1
    with pytest.raises(ValidationError):
        Compartment(name="s", dimensions=[d, d])
    cfg = ex_sir.get_config()
    with pytest.raises(ValidationError):
        SimulationConfig(compartments=cfg.compartments + [cfg.compartments[0]], initializer=cfg.initializer,
                         parameters=cfg.parameters)
    with pytest.raises(ValidationError):
        TransmissionParams(strains=[], strain_interactions={})
    with pytest.raises(ValidationError):
        TransmissionParams(strains=[Strain(strain_name="a", r0=2.0, infectious_period=7.0)],
                           strain_interactions={"b": {"b": 1.0}})


def test_solver_params_defaults_and_validation():
    """reference src/dynode/config/params.py:24-67, tests/test_config/test_params.py:133-161."""
    sp = SolverParams()
    assert sp.solver_method == Tsit5() and sp.ode_solver_rel_tolerance == 1e-5
    assert sp.ode_solver_abs_tolerance == 1e-6 and sp.max_steps == 10**6
    assert sp.constant_step_size == 0 and sp.discontinuity_points == []
    assert SolverParams(solver_method=Dopri5()).solver_method.method == "dopri5"
    for bad in (dict(ode_solver_rel_tolerance=0.0), dict(ode_solver_abs_tolerance=-1e-6), dict(max_steps=0),
                dict(constant_step_size=-1.0)):
        with pytest.raises(ValidationError):
            SolverParams(**bad)


# ------------------------------------------------------------------ save grid
def test_build_saveat_grid_and_mask():
    """reference odes.py:177-198: linspace(0, T, T//step + 1); step <= 0 -> 1."""
    assert build_saveat(0.0, 100).ts.shape == (101,)
    g = build_saveat(0.0, 100, 3).ts
    assert g.shape == (34,) and g[-1] == 100.0 and g[1] == pytest.approx(100 / 33)
    assert build_saveat(0.0, 300.0, 0).ts.shape == (301,)
    assert build_saveat(0.0, 100, 1, (0, 2), 3).mask == (True, False, True)
    assert build_saveat(0.0, 100, 1, None, 3).mask is None
    assert build_saveat(0.0, 100, 1, (0, 7), 3).mask is None   # out-of-range index: printed, all saved


# ------------------------------------------------------------------ RHS descriptors
def _flat(parts):
    return np.concatenate([np.asarray(p, float).ravel() for p in parts])


@pytest.mark.parametrize("case", ["sir", "sir_age", "seirs", "seasonal", "multi", "risk"])
def test_descriptor_host_evaluation_matches_oracle_rhs(case):
    rng = np.random.default_rng(1)
    if case == "sir":
        ode, st = rhs.sir_ode, tuple(rng.uniform(0.1, 1, (3, 1)))
        p = rhs.SIR_ODEParams(beta=np.array(0.3), gamma=np.array(0.1))
    elif case == "sir_age":
        ode, st = rhs.sir_ode, tuple(rng.uniform(1, 9, (3, 4)))
        p = rhs.SIR_ODEParams(beta=0.3, gamma=0.1, contact_matrix=rng.uniform(0.1, 1, (4, 4)))
    elif case == "seirs":
        ode, st = rhs.seirs_ode, tuple(rng.uniform(0.1, 1, (4, 1)))
        p = rhs.SEIRS_ODEParams(beta=0.3, gamma=0.1, sigma=0.3, omega=0.02)
    elif case == "seasonal":
        ode, st = rhs.seirs_ode_seasonal, tuple(rng.uniform(0.1, 1, (4, 1)))
        p = rhs.SEIRS_Seasonal_ODEParams(beta=0.3, gamma=0.1, sigma=0.3, omega=0.02,
                                         seasonality_params=rhs.SeasonalityParams(0.2, 0.4, 365.0))
    elif case == "multi":
        ode = rhs.seirs_multi_strain_ode
        st = (rng.uniform(1, 9, 2),) + tuple(rng.uniform(0.1, 3, (4, 2, 3)))
        p = rhs.SEIRS_MultiStrain_ODEParams(beta=rng.uniform(.2, .4, 3), gamma=rng.uniform(.1, .2, 3),
                                            sigma=rng.uniform(.2, .5, 3), omega=rng.uniform(.01, .03, 3),
                                            contact_matrix=rng.uniform(0.1, 1, (2, 2)))
    else:
        ode, st = rhs.sir_age_risk_ode, tuple(rng.uniform(1, 9, (3, 3, 2)))
        p = rhs.SIR_ODEParams(beta=0.3, gamma=0.1, contact_matrix=ex_risk.contact_tensor(rng.uniform(.1, 1, (3, 3)),
                                                                                          rng.uniform(.1, 1, (2, 2))))
    pk = ode.pack(st, p)
    got = _flat(ode(11.0, st, p))
    want = O.rhs(H.omodel(pk.model), 11.0, pk.y0, pk.params[0], pk.contact)
    np.testing.assert_allclose(got, want, rtol=1e-13)
    assert [g.shape for g in ode(11.0, st, p)] == [np.asarray(s).shape for s in st]


def test_age_risk_contact_tensor_literals():
    """reference tests/test_age_risk_groups/test_age_risk_groups.py: einsum("ij,kl->ikjl") known answers."""
    np.testing.assert_allclose(ex_risk.contact_tensor([[2.0]], [[3.0]]), np.full((1, 1, 1, 1), 6.0), atol=1e-6)
    age = np.array([[1.0, 0.5, 0.2], [0.5, 1.0, 0.3], [0.2, 0.3, 1.0]])
    t = ex_risk.contact_tensor(age, [[1.0]])
    assert t.shape == (3, 1, 3, 1) and np.allclose(t[:, 0, :, 0], age, atol=1e-6)
    risk = np.array([[1.0, 0.4], [0.4, 1.0]])
    t = ex_risk.contact_tensor(age, risk)
    assert t.shape == (3, 2, 3, 2)
    for i in range(3):
        for k in range(2):
            np.testing.assert_allclose(t[i, k], np.outer(age[i], risk[k]), atol=1e-6)
    state = ex_risk.get_config().initializer.get_initial_state()
    assert state[0].shape == (3, 2) and np.isclose(sum(s.sum() for s in state), 1000.0)


def test_pack_batches_and_rejects_bad_shapes():
    st = ex_ms.get_config().initializer.get_initial_state(ex_ms.get_config())
    p = ex_ms.get_odeparams(ex_ms.get_config())
    pk = rhs.seirs_multi_strain_ode.pack(st, p)
    assert pk.batch is None and pk.y0.shape == (26,) and pk.params.shape == (1, 12)
    p.beta = np.tile(p.beta, (7, 1))
    pk = rhs.seirs_multi_strain_ode.pack(st, p)
    assert pk.batch == 7 and pk.params.shape == (7, 12) and pk.y0.shape == (26,)
    pk = rhs.seirs_multi_strain_ode.pack(tuple(np.tile(a, (7,) + (1,) * a.ndim) for a in st), p)
    assert pk.y0.shape == (7, 26)
    with pytest.raises(ValueError):
        rhs.seirs_multi_strain_ode.pack(tuple(np.tile(a, (5,) + (1,) * a.ndim) for a in st), p)   # 5 vs 7
    with pytest.raises(ValueError):
        rhs.seirs_multi_strain_ode.pack(st[:4], p)
    with pytest.raises(ValueError):
        rhs.sir_ode.pack((np.ones(3), np.ones(3), np.ones(3)), rhs.SIR_ODEParams(beta=.3, gamma=.1))  # no contact


def test_get_odeparams_formulas():
    """A6: beta = r0/T_inf, gamma = 1/T_inf, sigma = 1/latent, omega = 1/waning (examples get_odeparams)."""
    p = ex_sir.get_odeparams(ex_sir.get_config(r_0=3.0, infectious_period=6.0))
    assert float(p.beta) == 0.5 and float(p.gamma) == pytest.approx(1 / 6)
    q = ex_seirs.get_seirs_odeparams(ex_seirs.get_config())
    assert (float(q.sigma), float(q.omega)) == (pytest.approx(1 / 3), pytest.approx(1 / 60))
    m = ex_ms.get_odeparams(ex_ms.get_config())
    np.testing.assert_allclose(m.beta, [2 / 7, 2.5 / 6, 1.8 / 8])
    np.testing.assert_allclose(m.omega, [1 / 60, 1 / 80, 1 / 50])
    y0 = ex_ms.get_config().initializer.get_initial_state(ex_ms.get_config())
    np.testing.assert_allclose(y0[2].sum(1), [7.5, 2.5])                      # i0 split over strains by r0
    np.testing.assert_allclose(y0[2][0] / y0[2][0].sum(), np.array([2.0, 2.5, 1.8]) / 6.3)
    a = ex_age.get_config().initializer.get_initial_state()
    np.testing.assert_allclose(a[0], [742.5, 247.5])
    np.testing.assert_allclose(ex_age.get_config().parameters.transmission_params.contact_matrix, [[.7, .3], [.3, .7]])


# ------------------------------------------------------------------ simulate(): error behaviour
def test_simulate_validation_errors_match_reference():
    """reference odes.py:93-112: TypeError for non-array state, AssertionError for param type / duration."""
    cfg = ex_sir.get_config()
    y0, p, sp = cfg.initializer.get_initial_state(), ex_sir.get_odeparams(cfg), cfg.parameters.solver_params
    with pytest.raises(TypeError):
        simulate(rhs.sir_ode, 10, ([0.9], [0.1], [0.0]), p, sp)
    with pytest.raises(AssertionError):
        simulate(rhs.sir_ode, 10, y0, rhs.SEIRS_ODEParams(beta=1, gamma=1, sigma=1, omega=1), sp)

    class Sub(rhs.SIR_ODEParams):  # exact `is`, not isinstance (odes.py:103)
        pass
    with pytest.raises(AssertionError):
        simulate(rhs.sir_ode, 10, y0, Sub(beta=1, gamma=1), sp)
    with pytest.raises(AssertionError):
        simulate(rhs.sir_ode, "10", y0, p, sp)
    with pytest.raises(TypeError):
        simulate(lambda t, y, a: y, 10, y0, p, sp)          # arbitrary callables are rejected loudly


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_simulate_without_gpu_fails_loudly_no_cpu_fallback():
    cfg = ex_sir.get_config()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        simulate(rhs.sir_ode, 10, cfg.initializer.get_initial_state(), ex_sir.get_odeparams(cfg),
                 cfg.parameters.solver_params)


# ------------------------------------------------------------------ sample / resolve (site names)
def test_sample_site_naming_rules():
    """reference tests/test_infer/test_sample.py:17-151: a, b_1, c_0, d_nested_dict; prefixes."""
    params = {"a": dist.Normal(), "b": [0, dist.Normal(), 2], "c": [dist.Normal()], "d": {"nested_dict": dist.Normal()},
              "e": 5.0}
    with handlers.seed(1), handlers.trace() as tr:
        out = sample_distributions(params)
    assert list(tr.sites) == ["a", "b_1", "c_0", "d_nested_dict"]
    assert out["e"] == 5.0 and out["b"][0] == 0 and isinstance(out["b"][1], torch.Tensor)
    with handlers.seed(1), handlers.trace() as tr:
        sample_distributions(params, _prefix="fit_")
    assert list(tr.sites) == ["fit_a", "fit_b_1", "fit_c_0", "fit_d_nested_dict"]
    with pytest.raises(RuntimeError):
        sample_distributions({"a": dist.Normal()})            # no randomness supplied
    assert float(sample_distributions({"a": dist.Normal()}, rng_key=3)["a"]) == float(
        sample_distributions({"a": dist.Normal()}, rng_key=3)["a"])


def test_resolve_deterministic_including_slices():
    params = {"x": 3.0, "y": DeterministicParameter("x"), "x_lst": [0.0, 1.5, 2.0],
              "y_lst": [0.0, DeterministicParameter("x_lst", index=1), 2.0],
              "z": DeterministicParameter("x_lst", index=slice(0, 2)),
              "w": DeterministicParameter("x", transform=lambda v: 2 * v)}
    with handlers.trace() as tr:
        out = resolve_deterministic(params, root_params=params)
    assert out["y"] == 3.0 and out["y_lst"] == [0.0, 1.5, 2.0] and out["z"] == [0.0, 1.5] and out["w"] == 6.0
    assert list(tr.sites) == ["y", "y_lst_1", "z", "w"]
    with pytest.raises(Exception, match="Was unable to find"):
        resolve_deterministic({"y": DeterministicParameter("missing")}, root_params={})


def test_sample_then_resolve_copies_and_names_strain_sites():
    """A15: posterior keys strains_0_r0 / strains_0_infectious_period (sir_infer_parameters.py:47-58)."""
    tp = TransmissionParams(
        strains=[Strain(strain_name="swo9",
                        r0=dist.TransformedDistribution(dist.Beta(0.5, 0.5), dist.transforms.AffineTransform(1.5, 1)),
                        infectious_period=dist.TruncatedNormal(loc=8, scale=2, low=2, high=15))],
        strain_interactions={"swo9": {"swo9": 1.0}}, contact_matrix=np.eye(2))
    with handlers.seed(0, batch=256), handlers.trace() as tr:
        out = sample_then_resolve(tp)
    assert list(tr.sites) == ["strains_0_r0", "strains_0_infectious_period"]
    r0, ti = out.strains[0].r0, out.strains[0].infectious_period
    assert r0.shape == (256,) and float(r0.min()) >= 1.5 and float(r0.max()) <= 2.5
    assert float(ti.min()) >= 2 and float(ti.max()) <= 15
    assert isinstance(tp.strains[0].r0, dist.Distribution)          # the original is untouched
    with handlers.substitute({"strains_0_r0": 2.0}), handlers.seed(0):
        fixed = sample_then_resolve(tp)
    assert float(fixed.strains[0].r0) == 2.0


def test_distribution_log_probs_against_scipy():
    from scipy import stats
    x = np.array([1.6, 2.0, 2.4])
    d = dist.TransformedDistribution(dist.Beta(0.5, 0.5), dist.transforms.AffineTransform(1.5, 1))
    np.testing.assert_allclose(d.log_prob(x), stats.beta(0.5, 0.5, loc=1.5).logpdf(x), rtol=1e-10)
    t = dist.TruncatedNormal(8, 2, low=2, high=15)
    np.testing.assert_allclose(t.log_prob([3.0, 8.0, 14.0]), stats.truncnorm(-3, 3.5, loc=8, scale=2).logpdf([3, 8, 14]),
                               rtol=1e-10)
    assert float(t.log_prob(1.0)) == -np.inf
    np.testing.assert_allclose(dist.Poisson([2.0, 30.0]).log_prob([3.0, 25.0]), stats.poisson([2.0, 30.0]).logpmf([3, 25]),
                               rtol=1e-10)
    assert float(t.median) == pytest.approx(stats.truncnorm(-3, 3.5, loc=8, scale=2).median(), rel=1e-9)


def test_reference_sample_tests_object_arrays_objects_and_prefixes():
    """The remaining cases of reference tests/test_infer/test_sample.py: distributions inside NumPy
    object arrays and pydantic objects, sample_then_resolve with a prefix, resolved value identity."""
    from pydantic import BaseModel, ConfigDict

    from dynode_amd.infer import sample_then_resolve

    params = {"a": dist.Normal(), "b": [1, dist.Normal()], "c": np.array([dist.Normal(), 1]),
              "d": {"nested_dict": dist.Normal()}, "e": DeterministicParameter("a")}
    with handlers.trace() as tr:
        out = sample_then_resolve(params, rng_key=0, _prefix="test_")
    assert list(tr.sites) == ["test_a", "test_b_1", "test_c_0", "test_d_nested_dict", "test_e"]
    assert isinstance(out["b"][1], torch.Tensor) and isinstance(out["c"][0], torch.Tensor) and out["c"][1] == 1
    assert float(out["e"]) == float(out["a"])

    class Holder(BaseModel):
        model_config = ConfigDict(arbitrary_types_allowed=True)
        x: object

    held = sample_distributions(Holder(x=dist.Normal()), rng_key=0)
    assert isinstance(held.x, torch.Tensor)
    both = sample_then_resolve({"a": dist.Normal(), "b": DeterministicParameter("a")}, rng_key=0)
    assert isinstance(both["a"], torch.Tensor) and both["a"] == both["b"]


# ---- MCMCProcess kwargs: honoured or refused, never dropped (reference inference.py:127-131,149-162 forwards them verbatim)
def _proc(**kw):
    from dynode_amd.infer.inference import MCMCProcess

    return MCMCProcess(numpyro_model=lambda: None, num_warmup=10, num_samples=20, num_chains=2, nuts_max_tree_depth=5,
                       progress_bar=False, **kw)


def test_vector_valued_latent_sites_take_consecutive_coordinates():
    """A latent site may be tensor-valued with element-wise independent distributions (numpyro: a distribution with a
    batch shape, e.g. one prior per strain): its elements are consecutive unconstrained coordinates, the model sees
    ``[chains, *shape]``, the log joint sums the elements.  Checked against the densities written out chain by chain."""
    from dynode_amd.infer.inference import Potential, init_to_median

    gen = torch.Generator().manual_seed(0)
    y = torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64) + 0.5 * torch.randn(40, 3, dtype=torch.float64, generator=gen)

    def model(y):
        mu = handlers.sample("mu", dist.Normal(torch.zeros(3), 10.0))
        s = handlers.sample("s", dist.Uniform(0.1, 2.0))
        w = handlers.sample("w", dist.TruncatedNormal(torch.tensor([[1.0, 2.0], [3.0, 4.0]]), 1.0, low=0.0, high=9.0))
        handlers.sample("y", dist.Normal(mu[..., None, :] + 0.0 * w.sum((-1, -2))[..., None, None], s[..., None, None]), obs=y)

    pot = Potential(model, dict(y=y), 0, torch.device("cpu"))
    assert pot.dim == 8 and pot.shapes == {"mu": (3,), "s": (), "w": (2, 2)} and dict(pot.slices) == {"mu": (0, 3), "s": (3, 1), "w": (4, 4)}
    assert not pot.scalar_sites and pot.site_table is None
    z = pot.initial(5, init_to_median, 1)
    assert tuple(z.shape) == (5, 8)
    x = pot.constrain(z)
    assert {k: tuple(v.shape) for k, v in x.items()} == {"mu": (5, 3), "s": (5,), "w": (5, 2, 2)}
    assert torch.equal(x["mu"], z[:, :3]) and bool(((x["w"] > 0) & (x["w"] < 9)).all())
    u, g = pot.potential_and_grad(z)
    assert tuple(u.shape) == (5,) and tuple(g.shape) == (5, 8) and bool(torch.isfinite(g).all())
    for c in range(5):
        want = (dist.Normal(0.0, 10.0).log_prob(x["mu"][c]).sum() + dist.Uniform(0.1, 2.0).log_prob(x["s"][c])
                + pot.bij["s"].log_abs_det_jacobian(z[c, 3])
                + dist.TruncatedNormal(torch.tensor([[1.0, 2.0], [3.0, 4.0]]), 1.0, low=0.0, high=9.0).log_prob(x["w"][c]).sum()
                + pot.bij["w"].log_abs_det_jacobian(z[c, 4:]).sum() + dist.Normal(x["mu"][c], x["s"][c]).log_prob(y).sum())
        assert abs(float(want) + float(u[c])) < 1e-9 * abs(float(want))
    # a value whose shape is not the distribution's batch shape is refused at construction
    with pytest.raises(ValueError, match="batch shape"):
        Potential(lambda: handlers.sample("a", _Odd()), {}, 0, torch.device("cpu"))


class _Odd(dist.Normal):
    """A distribution whose draws do not have its batch shape (for the refusal above)."""

    def sample(self, rng, sample_shape=()):
        return torch.zeros(tuple(sample_shape) + (2,), dtype=torch.float64)


def test_nuts_kwargs_the_reference_sets_itself_raise_like_numpyro_would():
    # reference: NUTS(model, dense_mass=True, max_tree_depth=..., init_strategy=..., **nuts_kwargs) -> duplicate keyword
    for key in ("dense_mass", "max_tree_depth", "init_strategy"):
        with pytest.raises(TypeError, match="multiple values"):
            _proc(nuts_kwargs={key: True})._check_kwargs()
    for key in ("num_warmup", "num_samples", "num_chains", "progress_bar"):
        with pytest.raises(TypeError, match="multiple values"):
            _proc(mcmc_kwargs={key: 1})._check_kwargs()


def test_unknown_or_unimplemented_kwargs_are_refused_not_dropped():
    with pytest.raises(TypeError, match="unsupported NUTS argument"):
        _proc(nuts_kwargs={"trajectory_length": 3.0})._check_kwargs()
    with pytest.raises(TypeError, match="unsupported MCMC argument"):
        _proc(mcmc_kwargs={"postprocess_fn": print})._check_kwargs()
    with pytest.raises(NotImplementedError, match="adapt_mass_matrix"):
        _proc(nuts_kwargs={"adapt_mass_matrix": False})._check_kwargs()
    with pytest.raises(ValueError, match="chain_method"):
        _proc(mcmc_kwargs={"chain_method": "pmap"})._check_kwargs()
    with pytest.raises(ValueError, match="thinning"):
        _proc(mcmc_kwargs={"thinning": 3})._check_kwargs()          # 20 draws are not a multiple of 3


def test_numpyro_kwargs_at_supported_values_pass():
    p = _proc(nuts_kwargs={"target_accept_prob": 0.9, "step_size": 0.5, "adapt_step_size": True, "adapt_mass_matrix": True,
                           "regularize_mass_matrix": True, "find_heuristic_step_size": False, "forward_mode_differentiation": True},
              mcmc_kwargs={"chain_method": "vectorized", "thinning": 4, "jit_model_args": True, "adaptation": "pooled"})
    assert p._check_kwargs() == 4
    assert _proc()._check_kwargs() == 1


def test_monomial_parameter_maps_are_recognised_exactly_or_not_at_all():
    """infer/folded.py: the structure test behind the three-launch potential.  The reference's get_odeparams family
    (examples/sir.py:87-92, seirs_multi_strain_age_stratified.py:187-209) is monomial in the sampled values."""
    from dynode_amd.infer.folded import _fit_monomials

    g = torch.Generator().manual_seed(0)
    x = torch.rand((12, 3), generator=g, dtype=torch.float64) * 4.0 + 0.5
    r0, t_inf, t_wane = x.unbind(1)
    params = torch.stack([r0 / t_inf, 1.0 / t_inf, 1.0 / t_wane, torch.full_like(r0, 0.25), torch.full_like(r0, -2.0),
                          3.0 * torch.sqrt(r0) * t_wane, torch.zeros_like(r0)], dim=1)
    coef, expo = _fit_monomials(x, params)
    want = torch.tensor([[1, -1, 0], [0, -1, 0], [0, 0, -1], [0, 0, 0], [0, 0, 0], [0.5, 0, 1], [0, 0, 0]], dtype=torch.float64)
    assert torch.equal(expo, want)
    assert torch.allclose(coef, torch.tensor([1, 1, 1, 0.25, -2.0, 3.0, 0.0], dtype=torch.float64), rtol=1e-12, atol=0)
    # sums, shifted powers and sign changes are refused, not approximated
    assert _fit_monomials(x, torch.stack([r0 + t_inf, 1.0 / t_inf], dim=1)) is None
    assert _fit_monomials(x, torch.stack([1.0 / (t_inf + 1.0)], dim=1)) is None
    assert _fit_monomials(x, torch.stack([r0 - 2.0], dim=1)) is None
    # a site that takes non-positive values cannot carry a power; constants next to it still fit
    xn = x.clone()
    xn[:, 2] -= 3.0
    assert _fit_monomials(xn, torch.stack([r0 / t_inf, torch.full_like(r0, 7.0)], dim=1)) is not None
    assert _fit_monomials(xn, torch.stack([xn[:, 2] * r0], dim=1)) is None


def test_on_demand_seip_lane_mappings_follow_the_dispatch_rules():
    """jit._seip_wave_group / _features (host logic, no compiler): which SEIP lane mapping an on-demand build instantiates and
    the feature word it registers -- the words `select_seip_entry` (csrc/dynode_hip.hip) looks for."""
    from dynode_amd import jit, synthetic

    def shape(**kw):
        return synthetic.seip(B=1, seed=0, t1=5.0, **kw).model

    cases = [
        (dict(A=8, L=3, K1=3, M1=4), (3, 3), 0x100 | 0x200 | 3),        # 64 lanes per tier, three tiers: one tier per wave
        (dict(A=8, L=4, K1=3, M1=4), (3, 6), 0x100 | 0x200 | 3),        # 128 lanes per tier: two waves per tier
        (dict(A=8, L=3, K1=4, M1=2), (4, 4), 0x100 | 0x200 | 4),
        (dict(A=8, L=3, K1=2, M1=3), (2, 2), 0x100 | 0x20 | 0x40 | 2),  # two tiers: a wave each
        (dict(A=8, L=4, K1=2, M1=2), (2, 4), 0x100 | 0x20 | 0x80 | 2),
        (dict(A=8, L=4, K1=1, M1=3), (1, 2), 0x100 | 0x40 | 1),         # one tier, 16 histories: two waves
        (dict(A=8, L=3, K1=1, M1=3), None, 0x100 | 1),                  # fits a wavefront: no wave group
        (dict(A=4, L=3, K1=3, M1=3), None, 0x100 | 0x20 | 3),           # half a wavefront: tier lanes inside the wave
        (dict(A=2, L=2, K1=2, M1=2), None, 0x100 | 2),
    ]
    for kw, wg, feat in cases:
        m = shape(n_knots=1, **kw)
        assert jit._seip_wave_group(m) == wg, kw
        assert jit._features(m, torch.float64) == feat, kw
    src = jit._source(shape(A=8, L=4, K1=3, M1=4, n_knots=1), torch.float32, 0, 0, 1)
    assert "launch_seip<float, 0, 8, 4, 3, 4, 3, 6>" in src


def test_observation_cache_never_scores_a_new_dataset_against_an_old_one():
    """a freed observation tensor's address is reused by the next tensor of the same shape with version 0
    again; the cache must tell the two apart (it pins the tensor it keyed) and must see in-place refills."""
    from dynode_amd.simulation import odes

    odes._OBS_CACHE.clear()
    dev = torch.device("cpu")
    seen = []
    for scale in (1.0, 3.0, 7.0, 11.0):
        data = torch.diff(torch.arange(0.0, 40.0).reshape(20, 2) ** 2 * scale, dim=0)   # same shape, freed every round
        obs, lg, shape = odes._observation_constants(data, torch.float64, dev)
        assert shape == (19, 2)
        assert torch.equal(obs, data.to(torch.float64))
        assert float(lg) == pytest.approx(float(torch.lgamma(data.double() + 1).sum()))
        seen.append(float(lg))
        del data
    assert len(set(seen)) == 4
    buf = torch.ones(5, 2)
    a = odes._observation_constants(buf, torch.float64, dev)
    assert odes._observation_constants(buf, torch.float64, dev)[0] is a[0]          # same object, same version: a hit
    buf.mul_(4.0)                                                                   # refilled in place
    b = odes._observation_constants(buf, torch.float64, dev)
    assert torch.equal(b[0], torch.full((5, 2), 4.0, dtype=torch.float64))


def test_the_stepping_loop_is_written_once():
    """One stepper for every right-hand side (the reference has one `diffeqsolve`, odes.py:133-144): the step-size controller,
    the clip to the end of the interval and the starting step are called from `csrc/stepper.hpp` (+ its two includes) and
    from nowhere else; the model families (`solve_kernel.hpp`: Solver, `seip_kernel.hpp`: Seip) implement the interface its
    header documents and carry no loop of their own."""
    import os
    import re

    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dynode_amd", "csrc")
    calls = {}
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hpp", ".inc", ".hip", ".def")):
            continue
        text = open(os.path.join(csrc, name)).read()
        text = re.sub(r"//[^\n]*", "", text)
        for fn in ("decide", "decide_ms", "clip_to_end", "initial_h1"):
            n = len(re.findall(r"Control<T>::(?:template )?" + fn + r"(?:<[^>]*>)?\(", text))   # (decide<STRICT>, initial_h1<STRICT>: test-only twins)
            if n:
                calls.setdefault(fn, {})[name] = n
    assert calls == {"decide": {"stepper.hpp": 1}, "decide_ms": {"stepper.hpp": 1}, "clip_to_end": {"stepper.hpp": 1},
                     "initial_h1": {"stepper_prologue.inc": 1}}
    hooks = ("init(", "carve(", "load_trajectory(", "begin_attempt(", "start_ok(", "weigh(", "count_once(", "begin_output(",
             "dense_begin(", "emit_row(", "fill_row(")
    for family in ("solve_kernel.hpp", "seip_kernel.hpp"):
        text = open(os.path.join(csrc, family)).read()
        for h in hooks:
            assert re.search(r"\b" + re.escape(h), text), (family, h)
        assert "__any(!done)" not in re.sub(r"//[^\n]*", "", text)      # (no "while a trajectory has steps to take" of their own)
    seip = re.sub(r"//[^\n]*", "", open(os.path.join(csrc, "seip_kernel.hpp")).read())
    assert "Stepper<Seip<" in seip and "while (__any" not in seip
