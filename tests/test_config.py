"""dynode_amd.config against the behaviours the reference's tests/test_config/*.py pin (bins,
dimensions, compartments, strains, transmission / solver params, SimulationConfig and its ``idx``,
deterministic parameters, placeholder samples, simulation days).  CPU only."""

import math
import os
import string
from dataclasses import dataclass
from datetime import date

import pytest
from pydantic import BaseModel, ValidationError

import dynode_amd.config as config
from dynode_amd.infer import distributions as dist
from dynode_amd.infer import handlers, resolve_deterministic

B, D, C = config.Bin, config.Dimension, config.Compartment
IntBin = config.DiscretizedPositiveIntBin


def strain(name, **kw):
    return config.Strain(strain_name=name, r0=2, infectious_period=5, **kw)


# ------------------------------------------------------------------ bins (test_bins.py)
def test_bins():
    b = IntBin(min_value=0, max_value=10)
    assert (b.min_value, b.max_value) == (0, 10)
    for bad in ((10, 0), (-5, 0)):
        with pytest.raises(ValidationError):
            IntBin(min_value=bad[0], max_value=bad[1])
    assert B(name="valid_name").name == "valid_name"
    for name in ["1_invalid_name", "invalid name"] + [f"invalid{c}name" for c in string.punctuation if c != "_"]:
        with pytest.raises(ValidationError):
            B(name=name)
    w = config.WaneBin(name="wane_bin", waiting_time=10.0, base_protection=0.8)
    assert (w.name, w.waiting_time, w.base_protection) == ("wane_bin", 10.0, 0.8)
    for kw in (dict(waiting_time=10.0, base_protection=1.2), dict(waiting_time=-5.0, base_protection=0.8)):
        with pytest.raises(ValidationError):
            config.WaneBin(name="invalid_wane_bin", **kw)


# ------------------------------------------------------------------ dimensions (test_dimension.py)
def test_plain_dimension_rules():
    d = D(name="valid_dimension", bins=[B(name="bin1")])
    assert d.name == "valid_dimension" and len(d.bins) == len(d) == 1 and d.bins[0].name == "bin1"
    ok = D(name="ages", bins=[IntBin(min_value=0, max_value=10), IntBin(min_value=11, max_value=20)])
    assert [b.min_value for b in ok.bins] == [0, 11]
    bad_sets = [
        [B(name="bin1"), IntBin(min_value=0, max_value=10)],                         # mixed types
        [],                                                                          # empty
        [B(name="bin1"), B(name="bin1")],                                            # duplicate names
        [IntBin(min_value=0, max_value=10), IntBin(min_value=5, max_value=15)],      # overlap
        [IntBin(min_value=11, max_value=20), IntBin(min_value=0, max_value=10)],     # unsorted
        [IntBin(min_value=0, max_value=10), IntBin(min_value=12, max_value=20)],     # gap
    ]
    for bins in bad_sets:
        with pytest.raises(ValidationError):
            D(name="d", bins=bins)


@pytest.mark.parametrize("seasonal", [False, True])
def test_vaccination_dimension(seasonal):
    d = config.VaccinationDimension(max_ordinal_vaccinations=2, seasonal_vaccination=seasonal)
    assert d.name == "vax" and len(d.bins) == 2 + 1 + int(seasonal) and d.max_shots == len(d.bins) - 1
    assert d.seasonal_vaccination is seasonal
    assert all(b.min_value == b.max_value == k and b.name == f"v{k}" for k, b in enumerate(d.bins))


def test_immune_history_dimensions():
    strains = [config.Strain(strain_name=f"s{i}", r0=2, infectious_period=2) for i in range(3)]
    full = config.FullStratifiedImmuneHistoryDimension(strains=strains)
    assert full.name == "hist" and len(full.bins) == 2 ** 3
    assert [b.name for b in full.bins] == ["none", "s0", "s1", "s2", "s0_s1", "s0_s2", "s1_s2", "s0_s1_s2"]
    with pytest.raises(ValidationError):                                            # duplicate strain names
        config.FullStratifiedImmuneHistoryDimension(strains=[strain("same_strain"), strain("same_strain")])
    last = config.LastStrainImmuneHistoryDimension(strains=strains)
    assert last.name == "hist" and [b.name for b in last.bins] == ["none", "s0", "s1", "s2"]
    assert isinstance(full, config.ImmuneHistoryDimension) and isinstance(last, config.ImmuneHistoryDimension)


def test_wane_dimension():
    times, prot = [1.0, 2.0, 3.0, math.inf], [0.5, 0.6, 0.7, 0.9]
    d = config.WaneDimension(waiting_times=times, base_protections=prot)
    assert d.name == "wane" and len(d.bins) == 4
    assert all((b.waiting_time, b.base_protection, b.name) == (times[i], prot[i], f"W{i}") for i, b in enumerate(d.bins))
    with pytest.raises(ValidationError):
        config.WaneDimension(waiting_times=[1.0, 2.0, 3.0, 4.0], base_protections=prot)   # last stage must be absorbing


# ------------------------------------------------------------------ compartments (test_compartment.py)
def test_compartments():
    c = C(name="valid_compartment", dimensions=[D(name="dim1", bins=[B(name="bin1")])])
    assert c.name == "valid_compartment" and c.shape == (1,) and c.idx.dim1 == 0 and c.idx.dim1.bin1 == 0
    same = lambda n, dn, bn: C(name=n, dimensions=[D(name=dn, bins=[B(name=bn)])])
    assert same("compartment1", "dim1", "bin1") == same("compartment1", "dim1", "bin1")
    assert same("compartment1", "dim1", "bin1") != same("compartment2", "dim1", "bin1")
    assert same("compartment1", "dim1", "bin1") != same("compartment1", "dim2", "bin2")
    with pytest.raises(ValidationError):
        C(name="invalid_compartment", dimensions=[D(name="dim1", bins=[B(name="bin1")]), D(name="dim1", bins=[B(name="bin2")])])


# ------------------------------------------------------------------ strains and params (test_strain.py, test_params.py)
def test_strains():
    s = strain("valid_strain")
    assert (s.strain_name, s.r0, s.infectious_period) == ("valid_strain", 2, 5)
    s = config.Strain(strain_name="valid_strain_dist", r0=dist.Uniform(1.0, 3.0), infectious_period=5)
    assert isinstance(s.r0, dist.Distribution) and s.infectious_period == 5
    s = strain("introduced_strain", is_introduced=True, introduction_time=100, introduction_percentage=0.1,
               introduction_scale=10, introduction_ages=[config.AgeBin(min_value=0, max_value=10)])
    assert s.is_introduced is True and (s.introduction_time, s.introduction_percentage, s.introduction_scale) == (100, 0.1, 10)
    assert s.introduction_ages == [config.AgeBin(min_value=0, max_value=10)]


def test_transmission_params():
    strains = [strain("strain1"), strain("strain2")]
    full = {"strain1": {"strain1": 1.0, "strain2": 0.5}, "strain2": {"strain1": 0.5, "strain2": 1.0}}
    tp = config.TransmissionParams(strains=strains, strain_interactions=full)
    assert tp.strains == strains and tp.strain_interactions == full
    three = {a: {b: 1.0 if a == b else 0.5 for b in ("strain1", "strain2", "strain3")} for a in ("strain1", "strain2", "strain3")}
    for bad in ({"strain1": full["strain1"], "strain2": {"strain1": 0.5}}, {"strain1": full["strain1"]}, three):
        with pytest.raises(ValidationError):
            config.TransmissionParams(strains=strains, strain_interactions=bad)
    for bad_strains in ([], [strain("strain1")], [strain("strain1"), strain("NOTstrain2")]):
        with pytest.raises(ValidationError):
            config.TransmissionParams(strains=bad_strains, strain_interactions=full)
    # optional per-strain fields must be given for every strain or for none
    for extra in (dict(exposed_to_infectious=5.0), dict(vaccine_efficacy={0: 0.8, 1: 0.9})):
        with pytest.raises(ValidationError):
            config.TransmissionParams(strains=[strain("strain1", **extra), strain("strain2")], strain_interactions=full)


def test_solver_params():
    sp = config.SolverParams(max_steps=1000, ode_solver_rel_tolerance=1e-6, ode_solver_abs_tolerance=1e-9)
    assert (sp.max_steps, sp.ode_solver_rel_tolerance, sp.ode_solver_abs_tolerance) == (1000, 1e-6, 1e-9)
    for kw in (dict(max_steps=-1000), dict(ode_solver_rel_tolerance=-1e-6)):
        with pytest.raises(ValidationError):
            config.SolverParams(**{**dict(max_steps=1000, ode_solver_rel_tolerance=1e-6, ode_solver_abs_tolerance=1e-9), **kw})


# ------------------------------------------------------------------ SimulationConfig (test_simulation_config.py)
@pytest.fixture
def cfg():
    return config.SimulationConfig(
        compartments=[C(name="compartment1", dimensions=[D(name="dim1", bins=[B(name="bin1")])])],
        initializer=config.Initializer(description="test initializer", initialize_date=date(2022, 2, 11), population_size=1000),
        parameters=config.Params(
            transmission_params=config.TransmissionParams(strains=[strain("strain1")], strain_interactions={"strain1": {"strain1": 1.0}}),
            solver_params=config.SolverParams()))


def extra(name, dim, bin_name):
    return C(name=name, dimensions=[D(name=dim, bins=[B(name=bin_name)])])


def test_simulation_config_index_and_flattening(cfg):
    assert len(cfg.compartments) == 1
    assert cfg.idx.compartment1 == 0 and cfg.idx.compartment1.dim1 == 0 and cfg.idx.compartment1.dim1.bin1 == 0
    assert cfg.parameters.transmission_params.strains[0].strain_name == "strain1"
    assert [b.name for b in cfg.flatten_bins()] == ["bin1"] and [d.name for d in cfg.flatten_dims()] == ["dim1"]
    cfg.compartments.append(extra("compartment2", "dim2", "bin2"))
    assert len(cfg.flatten_bins()) == 2 and len(cfg.flatten_dims()) == 2
    assert cfg.get_compartment("compartment1").name == "compartment1"
    with pytest.raises(AssertionError):
        cfg.get_compartment("non_existent_compartment")


def test_simulation_config_cross_compartment_validation(cfg):
    cfg.compartments.append(extra("compartment2", "dim1", "bin1"))           # same dimension, same bins: fine
    cfg.model_validate(cfg)
    cfg.compartments[1] = extra("compartment2", "dim1", "bin2")              # same dimension name, other bins
    with pytest.raises(ValidationError):
        cfg.model_validate(cfg)
    cfg.compartments[1] = extra("compartment1", "dim2", "bin2")              # duplicate compartment name
    with pytest.raises(ValidationError):
        cfg.model_validate(cfg)


def test_immune_history_must_come_from_the_models_strains(cfg):
    hist = lambda s: C(name="compartment2", dimensions=[config.LastStrainImmuneHistoryDimension(strains=[strain(s)])])
    cfg.compartments.append(hist("strain1"))
    cfg.model_validate(cfg)
    cfg.compartments[1] = hist("strain2")
    with pytest.raises(ValidationError):
        cfg.model_validate(cfg)


# ------------------------------------------------------------------ deterministic parameters
def test_deterministic_parameters():
    state = {"base_param": [1, 2, 3], "another_param": 5,
             "dependent_param": config.DeterministicParameter(depends_on="base_param", index=1),
             "dependent_param2": config.DeterministicParameter(depends_on="another_param")}
    assert state["dependent_param"].resolve(state) == 2 and state["dependent_param2"].resolve(state) == 5
    resolved = resolve_deterministic(dict(state), root_params=state)
    assert resolved["dependent_param"] == 2 and resolved["dependent_param2"] == 5
    for bad in (config.DeterministicParameter(depends_on="base_param", index=3),
                config.DeterministicParameter(depends_on="base_param", index=(1, 2)),
                config.DeterministicParameter(depends_on="missing_param")):
        with pytest.raises(Exception):
            bad.resolve(state)


# ------------------------------------------------------------------ placeholder samples
def test_placeholder_sample():
    draw = lambda: handlers.sample("sample", config.PlaceholderSample())
    config.PlaceholderSample()
    with pytest.raises(config.SamplePlaceholderError):
        with handlers.seed(0):
            draw()
    with handlers.substitute({"sample": 42}):
        assert float(draw()) == 42
    with pytest.raises(config.SamplePlaceholderError):
        with handlers.seed(0), handlers.substitute({"NOTsample": 42}):
            draw()


# ------------------------------------------------------------------ simulation days
@pytest.fixture
def no_init_date():
    key = f"DYNODE_INITIALIZATION_DATE({os.getpid()})"
    os.environ.pop(key, None)
    yield
    os.environ.pop(key, None)


def test_simulation_day(no_init_date):
    with pytest.raises(ValueError):
        config.simulation_day(2022, 2, 11)
    assert config.get_dynode_init_date_flag() is None
    config.set_dynode_init_date_flag(date(2022, 2, 11))
    assert config.get_dynode_init_date_flag() == date(2022, 2, 11)
    assert config.simulation_day(2022, 2, 11) == 0
    config.set_dynode_init_date_flag(date(2022, 2, 1))
    assert config.simulation_day(2022, 2, 11) == 10 and config.simulation_day(2022, 1, 31) == -1


def test_simulation_days_are_plain_ints_wherever_they_are_used(no_init_date):
    config.set_dynode_init_date_flag(date(2022, 2, 11))
    day = config.simulation_day

    class Model(BaseModel):
        simulation_day: int

    @dataclass(frozen=True)
    class Frozen:
        x: int

    assert {"d": day(2022, 2, 11)}["d"] == 0 and [day(2022, 2, 11), "x"][0] == 0
    assert Model(simulation_day=day(2022, 2, 11)).simulation_day == 0
    assert isinstance(Frozen(day(2022, 2, 11)).x, int) and Frozen(day(2022, 2, 11)).x == 0
    n = dist.Normal(loc=day(2022, 2, 11), scale=1.0)
    assert float(n.loc) == 0
    tn = dist.TruncatedNormal(loc=day(2022, 2, 11), scale=1.0, low=day(2022, 2, 10), high=day(2022, 2, 12))
    assert float(tn.loc) == 0 and tn.low == -1 and tn.high == 1
    import torch
    gen = torch.Generator().manual_seed(0)
    assert -1 <= float(tn.sample(gen)) <= 1 and math.isfinite(float(n.sample(gen)))


def test_introduced_strains_are_encoded_over_the_models_age_axis():
    """reference simulation_config.py:208-264: introduction_ages -> 0/1 mask over the age bins; ages
    that are not bins of the model are refused."""
    ages = [config.AgeBin(min_value=0, max_value=17), config.AgeBin(min_value=18, max_value=64), config.AgeBin(min_value=65, max_value=99)]
    intro = dict(is_introduced=True, introduction_time=100, introduction_percentage=0.1, introduction_scale=10)

    def build(strains):
        names = [s.strain_name for s in strains]
        return config.SimulationConfig(
            compartments=[C(name="s", dimensions=[D(name="age", bins=ages)])],
            initializer=config.Initializer(description="x", initialize_date=date(2022, 2, 11), population_size=1000),
            parameters=config.Params(
                transmission_params=config.TransmissionParams(strains=strains, strain_interactions={a: {b: 1.0 for b in names} for a in names}),
                solver_params=config.SolverParams()))

    cfg = build([strain("resident"), strain("newcomer", introduction_ages=[ages[1], ages[2]], **intro)])
    masks = [s.introduction_ages_mask_vector for s in cfg.parameters.transmission_params.strains]
    assert masks == [[0, 0, 0], [0, 1, 1]]
    assert build([strain("resident")]).parameters.transmission_params.strains[0].introduction_ages_mask_vector is None
    with pytest.raises(ValidationError):
        build([strain("newcomer", introduction_ages=[config.AgeBin(min_value=0, max_value=4)], **intro)])
