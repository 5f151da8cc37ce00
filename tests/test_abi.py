"""The C-ABI library loads and exports every symbol include/dynode_hip.h declares.

CPU-only: no compute call is made (argument validation happens before any HIP call, so the
error paths are safe to exercise without a GPU).
"""

import ctypes
import os
import re

import pytest

import helpers as H
from dynode_amd import ModelDesc, _abi


def _declared_symbols():
    text = open(os.path.join(H.ROOT, "include", "dynode_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(dyn_[a-z_0-9]+)\s*\(", text))


def test_library_exports_every_declared_symbol():
    lib = _abi.lib()
    declared = _declared_symbols()
    assert declared == set(_abi.EXPORTED_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.dyn_abi_version() == 9


def test_no_torch_types_in_the_header():
    text = open(os.path.join(H.ROOT, "include", "dynode_hip.h")).read()
    assert "torch" not in text and "at::" not in text and "#include <hip" not in text


@pytest.mark.parametrize("m", [
    ModelDesc(n_age=1), ModelDesc(n_age=8), ModelDesc(n_age=1, has_e=True, has_wane=True, seasonal=True),
    ModelDesc(n_age=2, n_strain=3, has_e=True, has_wane=True, has_c=True),
    ModelDesc(n_age=8, n_strain=4, has_e=True, has_wane=True, has_c=True, n_wane=8),
])
def test_dimension_queries_agree_with_python_mirror_and_oracle(m):
    lib = _abi.lib()
    c = m.c()
    assert lib.dyn_state_dim(ctypes.byref(c)) == m.state_dim == H.O.state_dim(H.omodel(m))
    assert lib.dyn_param_dim(ctypes.byref(c)) == m.param_dim == H.O.param_dim(H.omodel(m))
    n = lib.dyn_n_compartments(ctypes.byref(c))
    assert n == len(m.compartment_names)
    off = (ctypes.c_int32 * 8)()
    assert lib.dyn_compartment_offsets(ctypes.byref(c), off) == n
    sizes = tuple(off[i + 1] - off[i] for i in range(n))
    assert sizes == m.compartment_sizes
    assert list(off[:n + 1]) == list(H.O.compartment_offsets(H.omodel(m)))


def test_trajectories_per_wave():
    lib = _abi.lib()
    for A, want in ((1, 64), (2, 32), (3, 16), (8, 8), (9, 4), (33, 1), (64, 1)):
        assert lib.dyn_trajectories_per_wave(ctypes.byref(ModelDesc(n_age=A).c())) == want
    assert lib.dyn_trajectories_per_wave(ctypes.byref(ModelDesc(n_age=65).c())) == 0


def _opts(**kw):
    d = dict(method=0, dtype=0, rtol=1e-5, atol=1e-6, max_steps=10**6, constant_dt=0.0, jump_ts=None, n_jump=0)
    d.update(kw)
    return _abi.SolverOptsC(**d)


def test_supported_shapes_cover_the_baseline_configs():
    lib = _abi.lib()
    for m in (ModelDesc(n_age=1), ModelDesc(n_age=2), ModelDesc(n_age=8),
              ModelDesc(n_age=1, has_e=True, has_wane=True),
              ModelDesc(n_age=2, n_strain=3, has_e=True, has_wane=True, has_c=True),
              ModelDesc(n_age=8, n_strain=4, has_e=True, has_wane=True, has_c=True)):
        assert lib.dyn_is_supported(ctypes.byref(m.c()), ctypes.byref(_opts())) == 1, m
    assert lib.dyn_is_supported(ctypes.byref(ModelDesc(n_age=8, n_strain=7).c()), ctypes.byref(_opts())) == 0


def test_argument_errors_are_codes_not_crashes():
    lib = _abi.lib()
    m, o = ModelDesc(n_age=8).c(), _opts()
    one = ctypes.c_void_p(16)  # never dereferenced: validation fails first

    def call(model=m, opts=o, y0=one, B=4, t1=10.0, n_save=2, save_ts=one, out=one):
        return lib.dyn_solve_batch(ctypes.byref(model), ctypes.byref(opts), y0, 0, one, one, B, 0.0, t1, save_ts,
                                   n_save, None, out, one, one, one, None)

    assert call(y0=None) == -1                                   # DYN_ERR_NULL
    assert call(model=ModelDesc(n_age=0).c()) == -2              # DYN_ERR_MODEL
    assert call(model=ModelDesc(n_age=2, n_wane=3).c()) == -2    # W > 1 needs waning
    assert call(B=-1) == -3                                      # DYN_ERR_SIZE
    assert call(save_ts=None) == -3
    assert call(opts=_opts(method=5)) == -4                      # DYN_ERR_OPTS
    assert call(opts=_opts(rtol=0.0)) == -5                      # DYN_ERR_TOL (params.py: PositiveFloat)
    assert call(opts=_opts(max_steps=0)) == -5
    assert call(t1=-1.0) == -5
    assert call(opts=_opts(n_jump=2)) == -6                      # DYN_ERR_JUMP
    assert call(model=ModelDesc(n_age=8, n_strain=7).c()) == -7  # DYN_ERR_UNSUPPORTED
    assert b"no kernel compiled" in lib.dyn_last_error()
    assert call(B=0) == 0                                        # empty batch: nothing enqueued


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setattr(_abi, "LIB_PATH", "/nonexistent/libdynode_hip.so")
    with pytest.raises(_abi.HipLibraryMissing):
        _abi.lib()


def test_sampler_state_struct_matches_the_header_and_rng_known_answers():
    lib = _abi.lib()
    assert lib.dyn_nuts_state_size() == ctypes.sizeof(_abi.NutsStateC)
    text = open(os.path.join(H.ROOT, "include", "dynode_hip.h")).read()
    body = text[text.index("typedef struct dyn_nuts_state"):text.index("} dyn_nuts_state;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    ptrs = []
    for decl in re.findall(r"(?:const\s+)?(?:double|int32_t|int64_t)\s+(\*[^;]+);", body):
        ptrs += [n.strip().lstrip("*") for n in decl.split(",")]
    assert tuple(ptrs) == _abi.NUTS_POINTER_FIELDS + ("pot_lp", "pot_dlp", "pot_ll", "pot_dll")     # (the optional potential parts come last)
    for name, val in (("DYN_NUTS_MAX_DIM", _abi.NUTS_MAX_DIM), ("DYN_NUTS_MAX_DEPTH", _abi.NUTS_MAX_DEPTH),
                      ("DYN_NUTS_MAX_WINDOWS", _abi.NUTS_MAX_WINDOWS)):
        assert int(re.search(rf"#define {name} (\d+)", text).group(1)) == val

    # Philox4x32-10 known-answer vectors of the Random123 distribution (kat_vectors)
    def philox(ctr, key):
        c, k, o = (ctypes.c_uint32 * 4)(*ctr), (ctypes.c_uint32 * 2)(*key), (ctypes.c_uint32 * 4)()
        lib.dyn_philox4x32_10(c, k, o)
        return tuple(o)
    assert philox([0] * 4, [0] * 2) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert philox([0xffffffff] * 4, [0xffffffff] * 2) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == (
        0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_sampler_entry_validates_before_launching():
    lib = _abi.lib()
    assert lib.dyn_nuts_advance(None, None) == -1
    st = _abi.NutsStateC()
    st.n_chains, st.dim, st.max_depth = 4, 33, 5          # dimension above DYN_NUTS_MAX_DIM
    assert lib.dyn_nuts_advance(ctypes.byref(st), None) == -3
    st.dim, st.pooled = 9, 1                              # pooled windows stop at 8 dimensions: refused before any pointer is read
    st.pool = st.pool_ro = st.pend = 8                    # (non-null)
    assert lib.dyn_nuts_advance(ctypes.byref(st), None) == -7
    st.pooled = 0
    st.dim, st.max_depth = 2, 11
    assert lib.dyn_nuts_advance(ctypes.byref(st), None) == -3
    st.max_depth, st.n_chains = 5, 0                      # no chains: nothing to do, no launch
    assert lib.dyn_nuts_advance(ctypes.byref(st), None) == 0


def test_rows_per_chain_of_the_latent_map_are_validated():
    """`split_directions` of dyn_latent_param_map / dyn_potential_combine: 0 = every direction in the chain's one row, 1 = n_sites
    rows, r >= 2 = r rows with r >= n_sites (the rest padding; include/dynode_hip.h) -- anything else is a size error, found
    before any launch (C = 0: nothing runs, no GPU needed)."""
    lib = _abi.lib()
    sites = (_abi.SiteDescC * 3)()
    for sd in sites:                                       # Normal(0, 1), identity transform
        sd.dist, sd.aff_scale, sd.lo, sd.hi = 0, 1.0, -1e300, 1e300
        sd.p[0], sd.p[1] = 0.0, 1.0
    dummy = (ctypes.c_double * 16)()
    addr = ctypes.addressof(dummy)

    def rc(split, n=3):
        return lib.dyn_latent_param_map(sites, n, 0, None, None, None, None, 2, addr, addr, _abi.DYN_F32, split, None, None, None)

    assert rc(0) == 0 and rc(1) == 0 and rc(3) == 0 and rc(4) == 0 and rc(64) == 0
    assert rc(2) == -3 and rc(65) == -3 and rc(-1) == -3   # fewer rows than sites, more than a wavefront, negative
    assert lib.dyn_potential_combine(0, 3, None, None, None, None, 0.0, 2, None, None, None) == -3
    assert lib.dyn_potential_combine(0, 3, None, None, None, None, 0.0, 8, None, None, None) == 0


@pytest.mark.on_demand_build
def test_on_demand_kernel_build_and_registration():
    """dynode_amd/jit.py without a GPU: lane mapping choices, and a full build + dyn_register_instance
    round trip for a shape instances.def does not list (hipcc cross-compiles here)."""
    import torch

    from dynode_amd import jit

    assert jit.choose_spl(ModelDesc(n_age=8, n_strain=4, has_e=True, has_wane=True, has_c=True), 0) == 4
    assert jit.choose_spl(ModelDesc(n_age=8, n_strain=4, has_e=True, has_wane=True, has_c=True, n_wane=8), 0) in (1, 2)
    assert jit.choose_spl(ModelDesc(n_age=8, n_strain=4, has_e=True, has_wane=True, has_c=True), 2) < 4
    assert jit.choose_spl(ModelDesc(n_age=3, n_strain=5), 0) == 5
    with pytest.raises(RuntimeError):
        jit.choose_spl(ModelDesc(n_age=64, n_strain=30, has_wane=True, n_wane=12), 2)       # cannot be mapped to 64 lanes
    lib = _abi.lib()
    m = ModelDesc(n_age=3, n_strain=2, has_e=True, has_wane=False, has_c=False)
    assert lib.dyn_is_supported(ctypes.byref(m.c()), ctypes.byref(_opts())) == 0
    assert jit.ensure_kernel(m, torch.float32, "tsit5", 0) is True
    assert lib.dyn_is_supported(ctypes.byref(m.c()), ctypes.byref(_opts())) == 1
    assert jit.ensure_kernel(m, torch.float32, "tsit5", 0) is False                         # already there
    assert lib.dyn_register_instance(0, 0, 3, 1, 0, 0, 0, 1, 0, 1, 0, ctypes.c_void_p(1)) == -2   # ga must be a power of two
    assert lib.dyn_register_instance(0, 0, 4, 1, 0, 0, 0, 1, 0, 1, 0, None) == -1


def test_library_links_only_the_hip_runtime():
    """The boundary is a plain C-ABI shared object: no torch, no Python in its dependencies."""
    import subprocess

    deps = subprocess.run(["ldd", _abi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = [line.split()[0] for line in deps.splitlines() if line.strip()]
    assert any("amdhip64" in n for n in names)
    assert not any(bad in n for n in names for bad in ("torch", "c10", "python", "numpy"))


def test_trajectories_per_wave_follows_the_seip_lane_mapping():
    """SEIP: 8 ages x 4 histories is 32 lanes (2 trajectories per wave) with all tiers on one lane; the float kernel
    that is compiled in deals the tiers over two lanes (64 lanes, 1 per wave).  Three strains x 4 ages x 3 tiers runs one
    tier per wave with the (age, history) planes of two trajectories side by side in each of the three waves."""
    lib = _abi.lib()
    seip = lambda **kw: ModelDesc(has_e=True, has_wane=True, has_c=True, normalize=False, family=1, **kw)
    assert lib.dyn_trajectories_per_wave(ctypes.byref(seip(n_age=8, n_strain=2, n_wane=4, n_vax_tiers=3).c())) == 1
    assert lib.dyn_trajectories_per_wave(ctypes.byref(seip(n_age=4, n_strain=2, n_wane=4, n_vax_tiers=3).c())) == 4
    assert lib.dyn_trajectories_per_wave(ctypes.byref(seip(n_age=4, n_strain=3, n_wane=4, n_vax_tiers=3).c())) == 2
    assert lib.dyn_trajectories_per_wave(ctypes.byref(seip(n_age=2, n_strain=2, n_wane=2, n_vax_tiers=2).c())) == 8


def test_struct_layouts_of_the_binding_match_the_header():
    """ctypes mirrors of dyn_model_desc / dyn_solver_opts / dyn_dispatch_hints: sizes against the library's own sizeof, field
    names and order of the hints against the header text (a binding that drifts from the header would pass garbage hints)."""
    lib = _abi.lib()
    assert lib.dyn_model_desc_size() == ctypes.sizeof(_abi.ModelDescC) and lib.dyn_solver_opts_size() == ctypes.sizeof(_abi.SolverOptsC)
    text = open(os.path.join(H.ROOT, "include", "dynode_hip.h")).read()
    body = re.sub(r"/\*.*?\*/", "", text[text.index("typedef struct dyn_dispatch_hints"):text.index("} dyn_dispatch_hints;")], flags=re.S)
    assert tuple(re.findall(r"int32_t\s+(\w+);", body)) == tuple(n for n, _ in _abi.DispatchHintsC._fields_)
    opts = re.sub(r"/\*.*?\*/", "", text[text.index("typedef struct dyn_solver_opts"):text.index("} dyn_solver_opts;")], flags=re.S)
    assert tuple(re.findall(r"[\w\s\*]+?[\s\*](\w+);", opts)) == tuple(n for n, _ in _abi.SolverOptsC._fields_)


def test_dispatch_hints_context_sets_exactly_the_fields_it_names():
    from dynode_amd import engine

    o = _opts()
    with engine.dispatch_hints(pull=-1, replicas_log2=0, strains_per_lane=2, strict_control=1):
        with engine.dispatch_hints(pull=None, seip_tier_waves=-1):
            engine._apply_hints(o)
    h = o.hints
    assert (h.pull, h.replicas_log2, h.strains_per_lane, h.strict_control, h.seip_tier_waves, h.pull_waves) == (0, 1, 2, 1, -1, 0)
    assert engine.current_hints() == {}
    with pytest.raises(TypeError):
        with engine.dispatch_hints(no_such_hint=1):
            pass
    # out-of-range hints are an argument error of the call, before anything is enqueued
    o2 = _opts()
    o2.hints.replicas_log2 = 9
    m = ModelDesc(n_age=8)
    assert lib_call_rc(m, o2) == -4


def lib_call_rc(m, o):
    lib = _abi.lib()
    one = ctypes.c_void_p(8)            # non-null placeholders: the argument checks come before any pointer is read
    return lib.dyn_solve_batch(ctypes.byref(m.c()), ctypes.byref(o), one, 0, one, one, 4, 0.0, 10.0, one, 3, None, one, one, one, one, None)
