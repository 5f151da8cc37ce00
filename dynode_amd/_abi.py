"""ctypes binding of ``libdynode_hip.so`` (the C-ABI declared in include/dynode_hip.h).

This is the only door from Python into the HIP kernels.  There is no CPU fallback: if the
shared library is missing, or a shape is not compiled in, the call raises.
"""

from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass

_HERE = os.path.dirname(os.path.abspath(__file__))
# DYNODE_HIP_LIB overrides the library path (diagnostic builds only)
LIB_PATH = os.environ.get("DYNODE_HIP_LIB") or os.path.join(_HERE, "lib", "libdynode_hip.so")

DYN_TSIT5, DYN_DOPRI5 = 0, 1
DYN_F32, DYN_F64 = 0, 1
STATUS_OK, STATUS_MAX_STEPS, STATUS_NONFINITE = 0, 1, 2

ERR_NAMES = {
    -1: "DYN_ERR_NULL", -2: "DYN_ERR_MODEL", -3: "DYN_ERR_SIZE", -4: "DYN_ERR_OPTS",
    -5: "DYN_ERR_TOL", -6: "DYN_ERR_JUMP", -7: "DYN_ERR_UNSUPPORTED", -8: "DYN_ERR_LAUNCH",
}

# every symbol include/dynode_hip.h declares (checked by tests/test_abi.py)
def kernel_source_hash() -> str:
    """sha1 (12 hex digits) over the DEVICE code the solve kernels are built from -- the kernel headers of dynode_amd/csrc
    (solve_kernel.hpp, stepper.hpp and its includes, seip_kernel.hpp, nuts_device.hpp, latent_device.hpp), the instance lists
    and the Makefile (per-unit compiler flags).  Host-side dispatch (dynode_hip.hip) is not in it: which instance ran is
    checked by name.  Computable wherever the tree is (the GPU box has no .git): a profile records it (tools/summarize_prof.py
    -> profiles/traffic.json) and bench.py attaches profiled HBM traffic to its line only when the device code it runs is the
    device code that was profiled."""
    import hashlib

    csrc = os.path.join(_HERE, "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hpp", ".inc", ".def")) or f == "Makefile")
    h = hashlib.sha1()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


EXPORTED_SYMBOLS = (
    "dyn_abi_version", "dyn_model_desc_size", "dyn_solver_opts_size", "dyn_state_dim", "dyn_param_dim", "dyn_n_compartments",
    "dyn_compartment_offsets", "dyn_is_supported", "dyn_trajectories_per_wave", "dyn_trajectories_per_wave_for_batch",
    "dyn_last_error", "dyn_solve_batch", "dyn_solve_batch_jvp", "dyn_is_supported_jvp",
    "dyn_nuts_advance", "dyn_nuts_state_size", "dyn_philox4x32_10", "dyn_latent_sites",
    "dyn_solve_batch_loglik", "dyn_register_instance", "dyn_last_kernel_name", "dyn_solve_batch_record",
    "dyn_solve_batch_replay", "dyn_latent_param_map", "dyn_potential_combine", "dyn_solve_batch_ordered",
    "dyn_nuts_advance_mapped", "dyn_nuts_tail_size", "dyn_nuts_tail_pack", "dyn_fused_twin", "dyn_lean_twin",
)

MAX_SITES = 16          # sites of a folded potential / of the fused prior kernel (include/dynode_hip.h DYN_MAX_SITES)
DIST_NORMAL, DIST_UNIFORM, DIST_BETA, DIST_TRUNCNORMAL = 0, 1, 2, 3

NUTS_MAX_DIM, NUTS_MAX_DEPTH, NUTS_MAX_WINDOWS = 32, 10, 16
NUTS_REG_DIM = 8        # up to here: pooled windows, the one-launch sampler iteration (csrc/nuts_device.hpp kRegDim)
# pointer members of dyn_nuts_state, in declaration order (include/dynode_hip.h)
NUTS_POINTER_FIELDS = (
    "z_eval", "u_new", "g_new", "z", "u", "g", "eps", "eps_avg", "da_mu", "da_xbar", "da_gbar", "da_t",
    "imm", "mm_sqrt", "wf_n", "wf_mean", "wf_m2", "e0", "zl", "rl", "gl", "zr", "rr", "gr", "zp", "up", "gp",
    "weight", "r_sum", "sum_acc", "sgn", "zc", "rc", "gc", "r_half", "s_zp", "s_up", "s_gp", "s_weight",
    "s_rsum", "s_acc", "r_ck", "rs_ck", "it", "wi", "n_prop", "depth", "right", "leaf", "s_turn", "s_div",
    "s_n", "rng_ctr", "pool", "pool_ro", "pend", "out_z", "out_acc", "out_n", "out_div",
)
NUTS_INT32_FIELDS = ("it", "wi", "n_prop", "depth", "right", "leaf", "s_turn", "s_div", "s_n", "out_n", "out_div", "pend")
NUTS_INT64_FIELDS = ("rng_ctr", "pool", "pool_ro")


MAX_STRAINS = 8


class ModelDescC(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "n_age", "n_strain", "has_e", "has_wane", "has_c", "n_wane", "normalize", "seasonal", "has_intro",
        "n_vax_tiers")] + [("intro_age_mask", ctypes.c_uint64 * MAX_STRAINS), ("n_vax_knots", ctypes.c_int32),
                           ("family", ctypes.c_int32), ("seasonal_vax", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class DispatchHintsC(ctypes.Structure):
    """dyn_dispatch_hints (ABI 9): all zero = the library's measured choices.  ``engine.dispatch_hints`` fills it."""
    _fields_ = [(n, ctypes.c_int32) for n in ("pull", "pull_waves", "strains_per_lane", "replicas_log2", "producer_consumer",
                                              "general_instance", "seip_tier_lanes", "seip_tier_waves", "strict_control")]


class SolverOptsC(ctypes.Structure):
    _fields_ = [
        ("method", ctypes.c_int32),
        ("dtype", ctypes.c_int32),
        ("rtol", ctypes.c_double),
        ("atol", ctypes.c_double),
        ("max_steps", ctypes.c_int64),
        ("constant_dt", ctypes.c_double),
        ("jump_ts", ctypes.POINTER(ctypes.c_double)),
        ("n_jump", ctypes.c_int32),
        ("work_counter", ctypes.c_void_p),     # ABI 7: two zeroed int32 words on the device, or None (engine.work_counter)
        ("nuts_tail", ctypes.c_void_p),        # ABI 8: host blob of dyn_nuts_tail_pack, or None (infer/folded.py)
        ("hints", DispatchHintsC),             # ABI 9: dispatch overrides of tests / tuning tools (engine.dispatch_hints)
    ]


class SiteDescC(ctypes.Structure):
    _fields_ = [("dist", ctypes.c_int32), ("reserved", ctypes.c_int32), ("p", ctypes.c_double * 4),
                ("base_lo", ctypes.c_double), ("base_hi", ctypes.c_double), ("aff_loc", ctypes.c_double),
                ("aff_scale", ctypes.c_double), ("lo", ctypes.c_double), ("hi", ctypes.c_double)]


class NutsStateC(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int32) for n in ("n_chains", "dim", "max_depth", "num_warmup", "num_samples", "n_windows", "pooled")]
                + [("w_start", ctypes.c_int32 * NUTS_MAX_WINDOWS), ("w_end", ctypes.c_int32 * NUTS_MAX_WINDOWS),
                   ("seed", ctypes.c_uint64), ("target_accept", ctypes.c_double), ("max_delta_energy", ctypes.c_double)]
                + [(n, ctypes.c_void_p) for n in NUTS_POINTER_FIELDS]
                + [(n, ctypes.c_void_p) for n in ("pot_lp", "pot_dlp", "pot_ll", "pot_dll")]
                + [("pot_offset", ctypes.c_double), ("pot_ll_stride", ctypes.c_int32), ("pot_dll_stride", ctypes.c_int32)])


@dataclass(frozen=True)
class ModelDesc:
    """Python view of ``dyn_model_desc``: one member of the compartmental RHS family."""

    n_age: int = 1
    n_strain: int = 1
    has_e: bool = False
    has_wane: bool = False
    has_c: bool = False
    n_wane: int = 1
    normalize: bool = True
    seasonal: bool = False
    has_intro: bool = False          # externally introduced strains (Strain.is_introduced)
    intro_age_mask: tuple = ()       # per strain: bit a set = age bin a receives the introductions
    n_vax_tiers: int = 0             # > 1: n_age enumerates (age, vaccination tier) groups, 2 or 4 slots per age
    n_vax_knots: int = 0             # knots of the vaccination-rate splines (0..4)
    family: int = 0                  # 1 = SEIP with immune histories (include/dynode_hip.h, "SEIP")
    seasonal_vax: bool = False       # SEIP: yearly reset of the top vaccination tier

    def c(self) -> ModelDescC:
        masks = tuple(int(v) for v in self.intro_age_mask) + (0,) * (MAX_STRAINS - len(self.intro_age_mask))
        return ModelDescC(self.n_age, self.n_strain, int(self.has_e), int(self.has_wane),
                          int(self.has_c), self.n_wane, int(self.normalize), int(self.seasonal),
                          int(self.has_intro), int(self.n_vax_tiers), (ctypes.c_uint64 * MAX_STRAINS)(*masks),
                          int(self.n_vax_knots), int(self.family), int(self.seasonal_vax), 0)

    # pure-Python mirrors of dyn_state_dim & co (host logic must not need the .so)
    @property
    def seip_dims(self) -> tuple:
        """(A, L, H, K1, M1, n_knots) of a family-1 model."""
        return self.n_age, self.n_strain, 1 << self.n_strain, max(int(self.n_vax_tiers), 1), self.n_wane, self.n_vax_knots

    @property
    def compartment_names(self) -> tuple:
        if self.family == 1:
            return ("s", "e", "i", "c")
        return ("s",) + (("e",) if self.has_e else ()) + ("i", "r") + (("c",) if self.has_c else ())

    @property
    def compartment_sizes(self) -> tuple:
        if self.family == 1:
            A, L, H, K1, M1, _ = self.seip_dims
            return (A * H * K1 * M1,) + (A * H * K1 * L,) * 3
        A, AS = self.n_age, self.n_age * self.n_strain
        return (A,) + ((AS,) if self.has_e else ()) + (AS, AS * self.n_wane) + (
            (AS,) if self.has_c else ())

    @property
    def state_dim(self) -> int:
        return sum(self.compartment_sizes)

    @property
    def param_dim(self) -> int:
        if self.family == 1:
            A, L, H, K1, M1, nk = self.seip_dims
            return (3 * L + M1 + (3 * L if self.has_intro else 0) + (3 if self.seasonal else 0) + int(self.seasonal_vax) + A
                    + H * K1 * M1 * L + A * K1 * (4 + 2 * nk))
        vax = self.n_age * (self.n_strain + 4 + 2 * self.n_vax_knots) if self.n_vax_tiers > 1 else 0
        return self.n_strain * (2 + int(self.has_e) + int(self.has_wane) + (3 if self.has_intro else 0)) + (
            3 if self.seasonal else 0) + vax

    @property
    def vax_lanes(self) -> int:
        """Tier slots per age on the contact axis: 0 (no vaccination axis), 2 or 4."""
        return 0 if (self.n_vax_tiers <= 1 or self.family == 1) else (2 if self.n_vax_tiers <= 2 else 4)


class HipLibraryMissing(RuntimeError):
    pass


_lib = None


def lib() -> ctypes.CDLL:
    """Load libdynode_hip.so; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C dynode_amd/csrc`. dynode_amd has no CPU fallback."
            )
        L = ctypes.CDLL(LIB_PATH)
        pm, po = ctypes.POINTER(ModelDescC), ctypes.POINTER(SolverOptsC)
        L.dyn_abi_version.restype = ctypes.c_int32
        for name, struct in (("dyn_model_desc_size", ModelDescC), ("dyn_solver_opts_size", SolverOptsC)):
            getattr(L, name).restype = ctypes.c_int32
            if getattr(L, name)() != ctypes.sizeof(struct):      # a stale library next to newer Python (or the reverse): refuse, loudly
                raise ImportError(f"{LIB_PATH}: {name}() = {getattr(L, name)()} but the binding's struct has {ctypes.sizeof(struct)} bytes "
                                  "(rebuild: make -C dynode_amd/csrc)")
        for name in ("dyn_state_dim", "dyn_param_dim", "dyn_n_compartments",
                     "dyn_trajectories_per_wave"):
            getattr(L, name).argtypes = [pm]
            getattr(L, name).restype = ctypes.c_int32
        L.dyn_trajectories_per_wave_for_batch.argtypes = [pm, po, ctypes.c_int64]
        L.dyn_trajectories_per_wave_for_batch.restype = ctypes.c_int32
        L.dyn_compartment_offsets.argtypes = [pm, ctypes.c_void_p]
        L.dyn_compartment_offsets.restype = ctypes.c_int32
        L.dyn_is_supported.argtypes = [pm, po]
        L.dyn_is_supported.restype = ctypes.c_int32
        L.dyn_last_error.restype = ctypes.c_char_p
        L.dyn_last_kernel_name.restype = ctypes.c_char_p
        L.dyn_last_kernel_name.argtypes = []
        L.dyn_solve_batch.restype = ctypes.c_int
        L.dyn_solve_batch.argtypes = [
            pm, po, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_int32,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_void_p,
        ]
        base = [pm, po, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_double,
                ctypes.c_double, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                ctypes.c_void_p, ctypes.c_void_p]
        L.dyn_solve_batch_record.restype = ctypes.c_int
        L.dyn_solve_batch_record.argtypes = base + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
        L.dyn_solve_batch_replay.restype = ctypes.c_int
        L.dyn_solve_batch_replay.argtypes = base + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
        L.dyn_solve_batch_ordered.restype = ctypes.c_int
        L.dyn_solve_batch_ordered.argtypes = base + [ctypes.c_void_p, ctypes.c_void_p]
        L.dyn_nuts_advance_mapped.restype = ctypes.c_int
        L.dyn_nuts_advance_mapped.argtypes = ([ctypes.POINTER(NutsStateC), ctypes.POINTER(SiteDescC), ctypes.c_int32, ctypes.c_int32,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32] + [ctypes.c_void_p] * 6)
        L.dyn_nuts_tail_size.restype = ctypes.c_int32
        L.dyn_nuts_tail_size.argtypes = []
        L.dyn_nuts_tail_pack.restype = ctypes.c_int
        L.dyn_nuts_tail_pack.argtypes = ([ctypes.POINTER(NutsStateC), ctypes.POINTER(SiteDescC), ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32] + [ctypes.c_void_p] * 6)
        L.dyn_is_supported_jvp.argtypes = [pm, po, ctypes.c_int32]
        L.dyn_is_supported_jvp.restype = ctypes.c_int32
        L.dyn_fused_twin.argtypes = [pm, po, ctypes.c_int32]
        L.dyn_fused_twin.restype = ctypes.c_int32
        L.dyn_lean_twin.argtypes = [pm, po, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
        L.dyn_lean_twin.restype = ctypes.c_int32
        L.dyn_solve_batch_jvp.restype = ctypes.c_int
        L.dyn_solve_batch_jvp.argtypes = [
            pm, po, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_int32,
            ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_void_p,
        ]
        L.dyn_solve_batch_loglik.restype = ctypes.c_int
        L.dyn_solve_batch_loglik.argtypes = [
            pm, po, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
            ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
            ctypes.c_double, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ]
        L.dyn_register_instance.restype = ctypes.c_int
        L.dyn_register_instance.argtypes = [ctypes.c_int32] * 11 + [ctypes.c_void_p]
        L.dyn_nuts_advance.restype = ctypes.c_int
        L.dyn_nuts_advance.argtypes = [ctypes.POINTER(NutsStateC), ctypes.c_void_p]
        L.dyn_nuts_state_size.restype = ctypes.c_int32
        L.dyn_philox4x32_10.restype = None
        L.dyn_philox4x32_10.argtypes = [ctypes.POINTER(ctypes.c_uint32)] * 3
        L.dyn_latent_sites.restype = ctypes.c_int
        L.dyn_latent_sites.argtypes = [ctypes.POINTER(SiteDescC), ctypes.c_int32, ctypes.c_int64] + [ctypes.c_void_p] * 6
        L.dyn_latent_param_map.restype = ctypes.c_int
        L.dyn_latent_param_map.argtypes = ([ctypes.POINTER(SiteDescC), ctypes.c_int32, ctypes.c_int64] + [ctypes.c_void_p] * 4
                                           + [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
                                           + [ctypes.c_void_p] * 3)
        L.dyn_potential_combine.restype = ctypes.c_int
        L.dyn_potential_combine.argtypes = ([ctypes.c_int64, ctypes.c_int32] + [ctypes.c_void_p] * 4 + [ctypes.c_double, ctypes.c_int32]
                                            + [ctypes.c_void_p] * 3)
        _lib = L
    return _lib
