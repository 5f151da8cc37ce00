"""Kernel shapes on demand: compile one explicit instantiation of the solver template with hipcc and
register it with the library (``dyn_register_instance``).

``libdynode_hip.so`` ships the shapes of ``csrc/instances.def`` (the reference's examples, the
BASELINE configurations, their tangent kernels).  Any other member of the RHS family -- 5 ages x 2
strains, an SEIR without waning, a 12-stage waning chain, ... -- is the same template with different
constants; the first call with such a model builds ``dynode_amd/lib/jit/<shape>.so`` (10-20 s, kept
on disk) and later calls and processes load it directly.  This is template instantiation, not
tracing: the kernel source is csrc/solve_kernel.hpp, unchanged.

Disable with ``DYNODE_HIP_JIT=0`` (the call then fails with the library's "no kernel compiled for
..." message, which also names the line to add to instances.def for a permanent build).
"""

from __future__ import annotations

import ctypes
import hashlib
import os
import sys
import subprocess
import threading

import torch

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_OUT = os.path.join(_HERE, "lib", "jit")
_LOCK = threading.Lock()
_LOADED: dict = {}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def enabled() -> bool:
    return os.environ.get("DYNODE_HIP_JIT", "1") != "0"


def _group_width(n_age: int) -> int:
    g = 1
    while g < n_age:
        g <<= 1
    return g


def _seip_tier_lanes(model: _abi.ModelDesc, dtype=torch.float32) -> bool:
    """SEIP states of more than 32 values per lane would spill, and float states of at most 20 values per half fit two
    waves per SIMD: deal the tiers over two lanes when the group still fits a wave (csrc/seip_kernel.hpp, KT = 2; the
    same rule as solve_impl in csrc/dynode_hip.hip)."""
    A, L, H, K1, M1, _ = model.seip_dims
    per_tier = M1 + 3 * L
    small = dtype == torch.float32 and ((K1 + 1) // 2) * per_tier <= 20 and _group_width(A) * H >= 32
    return (K1 * per_tier > 32 or small) and K1 > 1 and _group_width(A) * H * 2 <= 64


def _seip_wave_group(model: _abi.ModelDesc):
    """(KT, NW) of the wave-group mapping a SEIP shape needs when its (age, history) plane fills a wavefront or more
    (csrc/seip_kernel.hpp: `Seip<..., KT, NW>`; the same choices as `select_seip_entry` in csrc/dynode_hip.hip), else None:
    one tier per wave from three tiers on, the two tiers on a wave each, a single tier's 128 lanes as two waves."""
    A, L, H, K1, M1, _ = model.seip_dims
    lanes = _group_width(A) * H
    if lanes < 64 or (lanes == 64 and K1 == 1):
        return None
    if lanes > 128:
        raise RuntimeError(f"{model}: {lanes} (age, immune history) lanes per tier; the SEIP wave groups go up to 128")
    nxh = lanes // 64
    return (K1, K1 * nxh) if K1 >= 2 else (1, nxh)


def _features(model: _abi.ModelDesc, dtype=torch.float32) -> int:
    """The kernel template's FEAT word: bit 0 = externally introduced strains, the rest = vaccination-tier lanes."""
    if model.family == 1:                                    # kSeip | tiers [| lane mapping] (csrc/dynode_hip.hip)
        feat = 0x100 | max(int(model.n_vax_tiers), 1)
        wg = _seip_wave_group(model)
        if wg is None:
            return feat | (0x20 if _seip_tier_lanes(model, dtype) else 0)
        kt, nw = wg
        if kt > 2:
            return feat | 0x200                              # kSeipTierWaves: one tier per wave
        return feat | (0x20 if kt == 2 else 0) | (0x40 if nw == 2 else 0x80)
    return int(model.has_intro) | (model.vax_lanes << 1)


def choose_spl(model: _abi.ModelDesc, n_dir: int) -> int:
    """Strains per lane: all of them on one lane per age while the register file holds it (y, 7 stage
    derivatives and the stage state of every plane: 9 * values * planes VGPRs), otherwise split the
    strains over a power-of-two number of lanes (csrc/solve_kernel.hpp, "Strain lanes")."""
    if model.family == 1:
        return 1                                             # SEIP: lanes are (age, immune history) pairs
    ga, S = _group_width(model.n_age), model.n_strain
    per_strain = int(model.has_e) + 1 + model.n_wane + int(model.has_c)
    best = None
    for spl in range(S, 0, -1):
        gs = S // spl
        if S % spl or gs & (gs - 1) or ga * gs > 64:
            continue
        best = spl
        if 9 * (1 + spl * per_strain) * (1 + n_dir) <= 230:
            return spl
    if best is None:
        raise RuntimeError(f"no lane mapping for {model}: {S} strains cannot be split over a power-of-two number of lanes")
    if 9 * (1 + best * per_strain) * (1 + n_dir) > 480:
        raise RuntimeError(f"{model} with {n_dir} tangent directions needs about {9 * (1 + best * per_strain) * (1 + n_dir)} "
                           "vector registers per lane: too large for the register-resident kernel (512 per lane)")
    return best


_STAMP = None


def _stamp() -> str:
    """Fingerprint of the kernel sources: a cached build made from other sources (different argument struct,
    different arithmetic) must never be loaded, so it is part of every cache file's name."""
    global _STAMP
    if _STAMP is None:
        h = hashlib.sha1()
        files = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if f.endswith((".hpp", ".h", ".inc"))]
        files.append(os.path.join(os.path.dirname(os.path.dirname(_CSRC)), "include", "dynode_hip.h"))   # the argument structs
        for path in files:
            if os.path.exists(path):
                with open(path, "rb") as fh:
                    h.update(fh.read())
        try:   # another compiler, another code object
            h.update(subprocess.run([HIPCC, "--version"], capture_output=True, timeout=60).stdout)
        except (OSError, subprocess.SubprocessError):
            pass
        _STAMP = h.hexdigest()[:10]
    return _STAMP


_SEIP_PLAIN = 0x1000     # csrc/dynode_hip.hip kSeipPlain


def _seip_plain(model, dtype) -> bool:
    """An on-demand SEIP build carries the plain instance too when calls on this model can reach it: float32 (where it was
    measured: 5-20 % on the built-in shapes) and a model without seasonal forcing, seasonal vaccination or introduced strains, with dose splines of at most two knots."""
    return (model.family == 1 and dtype == torch.float32 and not (model.seasonal or model.seasonal_vax or model.has_intro)
            and model.n_vax_knots <= 2)


_FUSED = 0x1000          # csrc/dynode_hip.hip kFused


def _fused_twin(model, dtype, n_dir) -> bool:
    """An on-demand tangent build of the s/e/i/r/c family carries the fused-sampler twin too (float32, the sampler's dtype for the
    solve; one or two directions per trajectory; no vaccination-tier lanes: solve_kernel.hpp FUSED)."""
    return model.family == 0 and dtype == torch.float32 and n_dir in (1, 2) and model.vax_lanes == 0


_STATIC_ADAPTIVE = 0x0C00          # csrc/dynode_hip.hip kStaticOnly | kAdaptiveNoJumps


def _static_twin(model, dtype, n_dir) -> bool:
    """An on-demand float32 tangent build carries the static-grid / adaptive-only twin too (10 % on the 2-age x 3-strain shape)."""
    return model.family == 0 and dtype == torch.float32 and n_dir > 0


def _name(model, dtype, method, n_dir, spl) -> str:
    return (f"{'f64' if dtype == torch.float64 else 'f32'}_m{method}_g{_group_width(model.n_age)}_s{model.n_strain}"
            f"_e{int(model.has_e)}w{int(model.has_wane)}c{int(model.has_c)}_W{model.n_wane}_nd{n_dir}_spl{spl}"
            f"_f{_features(model, dtype)}{'p' if _seip_plain(model, dtype) else ''}{'t' if _fused_twin(model, dtype, n_dir) else ''}{'s' if _static_twin(model, dtype, n_dir) else ''}_{_stamp()}")


def _source(model, dtype, method, n_dir, spl) -> str:
    t = "double" if dtype == torch.float64 else "float"
    b = lambda v: "true" if v else "false"
    if model.family == 1:
        A, L, _, K1, M1, _ = model.seip_dims
        wg = _seip_wave_group(model)
        args = f"{t}, {method}, {_group_width(A)}, {L}, {K1}, {M1}" + (
            f", {wg[0]}, {wg[1]}" if wg is not None else (", 2" if _seip_tier_lanes(model, dtype) else ""))
        src = (f'#include "{os.path.join(_CSRC, "seip_kernel.hpp")}"\n'
               f"namespace dyn {{ template hipError_t launch_seip<{args}>(const KArgs<{t}> &, hipStream_t); }}\n"
               f'extern "C" void *dyn_extra_launch(void) {{\n'
               f"    return (void *)(hipError_t(*)(const dyn::KArgs<{t}> &, hipStream_t)) & dyn::launch_seip<{args}>;\n}}\n")
        if _seip_plain(model, dtype):
            # ... and the same mapping compiled without seasonal terms, introductions, schedules and discontinuity points
            # (Seip OPT bit 0; dynode_hip.hip kSeipPlain): what a call that uses none of them is dispatched to
            kt, nw = wg if wg is not None else ((2, 1) if _seip_tier_lanes(model, dtype) else (1, 1))
            full = f"{t}, {method}, {_group_width(A)}, {L}, {K1}, {M1}, {kt}, {nw}, 1"
            src += (f"namespace dyn {{ template hipError_t launch_seip<{full}>(const KArgs<{t}> &, hipStream_t); }}\n"
                    f'extern "C" void *dyn_extra_launch_plain(void) {{\n'
                    f"    return (void *)(hipError_t(*)(const dyn::KArgs<{t}> &, hipStream_t)) & dyn::launch_seip<{full}>;\n}}\n")
        return src
    head = (f"{t}, {method}, {_group_width(model.n_age)}, {model.n_strain}, {b(model.has_e)}, {b(model.has_wane)}, "
            f"{b(model.has_c)}, {model.n_wane}, {n_dir}, {spl}")
    args = f"{head}, {_features(model)}"
    src = (f'#include "{os.path.join(_CSRC, "solve_kernel.hpp")}"\n'
           f"namespace dyn {{ template hipError_t launch<{args}>(const KArgs<{t}> &, hipStream_t); }}\n"
           f'extern "C" void *dyn_extra_launch(void) {{\n'
           f"    return (void *)(hipError_t(*)(const dyn::KArgs<{t}> &, hipStream_t)) & dyn::launch<{args}>;\n}}\n")
    if _static_twin(model, dtype, n_dir):
        # ... the tangent instance with a static grid and adaptive steps without discontinuity points as compile-time facts (FEAT
        # bits 10 + 11; dynode_hip.hip: what a gradient-solve without a caller's order or discontinuity points is dispatched to)
        stat = f"{head}, {_features(model) | _STATIC_ADAPTIVE}"
        src += (f"namespace dyn {{ template hipError_t launch<{stat}>(const KArgs<{t}> &, hipStream_t); }}\n"
                f'extern "C" void *dyn_extra_launch_static(void) {{\n'
                f"    return (void *)(hipError_t(*)(const dyn::KArgs<{t}> &, hipStream_t)) & dyn::launch<{stat}>;\n}}\n")
    if _fused_twin(model, dtype, n_dir):
        # ... and the same gradient-solve with the sampler's side of a NUTS iteration behind it (FEAT bit 12; dynode_hip.hip
        # kFused): what a call that carries dyn_solver_opts::nuts_tail is dispatched to -- one launch per iteration
        full = f"{head}, {_features(model) | _FUSED}"
        src += (f"namespace dyn {{ template hipError_t launch<{full}>(const KArgs<{t}> &, hipStream_t); }}\n"
                f'extern "C" void *dyn_extra_launch_fused(void) {{\n'
                f"    return (void *)(hipError_t(*)(const dyn::KArgs<{t}> &, hipStream_t)) & dyn::launch<{full}>;\n}}\n")
    return src


def _build(name: str, source, entry: str, what: str) -> str:
    """``lib/jit/<name>.so`` from ``source()`` unless it is there already (callers hold ``_LOCK``); ``entry``: a symbol the
    finished library must export."""
    os.makedirs(_OUT, exist_ok=True)
    so = os.path.join(_OUT, name + ".so")
    if not os.path.exists(so):
        if not os.path.exists(HIPCC):
            raise RuntimeError(f"{HIPCC} not found: cannot build the kernel for {what}; add it to csrc/instances.def "
                               "on a machine with ROCm and rebuild")
        # one builder per shape across processes (torchrun ranks miss the same shape at the same moment): the others
        # wait on the lock file and then find the finished library
        import fcntl

        with open(os.path.join(_OUT, name + ".lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if not os.path.exists(so):
                    tag = f"{name}.{os.getpid()}"
                    src, tmp = os.path.join(_OUT, tag + ".hip"), os.path.join(_OUT, tag + ".tmp.so")
                    with open(src, "w") as f:          # a source file of this process's own: never rewritten under a reader
                        f.write(source())
                    print(f"[dynode_amd] building the kernel for {name} (one-off, ~15 s) ...", file=sys.stderr, flush=True)
                    try:
                        flags = ["-fno-slp-vectorize"]      # as csrc/Makefile (SOLVE_FLAGS, SEIP_FLAGS)
                        subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", *flags, src, "-o", tmp],
                                       check=True)
                        probe = ctypes.CDLL(tmp)           # refuse to publish a library without its entry point
                        getattr(probe, entry)
                        os.replace(tmp, so)                # atomic: nobody ever loads a half-written file
                    finally:
                        for leftover in (src, tmp):
                            if os.path.exists(leftover):
                                os.remove(leftover)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return so


def ensure_fused_twin(model: _abi.ModelDesc, dtype=torch.float32, method: str = "tsit5", n_dir: int = 1) -> bool:
    """Make sure the tangent kernel of (model shape, dtype, method, n_dir) has the twin that carries the sampler's side of a
    NUTS iteration (``dyn_solver_opts::nuts_tail``, FEAT bit 12): shapes of ``csrc/instances.def`` other than the inference
    examples' have the tangent kernel built in without it.  True when the twin exists afterwards (built in, loaded or built now);
    False when it cannot (no tangent kernel, another family / dtype, no hipcc, JIT disabled) -- the sampler then keeps two
    launches per iteration."""
    mid = {"tsit5": _abi.DYN_TSIT5, "dopri5": _abi.DYN_DOPRI5}[method]
    L = _abi.lib()
    opts = _abi.SolverOptsC(mid, _abi.DYN_F64 if dtype == torch.float64 else _abi.DYN_F32, 1e-5, 1e-6, 10**6, 0.0, None, 0)
    mc = model.c()
    have = int(L.dyn_fused_twin(ctypes.byref(mc), ctypes.byref(opts), n_dir))
    if have >= 0:
        return have == 1
    if not (enabled() and _fused_twin(model, dtype, n_dir) and os.path.exists(HIPCC)):
        return False
    spl = -have
    name = _name(model, dtype, mid, n_dir, spl) + "_twin"
    with _LOCK:
        if name in _LOADED:
            return True
        t = "float"
        b = lambda v: "true" if v else "false"   # noqa: E731
        full = (f"{t}, {mid}, {_group_width(model.n_age)}, {model.n_strain}, {b(model.has_e)}, {b(model.has_wane)}, "
                f"{b(model.has_c)}, {model.n_wane}, {n_dir}, {spl}, {_features(model) | _FUSED}")
        text = (f'#include "{os.path.join(_CSRC, "solve_kernel.hpp")}"\n'
                f"namespace dyn {{ template hipError_t launch<{full}>(const KArgs<{t}> &, hipStream_t); }}\n"
                f'extern "C" void *dyn_extra_launch_fused(void) {{\n'
                f"    return (void *)(hipError_t(*)(const dyn::KArgs<{t}> &, hipStream_t)) & dyn::launch<{full}>;\n}}\n")
        so = _build(name, lambda: text, "dyn_extra_launch_fused", f"{model} (fused sampler twin)")
        extra = ctypes.CDLL(so)
        extra.dyn_extra_launch_fused.restype = ctypes.c_void_p
        rc = L.dyn_register_instance(opts.dtype, mid, _group_width(model.n_age), model.n_strain, int(model.has_e),
                                     int(model.has_wane), int(model.has_c), model.n_wane, n_dir, spl,
                                     _features(model, dtype) | _FUSED, ctypes.c_void_p(extra.dyn_extra_launch_fused()))
        if rc:
            raise RuntimeError(f"dyn_register_instance (fused twin): {_abi.ERR_NAMES.get(rc, rc)}")
        _LOADED[name] = extra
    return True


def ensure_lean_twin(model: _abi.ModelDesc, dtype, method: str, n_dir: int, obs_compartment: int, increments: bool) -> bool:
    """Make sure the tangent kernel of (model shape, dtype, method, n_dir) has its LEAN twin for a fused likelihood on
    ``obs_compartment`` (values or increments) -- the instance with the scored compartment, the likelihood mode and the model's
    run-time switches as compile-time facts (`dyn_lean_twin`; FEAT bit 13 + bits 17-20) -- and that twin's sampler-carrying
    twin (bit 12).  Built in for the two inference examples; for any other model a sampler is about to take thousands of
    gradient-solves on, built here on first use (one-off, ~40 s, kept in lib/jit).  True when the twin exists afterwards; False
    where no lean instance applies (seasonal forcing, introduced strains, float64, no hipcc, JIT disabled): the call keeps the
    general tangent instance."""
    mid = {"tsit5": _abi.DYN_TSIT5, "dopri5": _abi.DYN_DOPRI5}[method]
    L = _abi.lib()
    opts = _abi.SolverOptsC(mid, _abi.DYN_F64 if dtype == torch.float64 else _abi.DYN_F32, 1e-5, 1e-6, 10**6, 0.0, None, 0)
    mc = model.c()
    spl, feat = ctypes.c_int32(0), ctypes.c_int32(0)
    have = int(L.dyn_lean_twin(ctypes.byref(mc), ctypes.byref(opts), n_dir, int(obs_compartment), int(bool(increments)),
                               ctypes.byref(spl), ctypes.byref(feat)))
    if have >= 0:
        return have == 1
    if not (enabled() and dtype == torch.float32 and n_dir in (1, 2) and model.vax_lanes == 0 and os.path.exists(HIPCC)):
        return False
    name = _name(model, dtype, mid, n_dir, spl.value) + f"_lean{feat.value:x}"
    with _LOCK:
        if name in _LOADED:
            return True
        t = "float"
        b = lambda v: "true" if v else "false"   # noqa: E731
        head = (f"{t}, {mid}, {_group_width(model.n_age)}, {model.n_strain}, {b(model.has_e)}, {b(model.has_wane)}, "
                f"{b(model.has_c)}, {model.n_wane}, {n_dir}, {spl.value}")
        text = f'#include "{os.path.join(_CSRC, "solve_kernel.hpp")}"\n'
        for sym, f_ in (("dyn_extra_launch_lean", feat.value), ("dyn_extra_launch_lean_fused", feat.value | _FUSED)):
            text += (f"namespace dyn {{ template hipError_t launch<{head}, {f_}>(const KArgs<{t}> &, hipStream_t); }}\n"
                     f'extern "C" void *{sym}(void) {{\n'
                     f"    return (void *)(hipError_t(*)(const dyn::KArgs<{t}> &, hipStream_t)) & dyn::launch<{head}, {f_}>;\n}}\n")
        so = _build(name, lambda: text, "dyn_extra_launch_lean", f"{model} (lean twin, features {feat.value:#x})")
        extra = ctypes.CDLL(so)
        for sym, f_ in (("dyn_extra_launch_lean", feat.value), ("dyn_extra_launch_lean_fused", feat.value | _FUSED)):
            fn = getattr(extra, sym)
            fn.restype = ctypes.c_void_p
            rc = L.dyn_register_instance(opts.dtype, mid, _group_width(model.n_age), model.n_strain, int(model.has_e),
                                         int(model.has_wane), int(model.has_c), model.n_wane, n_dir, spl.value, f_, ctypes.c_void_p(fn()))
            if rc:
                raise RuntimeError(f"dyn_register_instance (lean twin): {_abi.ERR_NAMES.get(rc, rc)}")
        _LOADED[name] = extra
    return True


def ensure_kernel(model: _abi.ModelDesc, dtype=torch.float32, method: str = "tsit5", n_dir: int = 0) -> bool:
    """Make sure a kernel for (model shape, dtype, method, n_dir) exists; returns True if one had to be
    built or loaded.  Raises if hipcc is missing or the shape cannot be mapped to lanes."""
    mid = {"tsit5": _abi.DYN_TSIT5, "dopri5": _abi.DYN_DOPRI5}[method]
    L = _abi.lib()
    opts = _abi.SolverOptsC(mid, _abi.DYN_F64 if dtype == torch.float64 else _abi.DYN_F32, 1e-5, 1e-6, 10**6, 0.0, None, 0)
    mc = model.c()
    have = (L.dyn_is_supported_jvp(ctypes.byref(mc), ctypes.byref(opts), n_dir) if n_dir
            else L.dyn_is_supported(ctypes.byref(mc), ctypes.byref(opts)))
    if have:
        return False
    spl = choose_spl(model, n_dir)
    name = _name(model, dtype, mid, n_dir, spl)
    with _LOCK:
        if name in _LOADED:
            return False
        so = _build(name, lambda: _source(model, dtype, mid, n_dir, spl), "dyn_extra_launch", f"{model}")
        extra = ctypes.CDLL(so)
        extra.dyn_extra_launch.restype = ctypes.c_void_p
        rc = L.dyn_register_instance(opts.dtype, mid, _group_width(model.n_age), model.n_strain, int(model.has_e),
                                     int(model.has_wane), int(model.has_c), model.n_wane, n_dir, spl,
                                     _features(model, dtype), ctypes.c_void_p(extra.dyn_extra_launch()))
        if rc:
            raise RuntimeError(f"dyn_register_instance: {_abi.ERR_NAMES.get(rc, rc)}")
        if _seip_plain(model, dtype) and hasattr(extra, "dyn_extra_launch_plain"):
            extra.dyn_extra_launch_plain.restype = ctypes.c_void_p
            rc = L.dyn_register_instance(opts.dtype, mid, _group_width(model.n_age), model.n_strain, int(model.has_e),
                                         int(model.has_wane), int(model.has_c), model.n_wane, n_dir, spl,
                                         _features(model, dtype) | _SEIP_PLAIN, ctypes.c_void_p(extra.dyn_extra_launch_plain()))
            if rc:
                raise RuntimeError(f"dyn_register_instance (plain): {_abi.ERR_NAMES.get(rc, rc)}")
        if _fused_twin(model, dtype, n_dir) and hasattr(extra, "dyn_extra_launch_fused"):
            extra.dyn_extra_launch_fused.restype = ctypes.c_void_p
            rc = L.dyn_register_instance(opts.dtype, mid, _group_width(model.n_age), model.n_strain, int(model.has_e),
                                         int(model.has_wane), int(model.has_c), model.n_wane, n_dir, spl,
                                         _features(model, dtype) | _FUSED, ctypes.c_void_p(extra.dyn_extra_launch_fused()))
            if rc:
                raise RuntimeError(f"dyn_register_instance (fused): {_abi.ERR_NAMES.get(rc, rc)}")
        if _static_twin(model, dtype, n_dir) and hasattr(extra, "dyn_extra_launch_static"):
            extra.dyn_extra_launch_static.restype = ctypes.c_void_p
            rc = L.dyn_register_instance(opts.dtype, mid, _group_width(model.n_age), model.n_strain, int(model.has_e),
                                         int(model.has_wane), int(model.has_c), model.n_wane, n_dir, spl,
                                         _features(model, dtype) | _STATIC_ADAPTIVE, ctypes.c_void_p(extra.dyn_extra_launch_static()))
            if rc:
                raise RuntimeError(f"dyn_register_instance (static twin): {_abi.ERR_NAMES.get(rc, rc)}")
        _LOADED[name] = extra
    return True
