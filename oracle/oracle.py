"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE, not product).

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may import
this module.  The shipped path is ``dynode_amd`` -> ``libdynode_hip.so``.

The oracle restates the algorithm behind ``dynode.simulation.simulate``
(/root/reference/src/dynode/simulation/odes.py:35-145) -- see dynode_oracle.h for the
full citation list and the "parity unpinned" statement.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdynode_oracle.so")


class _ModelDesc(ctypes.Structure):
    _fields_ = [
        ("n_age", ctypes.c_int32),
        ("n_strain", ctypes.c_int32),
        ("has_e", ctypes.c_int32),
        ("has_wane", ctypes.c_int32),
        ("has_c", ctypes.c_int32),
        ("n_wane", ctypes.c_int32),
        ("normalize", ctypes.c_int32),
        ("seasonal", ctypes.c_int32),
        ("has_intro", ctypes.c_int32),
        ("n_vax_tiers", ctypes.c_int32),
        ("intro_age_mask", ctypes.c_uint64 * 8),
        ("n_vax_knots", ctypes.c_int32),
        ("family", ctypes.c_int32),
        ("seasonal_vax", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class _SolverOpts(ctypes.Structure):
    _fields_ = [
        ("method", ctypes.c_int32),
        ("dtype", ctypes.c_int32),
        ("rtol", ctypes.c_double),
        ("atol", ctypes.c_double),
        ("max_steps", ctypes.c_int64),
        ("constant_dt", ctypes.c_double),
        ("jump_ts", ctypes.POINTER(ctypes.c_double)),
        ("n_jump", ctypes.c_int32),
    ]


@dataclass(frozen=True)
class Model:
    """Plain description of one member of the compartmental RHS family."""

    n_age: int = 1
    n_strain: int = 1
    has_e: bool = False
    has_wane: bool = False
    has_c: bool = False
    n_wane: int = 1
    normalize: bool = True
    seasonal: bool = False
    has_intro: bool = False
    intro_age_mask: tuple = ()      # per strain: bit a = age bin a receives external introductions
    n_vax_tiers: int = 0            # > 1: n_age enumerates (age, tier) groups (see include/dynode_hip.h)
    n_vax_knots: int = 0
    family: int = 0                 # 1 = SEIP with immune histories (rhs_seip in dynode_oracle_impl.inc)
    seasonal_vax: bool = False

    def c(self) -> _ModelDesc:
        masks = tuple(int(v) for v in self.intro_age_mask) + (0,) * (8 - len(self.intro_age_mask))
        return _ModelDesc(
            self.n_age, self.n_strain, int(self.has_e), int(self.has_wane), int(self.has_c),
            self.n_wane, int(self.normalize), int(self.seasonal), int(self.has_intro), int(self.n_vax_tiers),
            (ctypes.c_uint64 * 8)(*masks), int(self.n_vax_knots), int(self.family), int(self.seasonal_vax), 0,
        )


def build(force: bool = False) -> str:
    """Compile the oracle with gcc via oracle/Makefile (building the checker is not using it)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.dyo_state_dim.argtypes = [ctypes.POINTER(_ModelDesc)]
        _lib.dyo_param_dim.argtypes = [ctypes.POINTER(_ModelDesc)]
        _lib.dyo_n_compartments.argtypes = [ctypes.POINTER(_ModelDesc)]
        _lib.dyo_compartment_offsets.argtypes = [ctypes.POINTER(_ModelDesc), ctypes.c_void_p]
        _lib.dyo_tableau_ptr.restype = ctypes.POINTER(ctypes.c_double)
        _lib.dyo_tableau_ptr.argtypes = [ctypes.c_int32, ctypes.c_int32]
        _lib.dyo_tsit5_dense_weights_f64.argtypes = [ctypes.c_double, ctypes.c_void_p]
        _lib.dyo_rhs_f64.argtypes = [
            ctypes.POINTER(_ModelDesc), ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_void_p,
        ]
        _lib.dyo_solve_batch_cpu.restype = ctypes.c_int
        _lib.dyo_solve_batch_cpu.argtypes = [
            ctypes.POINTER(_ModelDesc), ctypes.POINTER(_SolverOpts), ctypes.c_void_p,
            ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_double,
            ctypes.c_double, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
        ]
    return _lib


def state_dim(model: Model) -> int:
    return lib().dyo_state_dim(ctypes.byref(model.c()))


def param_dim(model: Model) -> int:
    return lib().dyo_param_dim(ctypes.byref(model.c()))


def compartment_offsets(model: Model) -> np.ndarray:
    n = lib().dyo_n_compartments(ctypes.byref(model.c()))
    off = np.zeros(n + 1, dtype=np.int32)
    lib().dyo_compartment_offsets(ctypes.byref(model.c()), off.ctypes.data)
    return off


def tableau(method: str):
    """(c[7], a[7,7], berr[7], cmid[7]) exactly as compiled into the oracle."""
    mid = {"tsit5": 0, "dopri5": 1}[method]
    get = lambda which, n: np.ctypeslib.as_array(lib().dyo_tableau_ptr(mid, which), (n,)).copy()
    return get(0, 7), get(1, 49).reshape(7, 7), get(2, 7), get(3, 7)


def tsit5_dense_weights(theta: float) -> np.ndarray:
    b = np.zeros(7)
    lib().dyo_tsit5_dense_weights_f64(float(theta), b.ctypes.data)
    return b


def rhs(model: Model, t: float, y, params, contact) -> np.ndarray:
    y = np.ascontiguousarray(y, dtype=np.float64)
    p = np.ascontiguousarray(params, dtype=np.float64)
    c = np.ascontiguousarray(contact, dtype=np.float64)
    assert y.size == state_dim(model) and p.size == param_dim(model)
    assert c.size == model.n_age**2
    out = np.zeros_like(y)
    lib().dyo_rhs_f64(ctypes.byref(model.c()), float(t), y.ctypes.data, p.ctypes.data,
                      c.ctypes.data, out.ctypes.data)
    return out


def solve(model: Model, y0, params, contact, t1, save_ts, *, t0=0.0, method="tsit5",
          dtype=np.float32, rtol=1e-5, atol=1e-6, max_steps=10**6, constant_dt=0.0,
          jump_ts=(), save_mask=None, n_threads=1):
    """Batched CPU solve.  Returns (ys [B, n_save, D_saved], status, n_accept, n_reject)."""
    dtype = np.dtype(dtype)
    assert dtype in (np.dtype(np.float32), np.dtype(np.float64))
    D, P, A = state_dim(model), param_dim(model), model.n_age
    params = np.ascontiguousarray(params, dtype=dtype).reshape(-1, P)
    B = params.shape[0]
    y0 = np.ascontiguousarray(y0, dtype=dtype)
    batched = y0.ndim == 2
    assert y0.shape == ((B, D) if batched else (D,)), (y0.shape, B, D)
    contact = np.ascontiguousarray(contact, dtype=dtype).reshape(A, A)
    save_ts = np.ascontiguousarray(save_ts, dtype=dtype)
    n_save = save_ts.shape[0]
    off = compartment_offsets(model)
    ncomp = len(off) - 1
    if save_mask is None:
        mask_arr, d_saved = None, D
    else:
        mask_arr = np.ascontiguousarray(save_mask, dtype=np.uint8)
        assert mask_arr.shape == (ncomp,)
        d_saved = int(sum(off[c + 1] - off[c] for c in range(ncomp) if mask_arr[c]))
    ys = np.empty((B, n_save, d_saved), dtype=dtype)
    status = np.zeros(B, dtype=np.int32)
    n_acc = np.zeros(B, dtype=np.int32)
    n_rej = np.zeros(B, dtype=np.int32)
    jt = np.ascontiguousarray(jump_ts, dtype=np.float64)
    opts = _SolverOpts(
        {"tsit5": 0, "dopri5": 1}[method], 0 if dtype == np.float32 else 1, rtol, atol,
        int(max_steps), float(constant_dt),
        jt.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) if jt.size else None, int(jt.size),
    )
    rc = lib().dyo_solve_batch_cpu(
        ctypes.byref(model.c()), ctypes.byref(opts), y0.ctypes.data, int(batched),
        params.ctypes.data, contact.ctypes.data, B, float(t0), float(t1), save_ts.ctypes.data,
        n_save, mask_arr.ctypes.data if mask_arr is not None else None, ys.ctypes.data,
        status.ctypes.data, n_acc.ctypes.data, n_rej.ctypes.data, int(n_threads),
    )
    if rc != 0:
        raise ValueError(f"dyo_solve_batch_cpu argument error {rc}")
    return ys, status, n_acc, n_rej
