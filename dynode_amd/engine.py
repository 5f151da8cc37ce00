"""Batched-solve driver: torch tensors in HBM -> ``dyn_solve_batch`` -> torch tensors in HBM.

PyTorch is plumbing here (device memory, streams); the arithmetic is the HIP kernel in
csrc/solve_kernel.hpp reached through the C-ABI.  There is no CPU path: without a GPU or
without ``libdynode_hip.so`` every call raises.
"""

from __future__ import annotations

import contextlib
import ctypes
import threading
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from . import _abi
from ._abi import ModelDesc

_METHODS = {"tsit5": _abi.DYN_TSIT5, "dopri5": _abi.DYN_DOPRI5}
_DTYPES = {torch.float32: _abi.DYN_F32, torch.float64: _abi.DYN_F64}


class SolveError(RuntimeError):
    """An argument error reported by the C-ABI (negative DYN_ERR_* code)."""

    def __init__(self, code: int, detail: str = ""):
        self.code = code
        name = _abi.ERR_NAMES.get(code, str(code))
        super().__init__(f"dyn_solve_batch failed: {name}" + (f" ({detail})" if detail else ""))


def _call_with_jit(call, L, model, dtype, method, n_dir) -> int:
    """Run the library call; if the only problem is a kernel shape that is not compiled in, build and
    register it (dynode_amd/jit.py) and call again."""
    rc = call()
    if rc == -7 and L.dyn_last_error().decode().startswith(("no kernel compiled", "no SEIP kernel compiled")):
        from . import jit

        if jit.enabled() and jit.ensure_kernel(model, dtype, method, n_dir):
            rc = call()
    return rc


@dataclass
class BatchResult:
    """Outputs of one batched solve, all resident on the device."""

    ys: torch.Tensor        # [B, n_save, D_saved]
    status: torch.Tensor    # [B] int32: 0 ok, 1 max_steps, 2 non-finite
    n_accept: torch.Tensor  # [B] int32
    n_reject: torch.Tensor  # [B] int32
    saved: tuple            # names of the saved compartments, in row order
    sizes: tuple            # flat size of each saved compartment
    dys: Optional[torch.Tensor] = None  # [B, n_save, n_dir, D_saved] tangents (solve_batch_jvp)
    schedule: Optional[tuple] = None    # (steps [B, cap, 2], count [B]) accepted steps, when recorded (record_steps=cap)


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError(
            "dynode_amd needs an AMD GPU (gfx950): torch.cuda.is_available() is False and "
            "there is no CPU fallback."
        )
    return torch.device("cuda", torch.cuda.current_device())


_CONST_CACHE: "dict[tuple, torch.Tensor]" = {}
_CONST_CACHE_MAX_BYTES = 1 << 16   # only small, typically constant inputs (y0, contact matrix, save grid)


def _dev(x, dtype, device) -> torch.Tensor:
    """Device tensor of ``x``.  Small host arrays are cached by content, so loops that pass the same
    constants again and again (NUTS: y0, contact matrix, save grid at every gradient-solve) do not
    pay a blocking host-to-device copy per call -- which also keeps the call HIP-graph capturable."""
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype).contiguous()
    arr = np.ascontiguousarray(x)
    if arr.nbytes <= _CONST_CACHE_MAX_BYTES:
        key = (arr.dtype.str, arr.shape, arr.tobytes(), dtype, str(device))
        hit = _CONST_CACHE.get(key)
        if hit is None:
            if len(_CONST_CACHE) > 256:
                _CONST_CACHE.clear()
            hit = torch.as_tensor(arr, dtype=dtype, device=device).contiguous()
            _CONST_CACHE[key] = hit
        return hit
    return torch.as_tensor(arr, dtype=dtype, device=device).contiguous()


# ---- dispatch hints (dyn_solver_opts::hints, ABI 9).  Which compiled instance / lane mapping / grid shape runs a call is the
# library's measured choice; tests and tuning tools pin one for the calls made inside a `with dispatch_hints(...)` block of
# the calling thread.  (Up to ABI 8 these were DYNODE_HIP_* environment variables read inside the library.)
_HINT_FIELDS = tuple(n for n, _ in _abi.DispatchHintsC._fields_)
_HINTS = threading.local()


def current_hints() -> dict:
    return dict(getattr(_HINTS, "value", None) or {})


def _merged_hints(kw: dict) -> dict:
    bad = set(kw) - set(_HINT_FIELDS) - {"work_min_batch"}
    if bad:
        raise TypeError(f"unknown dispatch hint(s) {sorted(bad)}; known: {_HINT_FIELDS + ('work_min_batch',)}")
    merged = dict(getattr(_HINTS, "value", None) or {})
    for k, v in kw.items():
        if v is None:
            merged.pop(k, None)
        else:
            merged[k] = int(v) + 1 if k == "replicas_log2" else int(v)
    return merged


@contextlib.contextmanager
def dispatch_hints(**kw):
    """Pin dispatch choices for the solves of this thread inside the block.  Keys = fields of ``dyn_dispatch_hints``
    (include/dynode_hip.h): ``pull`` (1 / -1), ``pull_waves`` (n), ``strains_per_lane`` (n), ``replicas_log2`` (k: exactly
    2^k replicas -- stored as k + 1), ``producer_consumer`` (1), ``general_instance`` (1), ``seip_tier_lanes`` (1 / -1),
    ``seip_tier_waves`` (-1); plus ``work_min_batch`` (host side: the smallest batch that gets a work counter).  Nested
    blocks merge; ``None`` removes a key."""
    before = getattr(_HINTS, "value", None)
    _HINTS.value = _merged_hints(kw)
    try:
        yield
    finally:
        _HINTS.value = before


def set_dispatch_hints(**kw) -> None:
    """`dispatch_hints` without a block: merge into this thread's hints until `clear_dispatch_hints` (test fixtures)."""
    _HINTS.value = _merged_hints(kw)


def clear_dispatch_hints() -> None:
    _HINTS.value = None


def _apply_hints(opts: "_abi.SolverOptsC") -> None:
    for k, v in (getattr(_HINTS, "value", None) or {}).items():
        if k in _HINT_FIELDS:
            setattr(opts.hints, k, v)


_WORK_COUNTERS: "dict[tuple, torch.Tensor]" = {}
# fewer trajectories than the GPU has SIMDs can never exceed one resident round (dispatch_hints(work_min_batch=1): the tests
# pull small batches through forced grids of a few waves)
_WORK_MIN_BATCH = 1024


def work_counter(B: int, device, stream) -> Optional[torch.Tensor]:
    """The two zeroed int32 words behind ``dyn_solver_opts.work_counter`` (work pulling, include/dynode_hip.h): one pair per
    (device, stream), created once -- the kernel leaves it zeroed, and launches of one stream cannot overlap.  While a HIP
    graph is being captured ON THAT STREAM the pair is a fresh one from the graph's own pool (replays of different graphs
    may overlap).  None for batches that can never exceed one resident round.

    A pulling launch that does not run to completion (a GPU fault, a reset) leaves the words non-zero, and the next launch
    would start its tickets past the end of the queue: `drop_work_counters` forgets the cached pairs -- `solve_batch` calls
    it whenever the library reports an error, and callers who catch a failed synchronize should too."""
    if B < int(current_hints().get("work_min_batch", _WORK_MIN_BATCH)):
        return None
    with torch.cuda.stream(stream):
        if torch.cuda.is_current_stream_capturing():
            return torch.zeros(2, dtype=torch.int32, device=device)
    key = (str(device), int(stream.cuda_stream))
    t = _WORK_COUNTERS.get(key)
    if t is None:
        if len(_WORK_COUNTERS) > 64:
            _WORK_COUNTERS.clear()
        with torch.cuda.stream(stream):
            t = _WORK_COUNTERS[key] = torch.zeros(2, dtype=torch.int32, device=device)
    return t


def drop_work_counters() -> None:
    """Forget the cached work-counter pairs (they are re-created zeroed on the next use)."""
    _WORK_COUNTERS.clear()


def save_mask_bytes(model: ModelDesc, save_mask: Optional[Sequence[bool]]):
    names = model.compartment_names
    if save_mask is None:
        return None, names, model.compartment_sizes
    mask = [bool(v) for v in save_mask]
    if len(mask) != len(names):
        raise ValueError(f"save_mask needs {len(names)} entries ({names}), got {len(mask)}")
    saved = tuple(n for n, m in zip(names, mask) if m)
    sizes = tuple(s for s, m in zip(model.compartment_sizes, mask) if m)
    return (ctypes.c_uint8 * len(mask))(*[int(v) for v in mask]), saved, sizes


def solve_batch(model: ModelDesc, y0, params, contact, t1: float, save_ts, *, t0: float = 0.0,
                method: str = "tsit5", dtype: torch.dtype = torch.float32, rtol: float = 1e-5,
                atol: float = 1e-6, max_steps: int = 10**6, constant_dt: float = 0.0,
                jump_ts: Sequence[float] = (), save_mask: Optional[Sequence[bool]] = None,
                out: Optional[torch.Tensor] = None, stats_out: Optional[tuple] = None,
                stream: Optional[torch.cuda.Stream] = None, dparams=None, dy0=None,
                dout: Optional[torch.Tensor] = None, record_steps: int = 0, replay: Optional[tuple] = None,
                order=None) -> BatchResult:
    """Integrate B parameter samples of ``model`` over [t0, t1] on the current GPU.

    Replaces the per-sample ``diffeqsolve`` call of dynode.simulation.simulate
    (/root/reference/src/dynode/simulation/odes.py:133-144).  Asynchronous: work is enqueued
    on ``stream`` (default: torch's current stream); nothing is synchronised here.

    With ``dparams`` ([B, n_dir, P] seed directions, optionally ``dy0`` [n_dir, D] / [B, n_dir, D])
    the forward-mode tangents of the saved trajectory are computed in the same launch
    (``dyn_solve_batch_jvp``) and returned as ``BatchResult.dys`` [B, n_save, n_dir, D_saved].

    Step schedules (SEIP family): ``record_steps=cap`` also returns the accepted steps of every trajectory
    (``BatchResult.schedule``); ``replay=(steps, count, leader)`` makes every trajectory take the recorded steps of row
    ``leader[b]`` of ``steps`` (``leader=None``: row b) instead of controlling its own.  For this family ``dparams`` is
    served by `_replayed_tangents`: central differences of replayed solves on the primal's step sequence.

    Grid: one wave per ``trajectories_per_wave`` trajectories, started by the hardware as earlier waves retire (a static
    grid).  The library launches a resident grid whose lane groups PULL trajectories from a queue as they finish
    (``dyn_solver_opts.work_counter``; csrc/stepper.hpp) only where that was measured to pay: when the caller supplies the
    queue (``order``) and a wave holds more than two trajectories.

    ``order``: the queue (``dyn_solve_batch_ordered``; it never changes a result).  ``None`` (default): the batch in its
    given order; an int32 device tensor [B]: that permutation -- a caller who knows which trajectories are expensive puts
    them first.

    float32 and the batch size: a batch that fills at most half a wave per SIMD takes a finer strain split (a shorter serial
    instruction stream), whose summation order differs -- the same parameter row solved in a 3072-row and in a 65536-row
    batch can differ in the last bits (and, rarely, in an accept / reject decision; float64 step counts do not).
    ``with dispatch_hints(strains_per_lane=n)`` pins the mapping (include/dynode_hip.h, dyn_trajectories_per_wave_for_batch).
    """
    device = require_gpu()
    L = _abi.lib()
    if model.family == 1 and dparams is not None:
        return _replayed_tangents(model, y0, params, contact, t1, save_ts, dparams, dy0, t0=t0, method=method, dtype=dtype,
                                  rtol=rtol, atol=atol, max_steps=max_steps, constant_dt=constant_dt, jump_ts=jump_ts,
                                  save_mask=save_mask, out=out, stats_out=stats_out, stream=stream, dout=dout)
    D, P, A = model.state_dim, model.param_dim, model.n_age
    params_t = _dev(params, dtype, device).reshape(-1, P)
    B = params_t.shape[0]
    y0_t = _dev(y0, dtype, device)
    batched = y0_t.dim() == 2
    if tuple(y0_t.shape) != ((B, D) if batched else (D,)):
        raise ValueError(f"y0 has shape {tuple(y0_t.shape)}, expected {(D,)} or {(B, D)}")
    contact_t = _dev(contact, dtype, device)
    if contact_t.numel() != A * A:
        raise ValueError(f"contact matrix must have {A}x{A} entries")
    ts_t = _dev(save_ts, dtype, device).reshape(-1)
    n_save = ts_t.shape[0]
    mask_c, saved, sizes = save_mask_bytes(model, save_mask)
    d_saved = int(sum(sizes))
    if out is None:
        out = torch.empty((B, n_save, d_saved), dtype=dtype, device=device)
    elif tuple(out.shape) != (B, n_save, d_saved) or out.dtype != dtype or not out.is_contiguous():
        raise ValueError("`out` must be a contiguous [B, n_save, D_saved] tensor of `dtype`")
    if stats_out is None:
        stats = torch.empty((3, B), dtype=torch.int32, device=device)
        status, n_acc, n_rej = stats[0], stats[1], stats[2]
    else:
        status, n_acc, n_rej = stats_out
    n_dir = 0
    dparams_t = dy0_t = None
    if dparams is not None:
        dparams_t = _dev(dparams, dtype, device)
        if dparams_t.dim() != 3 or dparams_t.shape[0] != B or dparams_t.shape[2] != P:
            raise ValueError(f"dparams must have shape [B={B}, n_dir, P={P}], got {tuple(dparams_t.shape)}")
        n_dir = dparams_t.shape[1]
        if dy0 is not None:
            dy0_t = _dev(dy0, dtype, device)
            if tuple(dy0_t.shape) not in ((n_dir, D), (B, n_dir, D)):
                raise ValueError(f"dy0 must have shape {(n_dir, D)} or {(B, n_dir, D)}")
        if dout is None:
            dout = torch.empty((B, n_save, n_dir, d_saved), dtype=dtype, device=device)
        elif tuple(dout.shape) != (B, n_save, n_dir, d_saved) or dout.dtype != dtype or not dout.is_contiguous():
            raise ValueError("`dout` must be a contiguous [B, n_save, n_dir, D_saved] tensor of `dtype`")
    if B == 0:  # empty batch: nothing to enqueue (zero-size tensors have null data pointers)
        return BatchResult(out, status, n_acc, n_rej, saved, sizes, dout)
    jt = np.ascontiguousarray(jump_ts, dtype=np.float64)
    s = stream if stream is not None else torch.cuda.current_stream(device)
    work_t = work_counter(B, device, s) if model.family == 0 else None
    opts = _abi.SolverOptsC(
        _METHODS[method], _DTYPES[dtype], float(rtol), float(atol), int(max_steps),
        float(constant_dt),
        jt.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) if jt.size else None, int(jt.size),
        work_t.data_ptr() if work_t is not None else None)
    _apply_hints(opts)
    sched = sched_n = leader_t = None
    if record_steps and replay is not None:
        raise ValueError("record_steps and replay exclude each other")
    if record_steps:
        sched = torch.empty((B, int(record_steps), 2), dtype=dtype, device=device)
        sched_n = torch.empty((B,), dtype=torch.int32, device=device)
    if replay is not None:
        sched, sched_n, leader = replay
        if sched.dtype != dtype or sched.dim() != 3 or sched.shape[2] != 2 or not sched.is_contiguous():
            raise ValueError("replay steps must be a contiguous [n_leaders, cap, 2] tensor of the solve dtype")
        if leader is not None:
            leader_t = torch.as_tensor(leader, dtype=torch.int64, device=device).contiguous()
            if leader_t.shape != (B,):
                raise ValueError(f"replay leader must have shape ({B},)")
        elif sched.shape[0] != B:
            raise ValueError("replay without a leader index needs one schedule row per trajectory")

    plain = n_dir == 0 and not record_steps and replay is None
    order_t = None
    if isinstance(order, torch.Tensor):
        if not plain:
            raise ValueError("a dispatch order goes with the plain solve (no tangents, no step schedules)")
        if order.dtype != torch.int32 or tuple(order.shape) != (B,) or not order.is_contiguous() or order.device != params_t.device:
            raise ValueError(f"order must be a contiguous int32 device tensor of shape ({B},)")
        order_t = order
    elif order is not None:
        raise ValueError("order must be None or an int32 device tensor")

    def call():
        common = (ctypes.byref(model.c()), ctypes.byref(opts), y0_t.data_ptr(), int(batched),
                  params_t.data_ptr(), contact_t.data_ptr(), B, float(t0), float(t1), ts_t.data_ptr(),
                  n_save, mask_c, out.data_ptr(), status.data_ptr(), n_acc.data_ptr(), n_rej.data_ptr())
        if order_t is not None:
            return L.dyn_solve_batch_ordered(*common, order_t.data_ptr(), ctypes.c_void_p(s.cuda_stream))
        if record_steps:
            return L.dyn_solve_batch_record(*common, sched.data_ptr(), sched_n.data_ptr(), int(record_steps), ctypes.c_void_p(s.cuda_stream))
        if replay is not None:
            return L.dyn_solve_batch_replay(*common, sched.data_ptr(), sched_n.data_ptr(),
                                            leader_t.data_ptr() if leader_t is not None else None, int(sched.shape[1]),
                                            ctypes.c_void_p(s.cuda_stream))
        if n_dir == 0:
            return L.dyn_solve_batch(
                ctypes.byref(model.c()), ctypes.byref(opts), y0_t.data_ptr(), int(batched),
                params_t.data_ptr(), contact_t.data_ptr(), B, float(t0), float(t1), ts_t.data_ptr(),
                n_save, mask_c, out.data_ptr(), status.data_ptr(), n_acc.data_ptr(), n_rej.data_ptr(),
                ctypes.c_void_p(s.cuda_stream))
        return L.dyn_solve_batch_jvp(
            ctypes.byref(model.c()), ctypes.byref(opts), y0_t.data_ptr(), int(batched),
            params_t.data_ptr(), contact_t.data_ptr(), B, float(t0), float(t1), ts_t.data_ptr(),
            n_save, mask_c, n_dir, dparams_t.data_ptr(),
            dy0_t.data_ptr() if dy0_t is not None else None,
            int(dy0_t is not None and dy0_t.dim() == 3), out.data_ptr(), dout.data_ptr(),
            status.data_ptr(), n_acc.data_ptr(), n_rej.data_ptr(), ctypes.c_void_p(s.cuda_stream))

    if work_t is not None and (order_t is not None or current_hints().get("pull", 0) > 0 or current_hints().get("pull_waves", 0) > 0):
        # a launch that may pull: rows the queue never hands out (counters left non-zero by a launch that did not run to its
        # end) must not read as solved -- every row starts at status -1 and the kernel overwrites it with 0 / 1 / 2
        with torch.cuda.stream(s):
            status.fill_(-1)
    rc = _call_with_jit(call, L, model, dtype, method, n_dir)
    if rc != 0:
        drop_work_counters()
        raise SolveError(rc, L.dyn_last_error().decode())
    # keep inputs alive until the stream has consumed them
    for t in (y0_t, params_t, contact_t, ts_t, dparams_t, dy0_t, sched, sched_n, leader_t, order_t, work_t):
        if t is not None:
            t.record_stream(s)
    return BatchResult(out, status, n_acc, n_rej, saved, sizes, dout, (sched, sched_n) if record_steps else None)


# relative size of the central-difference step of `_replayed_tangents`, per solve dtype: truncation ~ h^2, rounding ~ eps / h
_FD_REL_STEP = {torch.float32: 2e-3, torch.float64: 1e-5}
SCHEDULE_CAP = 2048      # accepted steps recorded per trajectory (the adaptive SEIP ensembles take about 250-350)


def schedule_capacity(model: ModelDesc, dtype, n_save: int) -> int:
    """Steps per trajectory a replayed schedule may hold: the kernel stages it in LDS next to the save grid and the
    model's tables (64 KB per wave in all), at most `SCHEDULE_CAP`."""
    A, nL, H, K1, M1, nk = model.seip_dims
    ga = 1
    while ga < A:
        ga <<= 1
    tpw = max(1, 64 // (ga * H))            # the most trajectories any lane mapping puts in a wave (one lane per age x history)
    per_traj = H * K1 * M1 * nL + A * K1 * 12          # (dose splines sit in LDS as rows of 12: csrc/seip_kernel.hpp kSplRow)
    words = 65536 // (8 if dtype == torch.float64 else 4) - n_save - 16 - tpw * per_traj      # wave groups: their mailbox has LDS of its own
    return int(max(8, min(SCHEDULE_CAP, words // (2 * tpw))))


def _replayed_tangents(model, y0, params, contact, t1, save_ts, dparams, dy0, *, out=None, stats_out=None, dout=None,
                       dtype=torch.float32, **kw) -> BatchResult:
    """Directional derivatives of a SEIP solve, with the outputs of ``dyn_solve_batch_jvp``.

    The SEIP kernels have no tangent planes.  What differentiating through the reference's solve computes -- the
    derivative of the trajectory with the step-size controller held fixed (its decisions are under stop_gradient) -- is
    obtained as: (1) the primal solve, recording its accepted steps (``dyn_solve_batch_record``); (2) ONE batched launch
    of the 2 n_dir B rows ``params +- h_k dparams[:, k]`` (and ``y0 +- h_k dy0[:, k]``) that replays, row by row, the
    step sequence of its primal (``dyn_solve_batch_replay``): on a fixed step sequence the solve is a smooth map, so
    (3) the central difference is its derivative to O(h^2).  ``h_k`` perturbs the most-moved parameter by
    ``_FD_REL_STEP`` of its own magnitude."""
    device = require_gpu()
    P, D = model.param_dim, model.state_dim
    params_t = _dev(params, dtype, device).reshape(-1, P)
    B = params_t.shape[0]
    dp = _dev(dparams, dtype, device)
    if dp.dim() != 3 or dp.shape[0] != B or dp.shape[2] != P:
        raise ValueError(f"dparams must have shape [B={B}, n_dir, P={P}], got {tuple(dp.shape)}")
    n_dir = dp.shape[1]
    y0_t = _dev(y0, dtype, device)
    dy = None
    if dy0 is not None:
        dy = _dev(dy0, dtype, device)
        if dy.dim() == 2:
            dy = dy.unsqueeze(0).expand(B, n_dir, D)
        if tuple(dy.shape) != (B, n_dir, D):
            raise ValueError(f"dy0 must have shape {(n_dir, D)} or {(B, n_dir, D)}")
    cap = schedule_capacity(model, dtype, len(save_ts))
    cdt = float(kw.get("constant_dt", 0.0) or 0.0)
    if cdt > 0.0:
        need = int(np.ceil((float(t1) - float(kw.get("t0", 0.0))) / cdt)) + 1 + len(kw.get("jump_ts", ()) or ())
        if need > cap:
            raise ValueError(f"constant_step_size={cdt} takes {need} steps over the horizon, but a replayed step schedule of this "
                             f"model holds at most {cap} (it is staged in LDS): use a larger step or split the horizon")
    stream = kw.get("stream")
    s = stream if stream is not None else torch.cuda.current_stream(device)
    base = solve_batch(model, y0_t, params_t, contact, t1, save_ts, dtype=dtype, out=out, stats_out=stats_out,
                       record_steps=cap, **kw)
    with torch.cuda.stream(s):          # everything between the two launches runs on the stream the launches use
        # step along direction k: the largest |dp_i| / (|p_i| + floor) becomes _FD_REL_STEP
        rel = (dp.abs() / (params_t.abs().unsqueeze(1) + 1e-3)).amax(dim=2)
        if dy is not None:
            y0b = y0_t if y0_t.dim() == 2 else y0_t.unsqueeze(0).expand(B, D)
            rel = torch.maximum(rel, (dy.abs() / (y0b.abs().unsqueeze(1) + 1e-3)).amax(dim=2))
        h = _FD_REL_STEP[dtype] / rel.clamp_min(1e-30)                                   # [B, n_dir]
        h = torch.where(rel > 0, h, torch.ones_like(h))                                   # a zero direction: derivative 0
        shift = (h.unsqueeze(-1) * dp).reshape(B * n_dir, P)
        rows = params_t.repeat_interleave(n_dir, dim=0)
        p_all = torch.cat([rows + shift, rows - shift], dim=0)                            # [2 n_dir B, P]
        if dy is not None:
            ys = (h.unsqueeze(-1) * dy).reshape(B * n_dir, D)
            yrows = (y0_t if y0_t.dim() == 2 else y0_t.unsqueeze(0).expand(B, D)).repeat_interleave(n_dir, dim=0)
            y_all = torch.cat([yrows + ys, yrows - ys], dim=0)
        else:
            y_all = y0_t.repeat_interleave(n_dir, dim=0).repeat(2, 1) if y0_t.dim() == 2 else y0_t
        leader = torch.arange(B, device=device).repeat_interleave(n_dir).repeat(2)
    kw.pop("constant_dt", None)                                                       # the recording already holds the steps
    pert = solve_batch(model, y_all, p_all, contact, t1, save_ts, dtype=dtype, replay=base.schedule + (leader,), **kw)
    n = B * n_dir
    with torch.cuda.stream(s):
        # A primal that needed more accepted steps than the schedule holds (count -1) cannot be followed, and a perturbed
        # row can fail by itself: such a trajectory reports a non-zero status (max_steps when the schedule overflowed, else
        # the follower's code) and ZERO tangents -- never inf - inf = NaN behind an OK status.  On device, no sync.
        follower = pert.status.reshape(2, B, n_dir).amax(dim=(0, 2))
        overflow = base.schedule[1] < 0
        status = torch.where(base.status != 0, base.status,
                             torch.where(overflow, torch.ones_like(base.status), follower.to(base.status.dtype)))
        base.status.copy_(status)
        dys = (pert.ys[:n] - pert.ys[n:]).reshape(B, n_dir, pert.ys.shape[1], -1) / (2.0 * h).reshape(B, n_dir, 1, 1)
        dys = torch.where((status != 0).reshape(B, 1, 1, 1), torch.zeros_like(dys), dys)
        dys = dys.permute(0, 2, 1, 3).contiguous()
        if dout is not None:
            dout.copy_(dys)
            dys = dout
    return BatchResult(base.ys, base.status, base.n_accept, base.n_reject, base.saved, base.sizes, dys, base.schedule)


def solve_batch_loglik(model: ModelDesc, y0, params, contact, t1: float, save_ts, obs, obs_compartment: int, *,
                       dparams, increments: bool = True, floor: float = 1e-6, dy0=None, t0: float = 0.0,
                       method: str = "tsit5", dtype: torch.dtype = torch.float32, rtol: float = 1e-5,
                       atol: float = 1e-6, max_steps: int = 10**6, constant_dt: float = 0.0,
                       jump_ts: Sequence[float] = (), stream: Optional[torch.cuda.Stream] = None,
                       nuts_tail: Optional[int] = None):
    """Tangent solve with the Poisson observation likelihood fused in (``dyn_solve_batch_loglik``):
    returns ``(logp [B], dlogp [B, n_dir], status, n_accept, n_reject)`` and writes no trajectory.

    ``logp = sum(obs * log(rate) - rate)`` with ``rate = max(v, floor)``, ``v`` the compartment
    ``obs_compartment`` at the save times (``increments=False``, ``len(save_ts)`` rows of ``obs``) or
    its increments between them (``increments=True``, one row fewer); the constant
    ``-lgamma(obs + 1)`` is left to the caller.  ``dparams`` [B, n_dir, P] are the seed directions.

    ``nuts_tail``: host address of a ``dyn_nuts_tail_pack`` blob (``infer/folded.py``) -- the launch then also runs the
    sampler's side of the iteration for the chains it scored (``dyn_solver_opts::nuts_tail``); `SolveError` with code
    ``DYN_ERR_UNSUPPORTED`` when this call cannot carry it (nothing was enqueued).
    """
    device = require_gpu()
    L = _abi.lib()
    D, P, A = model.state_dim, model.param_dim, model.n_age
    params_t = _dev(params, dtype, device).reshape(-1, P)
    B = params_t.shape[0]
    y0_t = _dev(y0, dtype, device)
    batched = y0_t.dim() == 2
    if tuple(y0_t.shape) != ((B, D) if batched else (D,)):
        raise ValueError(f"y0 has shape {tuple(y0_t.shape)}, expected {(D,)} or {(B, D)}")
    contact_t = _dev(contact, dtype, device)
    if contact_t.numel() != A * A:
        raise ValueError(f"contact matrix must have {A}x{A} entries")
    ts_t = _dev(save_ts, dtype, device).reshape(-1)
    n_save = ts_t.shape[0]
    sizes = model.compartment_sizes
    if not 0 <= obs_compartment < len(sizes):
        raise ValueError(f"obs_compartment {obs_compartment} out of range for {model.compartment_names}")
    n_obs = n_save - int(bool(increments))
    obs_t = _dev(obs, dtype, device).reshape(-1)
    if n_obs < 1 or obs_t.numel() != n_obs * sizes[obs_compartment]:
        raise ValueError(f"obs must hold {n_obs} rows of {sizes[obs_compartment]} values "
                         f"({'increments between' if increments else 'values at'} the {n_save} save times), got {obs_t.numel()} values")
    dparams_t = _dev(dparams, dtype, device)
    if dparams_t.dim() != 3 or dparams_t.shape[0] != B or dparams_t.shape[2] != P:
        raise ValueError(f"dparams must have shape [B={B}, n_dir, P={P}], got {tuple(dparams_t.shape)}")
    n_dir = dparams_t.shape[1]
    dy0_t = None
    if dy0 is not None:
        dy0_t = _dev(dy0, dtype, device)
        if tuple(dy0_t.shape) not in ((n_dir, D), (B, n_dir, D)):
            raise ValueError(f"dy0 must have shape {(n_dir, D)} or {(B, n_dir, D)}")
    logp = torch.empty(B, dtype=torch.float64, device=device)
    dlogp = torch.empty((B, n_dir), dtype=torch.float64, device=device)
    stats = torch.empty((3, B), dtype=torch.int32, device=device)
    if B == 0:
        return logp, dlogp, stats[0], stats[1], stats[2]
    jt = np.ascontiguousarray(jump_ts, dtype=np.float64)
    opts = _abi.SolverOptsC(
        _METHODS[method], _DTYPES[dtype], float(rtol), float(atol), int(max_steps), float(constant_dt),
        jt.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) if jt.size else None, int(jt.size))
    _apply_hints(opts)
    if nuts_tail:
        opts.nuts_tail = int(nuts_tail)
    s = stream if stream is not None else torch.cuda.current_stream(device)
    def call():
        return L.dyn_solve_batch_loglik(
            ctypes.byref(model.c()), ctypes.byref(opts), y0_t.data_ptr(), int(batched), params_t.data_ptr(),
            contact_t.data_ptr(), B, float(t0), float(t1), ts_t.data_ptr(), n_save, int(obs_compartment),
            int(bool(increments)), float(floor), obs_t.data_ptr(), n_dir, dparams_t.data_ptr(),
            dy0_t.data_ptr() if dy0_t is not None else None, int(dy0_t is not None and dy0_t.dim() == 3),
            logp.data_ptr(), dlogp.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(),
            ctypes.c_void_p(s.cuda_stream))

    rc = _call_with_jit(call, L, model, dtype, method, n_dir)
    if rc != 0:
        raise SolveError(rc, L.dyn_last_error().decode())
    for t in (y0_t, params_t, contact_t, ts_t, obs_t, dparams_t, dy0_t):
        if t is not None:
            t.record_stream(s)
    return logp, dlogp, stats[0], stats[1], stats[2]
