#!/usr/bin/env python3
"""bench.py -- trajectories/sec of the fused 365-day solve on N MI355X GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg3d136|cfg2|cfg5|seip|seip3|seip83|seip84]
                    [--scaling weak|strong] [--batch B]

A "step" is one pass of the hot path over one batch of synthetic parameter samples: ONE launch
of the fused Tsit5+RHS kernel integrating B trajectories over 365 days with daily dense output
(366 rows x D floats per trajectory written to HBM).  Inputs are resident in HBM before the
timed region.  Default workload = BASELINE.json cfg 3 AS WORDED, `seirs_multi_strain_age_stratified`
with 8 age x 4 strain x 8 immunity bins (the Erlang waning chain of SURVEY.md 8d: D = 360),
16384 samples per GPU.  The same model without the bins axis (the reference's own RHS, D = 136)
is `--workload cfg3d136`; at N = 1 its roofline is reported as a second block, `roofline_d136`.

N > 1: one process per GPU (torchrun), trajectories sharded in contiguous blocks, NO data-path
collective (trajectories are independent); a barrier + synchronize brackets the timed region and
the MAX over ranks is taken.  `--scaling weak` (default): B trajectories per GPU, rank-offset
seeds.  `--scaling strong`: one global batch (`--batch`, default 65536) split by
`sharding.shard_bounds`.  After the timed region rank 0 re-solves every other rank's shard itself
and compares status / step-count / output checksums (`config.shards_match_single_process`).

Rank 0 prints ONE JSON line, including
  roofline     : algorithmic HBM bytes per launch / mean kernel duration (HIP events on the
                 launch stream) against the 8 TB/s HBM3E peak, with the name of the kernel instance
                 that was dispatched (dyn_last_kernel_name) and, when profiles/traffic.json holds a
                 PMC measurement of that same instance, the measured HBM traffic per launch;
                 `measured_on_this_box` (N = 1): this box's device-to-device copy and fill rates on a
                 buffer the size of one launch's output, and the achieved rate as a fraction of each,
  cpu_baseline : the CPU oracle (oracle/, "port") timed on this host's cores on a bounded sample.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SETTLE_LAUNCHES = 12   # untimed launches before the warm-up ones (clock ramp after idle, see measure()) ...
SETTLE_MS = 40.0       # ... or as many as it takes to put this much work in front of them, whichever is more
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy peak ~6290
SEEDS = {"cfg2": 0, "cfg3": 1, "cfg3d136": 1, "cfg5": 5, "seip": 7, "seip3": 7, "seip83": 7, "seip84": 7}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=sorted(SEEDS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--batch", type=int, default=0,
                    help="weak: trajectories per GPU (0 = config default); strong: trajectories of the whole job (0 = 65536)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the side measurements (N=1 only): roofline_d136, cfg2/cfg5/seip, cfg4")
    ap.add_argument("--no-shard-check", action="store_true", help="N>1: skip rank 0's re-solve of the other ranks' shards")
    ap.add_argument("--cpu-sample", type=int, default=0, help="trajectories in the CPU sample (0 = auto)")
    return ap.parse_args()


def effective_cpus() -> int:
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (a GPU box
    exposes every host thread in the mask but grants the job a 16-CPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    env = os.environ.get("DYNODE_BENCH_CPUS")
    return int(env) if env else n


def cpu_baseline(wl, sample: int):
    """Time the oracle (CPU restatement, fp32, OpenMP over trajectories) on a bounded sample."""
    import numpy as np

    from oracle import oracle as O  # checker/baseline only; never on the product path

    cores = effective_cpus()
    m = wl.model
    om = O.Model(m.n_age, m.n_strain, m.has_e, m.has_wane, m.has_c, m.n_wane, m.normalize, m.seasonal, m.has_intro,
                 tuple(m.intro_age_mask), m.n_vax_tiers, m.n_vax_knots, m.family, m.seasonal_vax)
    y0 = wl.y0[:sample] if wl.y0.ndim == 2 else wl.y0
    p = wl.params[:sample]
    O.solve(om, y0[:cores] if wl.y0.ndim == 2 else y0, p[:cores], wl.contact, wl.t1, wl.save_ts,
            dtype=np.float32, n_threads=cores)  # warm-up: thread pool, page faults
    # about 10 s of CPU work: repeated passes over the same sample; median pass time is reported
    times, t_all = [], time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_all < 10.0 and len(times) < 200):
        t = time.perf_counter()
        _, st, _, _ = O.solve(om, y0, p, wl.contact, wl.t1, wl.save_ts, dtype=np.float32, n_threads=cores)
        times.append(time.perf_counter() - t)
    assert int(st.max()) == 0
    med = float(np.median(times))
    # one thread, for a per-core figure: a 1/32 slice of the sample
    n1 = max(sample // 32, 1)
    t = time.perf_counter()
    O.solve(om, y0[:n1] if wl.y0.ndim == 2 else y0, p[:n1], wl.contact, wl.t1, wl.save_ts, dtype=np.float32, n_threads=1)
    single = n1 / (time.perf_counter() - t)
    return {
        "value": sample / med, "unit": "trajectories/s", "cores": cores, "kind": "port", "single_thread_value": single,
        "sample": f"first {sample} trajectories of the same workload, fp32 oracle (oracle/dynode_oracle.c), "
                  f"{cores} OpenMP threads, {len(times)} passes in {sum(times):.1f} s, median pass {med:.3f} s "
                  f"(best {min(times):.3f} s)",
    }


_CFG4 = {}


def cfg4_truth():
    """Tensor-grid quadrature of cfg 4's two-parameter posterior (float64 HIP solves), once per bench run."""
    import torch

    from dynode_amd.infer.checks import GridPosterior
    from dynode_amd.infer.inference import Potential
    from dynode_amd.simulation import odes
    from examples import sir_infer_parameters as ex

    if "truth" not in _CFG4:
        data = ex.synthetic_incidence(100)
        kw = dict(config=ex.get_config(), tf=100, obs_data=data)
        odes.enable_x64(True)
        try:
            pot = Potential(ex.model, kw, 0, torch.device("cuda"))
            _CFG4["truth"] = GridPosterior.from_potential(pot, [torch.linspace(-14.0, 14.0, 1401, dtype=torch.float64),
                                                                torch.linspace(-6.0, 6.0, 601, dtype=torch.float64)]).refined(2)
        finally:
            odes.enable_x64(False)
        _CFG4["kw"] = kw
    return _CFG4["truth"], _CFG4["kw"]


CFG4_TAILS = ((0, 2.0), (0, 4.0))     # z0 > 2: 0.74 % of the mass, 9 % of r0's variance; z0 > 4: 0.13 %


def nuts_side_measurement(chains=128, warmup=1000, samples=1000, fused=True, adaptation="per_chain", seeds=(8675314,), fold=True):
    """cfg 4 at one GPU's share: NUTS on the 2-age SIR (tf=100, Poisson incidence), 1024 / 8 = 128 chains x
    (1000 warm-up + 1000 draws), tree depth 10, checked against tensor-grid quadrature of the 2-parameter posterior
    (`dynode_amd/infer/checks.py`).  Unit of work = one gradient-solve (fused solve + tangents for every chain) per iteration.

    The chains advance independently, so a run lasts as many gradient-solves as its SLOWEST chain needs leapfrogs: the wall
    time is one chain's luck while the time per gradient-solve is the engine's.  The posterior figures of ONE run are a
    realization too (the target's exponential tail along r0's unconstrained coordinate holds 9 % of the variance in 0.74 % of
    the mass), so the run is repeated under ``seeds`` (the first is the reference's, 8675314, src/dynode/infer/inference.py:45)
    and the verdict is the POOLED block: sd ratio over all chains with its across-chain standard error, z statistics of mean
    and variance, Fisher's combination of the runs' KS p-values (thinning from the effective sample size of the squares).
    ``seconds`` / ``gradient_solves`` are the first seed's, ``seconds_median`` over all."""
    import numpy as np
    import torch

    from dynode_amd.infer import checks
    from dynode_amd.infer.inference import MCMCProcess
    from examples import sir_infer_parameters as ex

    truth, kw = cfg4_truth()
    model = ex.model_fused if fused else ex.model
    own = {"adaptation": adaptation, "fold": fold}
    # one-off costs (lazy loading of the kernels, structure discovery of the potential, the caching allocator's first blocks)
    # are paid by a short untimed run of the same program: 16 chains x (20 + 20)
    MCMCProcess(numpyro_model=model, num_warmup=20, num_samples=20, num_chains=16, nuts_max_tree_depth=10,
                progress_bar=False, mcmc_kwargs=own).infer(**kw)
    runs, eps, imm, first = [], [], [], None
    for seed in seeds:
        proc = MCMCProcess(numpyro_model=model, num_warmup=warmup, num_samples=samples, num_chains=chains, nuts_max_tree_depth=10,
                           progress_bar=False, inference_prngkey=int(seed), mcmc_kwargs=own)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mcmc = proc.infer(**kw)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        st = checks.run_statistics(truth, mcmc.nuts.samples.cpu().numpy(), tails=CFG4_TAILS)
        st.update(seed=int(seed), seconds=el, gradient_solves=int(mcmc.nuts.potential_evals), divergences=int(mcmc.nuts.diverging.sum()),
                  mean_leapfrogs_per_transition=float(mcmc.nuts.num_steps.double().mean()))
        runs.append(st)
        eps.append(mcmc.nuts.step_size.cpu())
        imm.append(mcmc.nuts.inverse_mass.cpu())
        first = first or (el, mcmc)
    el, mcmc = first
    pooled = checks.pool_runs(truth, runs)
    _CFG4[("kernels", fused, adaptation)] = (torch.cat(eps), torch.cat(imm))
    secs = sorted(r["seconds"] for r in runs)
    return {"workload": f"cfg4 sir_infer_parameters: NUTS {chains} chains (one GPU's share of 1024) x ({warmup} warm-up + {samples} draws), "
                        f"tree depth 10, warm-up adaptation {adaptation}"
                        + (", Poisson likelihood fused into the solve kernel (examples model_fused)" if fused else
                           ", the reference-shaped model() (simulate -> diff(R) -> Poisson scored in torch)"
                           + (": its likelihood recognised as the solve's fused one (infer/folded.py), one launch per iteration" if fold else
                              " on the general autograd potential (mcmc_kwargs fold=False)")),
            "launches_per_iteration": getattr(mcmc, "launches_per_iteration", None),
            "seconds": el, "seconds_median": secs[len(secs) // 2], "gradient_solves": int(mcmc.nuts.potential_evals),
            "us_per_gradient_solve": 1e6 * el / max(int(mcmc.nuts.potential_evals), 1),
            "transitions_per_s": chains * (warmup + samples) / el, "gradient_solves_per_s": mcmc.nuts.potential_evals / el,
            "chain_gradients_per_s": mcmc.nuts.potential_evals * chains / el,
            "mean_leapfrogs_per_transition": float(mcmc.nuts.num_steps.double().mean()),
            "divergences": int(mcmc.nuts.diverging.sum()),
            "posterior_vs_quadrature": {"pooled_over_sampler_seeds": pooled, "per_seed": runs}}


def nuts_multi_strain_side(sites: int, chains=128, warmup=300, samples=300):
    """NUTS beyond the inference example's shape: every strain's r0 / infectious (/ latent) period of the
    reference's 2-age x 3-strain model (examples/infer_multi_strain.py), six or nine sampled sites, default settings."""
    import torch

    from dynode_amd.infer.inference import MCMCProcess
    from examples import infer_multi_strain as ex_m

    kw = dict(config=ex_m.get_config(sites), tf=120, obs_data=ex_m.synthetic_incidence(120))
    MCMCProcess(numpyro_model=ex_m.model, num_warmup=20, num_samples=20, num_chains=16, nuts_max_tree_depth=8, progress_bar=False).infer(**kw)
    proc = MCMCProcess(numpyro_model=ex_m.model, num_warmup=warmup, num_samples=samples, num_chains=chains, nuts_max_tree_depth=8, progress_bar=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mcmc = proc.infer(**kw)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    post = proc.get_samples()
    truth = dict(zip((f"strains_{k}_r0" for k in range(3)), ex_m.TRUTH["r0s"]))
    return {"workload": f"NUTS on the 2-age x 3-strain model, {sites} sampled sites, {chains} chains x ({warmup} + {samples}), tree depth 8, per-chain adaptation",
            "seconds": el, "sampler": mcmc.sampler, "folded_potential": bool(proc._folded_potential),
            "launches_per_iteration": mcmc.launches_per_iteration, "gradient_solves": int(mcmc.nuts.potential_evals),
            "us_per_gradient_solve": 1e6 * el / max(int(mcmc.nuts.potential_evals), 1),
            "mean_leapfrogs_per_transition": float(mcmc.nuts.num_steps.double().mean()), "divergences": int(mcmc.nuts.diverging.sum()),
            "r0_posterior_mean_sd_truth": {k: [float(post[k].mean()), float(post[k].std()), float(v)] for k, v in truth.items()}}


def nuts_kernel_checks(fused: bool, chains: int, transitions: int = 100, decoupled_starts: int = 0):
    """The calibrated checks of `dynode_amd/infer/checks.py` on the kernels the runs above adapted (per-chain adaptation):

    stationarity  ``chains`` chains started at independent exact posterior draws, each with the (step size, mass matrix) of a
                  random production chain, ``transitions`` transitions, nothing adapting: the end states are i.i.d. posterior
                  draws if the transition kernel is invariant -- KS p-values exactly uniform, tail counts binomial.
    decoupled     the same kernels, ``decoupled_starts`` exact starts each, 1000 draws: the production statistic (pooled sd
                  ratio) without the coupling between a chain's adapted kernel and the state its own warm-up left it in."""
    import numpy as np
    import torch

    from dynode_amd.infer import checks
    from dynode_amd.infer.folded import discover
    from dynode_amd.infer.inference import Potential
    from dynode_amd.infer.nuts import KernelNUTS
    from examples import sir_infer_parameters as ex

    truth, kw = cfg4_truth()
    eps, imm = _CFG4[("kernels", fused, "per_chain")]
    rng = np.random.default_rng(20261004)

    def sampler(seed):
        pot = Potential(ex.model_fused if fused else ex.model, kw, seed, torch.device("cuda"))
        folded = discover(pot, seed=seed) if fused else None      # (not fused: the general autograd potential, the independent check)
        return KernelNUTS(folded if folded is not None else pot.potential_and_grad, max_tree_depth=10, seed=seed)

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = {"stationarity": checks.stationarity(truth, sampler(4242), eps, imm, chains, transitions, rng, tails=CFG4_TAILS)}
    torch.cuda.synchronize()
    out["stationarity"]["seconds"] = time.perf_counter() - t0
    out["stationarity"]["kernels"] = f"{eps.shape[0]} adapted (step size, dense mass matrix) pairs of the per-chain production runs above"
    if decoupled_starts:
        C = eps.shape[0] * decoupled_starts
        pick = torch.arange(eps.shape[0]).repeat(decoupled_starts)
        z0 = torch.from_numpy(truth.draws(C, rng)).cuda()
        res = sampler(777).run(z0, 0, 1000, step_size=eps[pick].cuda(), inverse_mass=imm[pick].cuda())
        st = checks.run_statistics(truth, res.samples.cpu().numpy(), thin=50, tails=CFG4_TAILS)
        out["decoupled_start"] = {"chains": C, "draws": 1000, "divergences": int(res.diverging.sum()),
                                  "pooled": checks.pool_runs(truth, [st])}
    return out


def trajectories_per_wave(model, B: int) -> int:
    """Trajectories that share a wavefront in the mapping a float32 Tsit5 call of B rows actually gets (the library takes a finer
    strain split for batches that fill at most half a wave per SIMD: dyn_trajectories_per_wave_for_batch)."""
    import ctypes

    from dynode_amd import _abi

    o = _abi.SolverOptsC(method=_abi.DYN_TSIT5, dtype=_abi.DYN_F32, rtol=1e-5, atol=1e-6, max_steps=10**6)
    return int(_abi.lib().dyn_trajectories_per_wave_for_batch(ctypes.byref(model.c()), ctypes.byref(o), int(B)))


def kernel_name() -> str:
    from dynode_amd import _abi

    return _abi.lib().dyn_last_kernel_name().decode()


def profiled_traffic(workload: str, B: int, name: str):
    """HBM bytes per launch from the committed PMC passes (tools/profile.sh -> profiles/traffic.json) -- attached only when
    the entry was profiled on the same kernel INSTANCE and the same kernel SOURCES as this run (`_abi.kernel_source_hash`: the
    PMC passes need rocprofv3 around the whole process, so the line cannot sample them itself; a kernel edit that keeps the
    instance's name must not keep its traffic).  Returns (bytes or None, provenance)."""
    from dynode_amd import _abi

    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    here = _abi.kernel_source_hash()
    if not os.path.exists(tpath):
        return None, {"kernel_source_hash": here, "profiled": None}
    rec = json.load(open(tpath)).get(f"{workload}:{B}")
    prov = {"kernel_source_hash": here, "profiled": None if not rec else {k: rec.get(k) for k in ("rev", "kernel_source_hash", "source")}}
    if rec and rec.get("kernel") == name and rec.get("kernel_source_hash") == here:
        return rec["hbm_bytes_per_launch"], prov
    if rec:
        prov["stale"] = "instance differs" if rec.get("kernel") != name else "kernel sources changed since the profile"
    return None, prov


def latency_floor(workload: str, B: int, name: str, stats, tpw: int, rep_log2: int = 0):
    """A launch of at most two waves per SIMD ends when its SLOWEST wave's instruction stream does, however idle the HBM is:
    floor = (vector instructions of the slowest wave) x 4 cycles (a wave issues at most one vector instruction per four
    cycles) / clock.  The slowest wave's count = the profiled mean per wave (SQ_INSTS_VALU / SQ_WAVES of this instance,
    profiles/traffic.json, same kernel sources) x (its loop iterations / the mean over waves), iterations of a wave = the most
    step attempts among its trajectories.  None without a matching profile entry."""
    from dynode_amd import _abi

    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath) or tpw <= 0 or B % tpw:
        return None
    rec = json.load(open(tpath)).get(f"{workload}:{B}")
    if not rec or rec.get("kernel") != name or rec.get("kernel_source_hash") != _abi.kernel_source_hash() or not rec.get("valu_insts_per_wave"):
        return None
    # trajectories that share a wave: B / waves of the profiled launch (replicated small states: fewer than `tpw`)
    g = max(1, int(round(B / rec["waves_per_launch"]))) if rec.get("waves_per_launch") else tpw
    if B % g:
        return None
    iters = (stats[1] + stats[2]).reshape(B // g, g).amax(dim=1).float()
    slowest = rec["valu_insts_per_wave"] * float(iters.max() / iters.mean())
    clock = rec.get("clock_ghz") or 2.4
    return {"latency_floor_ms": slowest * 4.0 / (clock * 1e9) * 1e3, "valu_insts_slowest_wave": slowest, "clock_ghz": clock,
            "iterations_slowest_over_mean": float(iters.max() / iters.mean()),
            "model": "slowest wave's vector instructions x 4 cycles / clock (profiles/traffic.json: SQ_INSTS_VALU / SQ_WAVES of this instance)"}


def measure(wl, dev, steps: int, warmup: int, fence, order_hint_too=True):
    """Resident inputs, `warmup` untimed launches, then `steps` launches with a HIP event pair around each -- the batch in its
    GIVEN order, nothing learned or cached between launches (`solve_batch(order=None)`, the default).  ``order_hint_too``:
    after the timed region, the same batch once more with a caller-supplied queue (most step attempts first, the exact counts
    of the launches just timed): what `dyn_solve_batch_ordered` is worth to a caller who knows its batch.  Never the headline."""
    import numpy as np
    import torch

    from dynode_amd.engine import solve_batch

    m, f32 = wl.model, torch.float32
    y0 = torch.as_tensor(wl.y0, dtype=f32, device=dev)
    params = torch.as_tensor(wl.params, dtype=f32, device=dev)
    contact = torch.as_tensor(wl.contact, dtype=f32, device=dev)
    ts = torch.as_tensor(wl.save_ts, dtype=f32, device=dev)
    out = torch.empty((wl.B, wl.n_save, m.state_dim), dtype=f32, device=dev)
    stats = torch.empty((3, wl.B), dtype=torch.int32, device=dev)

    def step(order=None):
        return solve_batch(m, y0, params, contact, wl.t1, ts, dtype=f32, out=out, stats_out=(stats[0], stats[1], stats[2]), order=order)

    # The GPU needs some 20 ms of work to come up to its sustained clocks after the idle time of process start and input
    # generation (per-launch times of a cold start: 2.82, 2.72, 2.67, 2.60, 2.58, 2.58, 2.50, 2.54 ... ms): SETTLE untimed
    # launches in front of the caller's `warmup`, so that the K timed steps measure the steady state whatever W is.
    # ... counted in WORK, not launches: a 0.13 ms launch (cfg 2) would otherwise be timed 2 ms after the idle gap.
    p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    step()
    p0.record()
    step()
    p1.record()
    torch.cuda.synchronize()
    settle = max(SETTLE_LAUNCHES, int(np.ceil(SETTLE_MS / max(p0.elapsed_time(p1), 1e-3))))
    for _ in range(settle + warmup):
        step()
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for e0, e1 in ev:
        e0.record()
        step()
        e1.record()
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))
    order_info = {"kind": "none: the batch in its given order, no forecast, nothing carried over from earlier launches"}
    if order_hint_too:
        hint = torch.argsort(stats[1] + stats[2], descending=True, stable=True).to(torch.int32)
        for _ in range(max(warmup, 3)):      # (the sort's first call idles the GPU for a moment: let the clocks come back before timing)
            step(order=hint)
        given = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(steps, 10))]
        for e0, e1 in given:
            e0.record()
            step(order=hint)
            e1.record()
        torch.cuda.synchronize()
        order_info["with_caller_supplied_order_ms_per_launch"] = float(np.mean([e0.elapsed_time(e1) for e0, e1 in given]))
        order_info["caller_supplied_order"] = "most step attempts first, exact counts of this batch (dyn_solve_batch_ordered): a side figure, not the headline"
    pipelined = None
    if order_hint_too and 2 * out.numel() * 4 < 0.5 * torch.cuda.get_device_properties(dev).total_memory:
        # What an ensemble driver that issues batch after batch would see: launches alternating between TWO streams (own output
        # buffers), so that the head of launch i + 1 fills the SIMDs the tail of launch i leaves idle (8192 waves over 3072
        # resident slots = 2.67 rounds: the last round is a third empty).  Sustained ms per launch; a side figure, never `value`.
        out2, stats2 = torch.empty_like(out), torch.empty_like(stats)
        streams = (torch.cuda.Stream(), torch.cuda.Stream())
        bufs = ((out, stats), (out2, stats2))
        torch.cuda.synchronize()
        n = 2 * min(steps, 10)

        def both(count):
            for i in range(count):
                o, st_ = bufs[i % 2]
                with torch.cuda.stream(streams[i % 2]):
                    solve_batch(m, y0, params, contact, wl.t1, ts, dtype=f32, out=o, stats_out=(st_[0], st_[1], st_[2]), stream=streams[i % 2])

        both(4)
        torch.cuda.synchronize()
        t_p = time.perf_counter()
        both(n)
        torch.cuda.synchronize()
        pipelined = (time.perf_counter() - t_p) / n * 1e3
        same = bool(torch.equal(out2, out))
        order_info["two_stream_pipelined_ms_per_launch"] = pipelined
        order_info["two_stream_pipelined"] = ("launches alternate between two HIP streams with their own output buffers: launch i + 1's first waves "
                                              "overlap launch i's last round (sustained wall-clock per launch over %d launches; outputs identical: %s)" % (n, same))
        del out2, stats2
        torch.cuda.empty_cache()
    return {"elapsed": elapsed, "kernel_ms": kern_ms, "out": out, "stats": stats, "kernel": kernel_name(), "dispatch_order": order_info,
            "untimed": 2 + settle + warmup}


def checksums(out, stats):
    """Order-sensitive integer / float64 digests of one shard's results (bit-exact comparable across GPUs)."""
    import torch

    idx = torch.arange(1, stats.shape[1] + 1, device=stats.device, dtype=torch.int64)
    return [int(stats[0].sum()), int((stats[1].long() * idx).sum()), int((stats[2].long() * idx).sum()),
            float(out.sum(dtype=torch.float64)), float(out[:, -1].abs().sum(dtype=torch.float64))]


def measured_device_bandwidth(dev, out_bytes: int):
    """the nominal 8 TB/s next to what this box's HBM delivers to the simplest possible kernels --
    a device-to-device copy (one read + one write per byte) and a fill (writes only, the solve's pattern: > 99 % of its
    algorithmic bytes are output rows) of a buffer the size of one launch's output.  torch's copy / fill kernels on the
    current stream, HIP events around 10 repetitions after 3 untimed ones; GB/s of bytes moved."""
    import torch

    n = max(out_bytes // 4, 1 << 26)
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    res = {}
    for name, fn, moved in (("copy", lambda: dst.copy_(src), 8 * n), ("fill", lambda: dst.fill_(1.0), 4 * n)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name + "_GBps"] = moved * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    res["buffer_bytes"] = 4 * n
    del src, dst
    torch.cuda.empty_cache()
    return res


def roofline_block(wl, workload: str, res):
    bytes_traj = wl.bytes_per_trajectory(4)
    achieved = bytes_traj * wl.B / (res["kernel_ms"] * 1e-3) / 1e9  # GB/s per GPU, dominant (only) kernel
    traffic, prov = profiled_traffic(workload, wl.B, res["kernel"])
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_provenance": prov, "kernel": res["kernel"], "kernel_ms": res["kernel_ms"],
            "algorithmic_bytes_per_trajectory": bytes_traj, "trajectories_per_launch": wl.B, "dispatch_order": res["dispatch_order"]}


def describe(wl, workload: str) -> str:
    m = wl.model
    if m.family == 1:
        return ("seip (ode_model.md): {0} ages x {2} immune histories x {3} vaccination tiers x {4} waning states, {1} strains, "
                "D={5}").format(*m.seip_dims[:5], m.state_dim)
    tag = {"cfg3": " = BASELINE cfg 3 (8 age x 4 strain x 8 immunity bins)",
           "cfg3d136": " = the cfg 3 model without the immunity-bins axis (reference RHS)",
           "cfg2": " = BASELINE cfg 2", "cfg5": " = BASELINE cfg 5, one GPU's share"}.get(workload, "")
    return f"{wl.name} ({workload}{tag}): A={m.n_age} S={m.n_strain} W={m.n_wane} D={m.state_dim}"


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from dynode_amd import sharding, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torchrun --nproc-per-node {args.gpus}")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: there is no CPU fallback")
    # DYNODE_BENCH_REHEARSAL=1: rehearse the N > 1 code path on a box with ONE GPU (all ranks on
    # cuda:0, gloo instead of RCCL, which refuses two ranks on one device).  Never set by the driver.
    rehearsal = os.environ.get("DYNODE_BENCH_REHEARSAL") == "1"
    torch.cuda.set_device(0 if rehearsal else local_rank)
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    # DYNODE_BENCH_SINGLE_RANK_GROUP=1 under `torch.distributed.run --nproc-per-node 1`: the RCCL process group and every
    # collective of the N > 1 path with ONE rank -- the backend a one-GPU box can run for real (tests/test_gpu_multirank.py).
    # Never set by the driver.
    grouped = world > 1 or (os.environ.get("DYNODE_BENCH_SINGLE_RANK_GROUP") == "1" and "RANK" in os.environ)
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    cdev = torch.device("cpu") if rehearsal else dev     # where collectives run

    # ---- synthetic workload: weak = same recipe on every rank with a rank-offset seed; strong = one global batch
    gen = synthetic.WORKLOADS[args.workload]
    seed0 = SEEDS[args.workload]

    def shard_of(r: int):
        if args.scaling == "weak":
            return gen(args.batch or gen.__defaults__[0], seed0 + 1000 * r)
        full = gen(args.batch or 65536, seed0)
        lo, hi = sharding.shard_bounds(full.B, r, world)
        full.params = full.params[lo:hi]
        if full.y0.ndim == 2:
            full.y0 = full.y0[lo:hi]
        return full

    wl = shard_of(rank)
    m = wl.model
    B = wl.B
    total_per_step = (args.batch or 65536) if args.scaling == "strong" else B * world

    def fence():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    res = measure(wl, dev, args.steps, args.warmup, fence, order_hint_too=not args.no_extra and world == 1)
    elapsed, kern_ms = res["elapsed"], res["kernel_ms"]
    ok = int(res["stats"][0].max()) == 0
    steps_mean = float((res["stats"][1] + res["stats"][2]).float().mean())
    # lock-step cost of a static grid: a wave's loop runs until the slowest of its trajectories is through
    import ctypes
    from dynode_amd import _abi
    tpw = trajectories_per_wave(m, B)
    wave_iters = None
    if tpw > 0 and B % tpw == 0:
        attempts = (res["stats"][1] + res["stats"][2]).reshape(B // tpw, tpw)
        wave_iters = float(attempts.amax(dim=1).float().mean())

    shards_match, shard_digests = None, None
    ranks_seen, per_rank = 1, None
    if grouped:
        # what the process group itself says (not the environment), and every rank's own clock: a SCALE record checks itself
        ranks_seen = int(dist.get_world_size())
        mine_t = torch.tensor([float(rank), elapsed / args.steps * 1e3, kern_ms, float(torch.cuda.current_device())], dtype=torch.float64, device=cdev)
        every_t = [torch.empty_like(mine_t) for _ in range(world)]
        dist.all_gather(every_t, mine_t)
        per_rank = [{"rank": int(e[0]), "ms_per_step": float(e[1]), "kernel_ms": float(e[2]), "device": int(e[3])} for e in (x.cpu() for x in every_t)]
        t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern_ms = float(t[0]), float(t[1])
        res["kernel_ms"] = kern_ms
        okt = torch.tensor([int(ok)], device=cdev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())
        if not args.no_shard_check:
            # every rank's digests travel to rank 0, which repeats each shard on its own GPU (outside the timed region)
            mine = torch.tensor(checksums(res["out"], res["stats"]), dtype=torch.float64, device=cdev)
            every = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            shard_digests = [[float(v) for v in e.cpu()] for e in every]
            if rank == 0:
                shards_match = True
                for r in range(1, world):
                    del res["out"]
                    again = measure(shard_of(r), dev, 1, 0, torch.cuda.synchronize, order_hint_too=False)
                    res["out"] = again["out"]
                    want = torch.tensor(checksums(again["out"], again["stats"]), dtype=torch.float64)
                    shards_match = shards_match and bool(torch.equal(want, every[r].cpu()))

    if rank == 0:
        line = {
            "metric": "trajectories/sec (365-day solve)",
            "value": total_per_step * args.steps / elapsed,
            "unit": "trajectories/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{describe(wl, args.workload)}, "
                            + (f"{B} parameter samples per GPU" if args.scaling == "weak" else f"{total_per_step} parameter samples split over {world} GPU(s)")
                            + f", 365 days, Tsit5 rtol=1e-5 atol=1e-6, daily save (n_save={wl.n_save}), all compartments saved",
                "trajectories_per_gpu": B,
                "state_dim": m.state_dim,
                "solver": "tsit5",
                "untimed_launches_before_the_timed_region": res["untimed"],
                "mean_steps_per_trajectory": steps_mean,
                # static grid, trajectories dealt to waves in the given order: mean over waves of the most step attempts among
                # a wave's trajectories (= its loop iterations; the wave's other trajectories idle for the difference)
                "trajectories_per_wave": tpw,
                "mean_loop_iterations_per_wave": wave_iters,
                "all_status_ok": ok,
                "parallelism": f"{world} x independent shards, no data-path collective",
                # N > 1: dist.get_world_size() after init_process_group, the backend, and every rank's own timing (value uses the MAX)
                "ranks_seen": ranks_seen,
                "backend": (dist.get_backend() if grouped else None),
                "per_rank": per_rank,
                "shards_match_single_process": shards_match,
                # per rank: [status sum, index-weighted accepted / rejected step counts, float64 sum of the output, of |last row|]
                "shard_digests": shard_digests,
            },
            "roofline": roofline_block(wl, args.workload, res),
        }
        del res
        if world == 1 and not args.no_extra:
            torch.cuda.empty_cache()
            bw = measured_device_bandwidth(dev, line["roofline"]["algorithmic_bytes_per_trajectory"] * B)
            line["roofline"]["measured_on_this_box"] = dict(
                bw, frac_of_copy=line["roofline"]["achieved"] / bw["copy_GBps"], frac_of_fill=line["roofline"]["achieved"] / bw["fill_GBps"],
                note="device-to-device copy and fill of a buffer the size of one launch's output (torch kernels, HIP events): "
                     "`frac` above stays against the nominal 8 TB/s")
        if world == 1 and not args.no_extra and args.workload == "cfg3":
            torch.cuda.empty_cache()
            # the same model without the bins axis, as a second full roofline block (20 launches, one event pair each)
            w2 = synthetic.WORKLOADS["cfg3d136"]()
            r2 = measure(w2, dev, 20, 3, torch.cuda.synchronize)
            line["roofline_d136"] = dict(roofline_block(w2, "cfg3d136", r2), workload=describe(w2, "cfg3d136") + f", B={w2.B}",
                                         trajectories_per_s=w2.B / (r2["kernel_ms"] * 1e-3),
                                         all_status_ok=int(r2["stats"][0].max()) == 0)
            del r2
            # the other single-GPU configs of BASELINE.json, 20 launches each (not the headline value)
            line["other_workloads"] = {}
            for name in ("cfg2", "cfg5", "seip", "seip3", "seip83", "seip84"):
                torch.cuda.empty_cache()
                w2 = synthetic.WORKLOADS[name]()
                r2 = measure(w2, dev, 20, 3, torch.cuda.synchronize)
                blk = roofline_block(w2, name, r2)
                line["other_workloads"][name] = {
                    "workload": describe(w2, name) + f", B={w2.B}", "trajectories_per_s": w2.B / (r2["kernel_ms"] * 1e-3),
                    "ms_per_launch": r2["kernel_ms"], "hbm_frac": blk["frac"], "kernel": blk["kernel"],
                    "with_caller_supplied_order_ms_per_launch": r2["dispatch_order"].get("with_caller_supplied_order_ms_per_launch"),
                    "two_stream_pipelined_ms_per_launch": r2["dispatch_order"].get("two_stream_pipelined_ms_per_launch"),
                    "all_status_ok": int(r2["stats"][0].max()) == 0}
                if w2.model.family == 1:
                    # the bench call runs a "plain" instance (no seasonal terms / introductions / schedules / discontinuity points
                    # compiled in); the general instance of the same shape, for a call that uses any of them, beside it
                    from dynode_amd import engine

                    with engine.dispatch_hints(general_instance=1):
                        r3 = measure(w2, dev, 10, 2, torch.cuda.synchronize, order_hint_too=False)
                    line["other_workloads"][name].update(general_instance_ms_per_launch=r3["kernel_ms"], general_instance_kernel=r3["kernel"])
                    del r3
                if name in ("cfg2", "cfg5"):
                    # launches of one or two waves per SIMD: the bound is the slowest wave's instruction stream, not HBM
                    tpw2 = trajectories_per_wave(w2.model, w2.B)
                    lf = latency_floor(name, w2.B, blk["kernel"], r2["stats"], tpw2)
                    if lf:
                        lf["frac_of_latency_floor"] = lf["latency_floor_ms"] / r2["kernel_ms"]
                        line["other_workloads"][name].update(lf)
                del r2
            torch.cuda.empty_cache()
            # cfg 4.  numpyro's per-chain adaptation under eight sampler seeds, pooled (the fused-likelihood model), four for the
            # program a drop-in user actually has -- the reference-shaped model() (simulate -> diff(R) -> Poisson scored in
            # torch, examples/sir_infer_parameters.py:model; general autograd potential, same sampler kernel) --, the calibrated
            # kernel checks on what those runs adapted, and the pooled-window variant's time
            seeds = (8675314, 1001, 1002, 1003, 1004, 1005, 1006, 1007)
            line["other_workloads"]["cfg4"] = nuts_side_measurement(seeds=seeds)
            line["other_workloads"]["cfg4"]["kernel_checks"] = nuts_kernel_checks(True, 102400, decoupled_starts=4)
            # (round 4: by default the model's torch-written likelihood is recognised and folded -- the first block; the second
            # keeps the general autograd potential, whose posterior checks are the independent ones)
            line["other_workloads"]["cfg4_reference_shaped_model"] = nuts_side_measurement(fused=False, seeds=seeds[:2])
            line["other_workloads"]["cfg4_reference_shaped_model_general_potential"] = nuts_side_measurement(fused=False, seeds=seeds[:4], fold=False)
            line["other_workloads"]["cfg4_reference_shaped_model_general_potential"]["kernel_checks"] = nuts_kernel_checks(False, 25600)
            line["other_workloads"]["cfg4_pooled_adaptation"] = nuts_side_measurement(adaptation="pooled")
            for sites in (6, 9):
                line["other_workloads"][f"nuts_multi_strain_{sites}_sites"] = nuts_multi_strain_side(sites)
        if world == 1 and not args.no_cpu_baseline:
            sample = args.cpu_sample or (1024 if m.family == 1 else 8192 if m.state_dim >= 300 else 16384 if m.state_dim >= 100 else 65536)
            line["cpu_baseline"] = cpu_baseline(wl, min(sample, B))
        print(json.dumps(line), flush=True)
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
