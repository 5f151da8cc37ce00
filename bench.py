#!/usr/bin/env python3
"""bench.py -- trajectories/sec of the fused 365-day solve on N MI355X GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg5|cfg3w8|seip]

A "step" is one pass of the hot path over one batch of synthetic parameter samples: ONE launch
of the fused Tsit5+RHS kernel integrating B trajectories over 365 days with daily dense output
(366 rows x D floats per trajectory written to HBM).  Inputs are resident in HBM before the
timed region.  Default workload = BASELINE.json cfg 3, `seirs_multi_strain_age_stratified`
(8 age x 4 strain SEIRS, D = 136, 16384 samples per GPU), the configuration the north-star
target ("365-day SEIRS trajectories/sec ... % of HBM roofline") is quoted on.

N > 1: one process per GPU (torchrun), trajectories sharded in contiguous blocks with
rank-offset seeds, NO data-path collective (trajectories are independent); a barrier +
synchronize brackets the timed region and the MAX over ranks is taken.  Weak scaling.

Rank 0 prints ONE JSON line, including
  roofline     : algorithmic HBM bytes per launch / mean kernel duration (HIP events on the
                 launch stream) against the 8 TB/s HBM3E peak,
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy peak ~6290


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3", "cfg3w8", "cfg5", "seip", "seip3"])
    ap.add_argument("--batch", type=int, default=0, help="trajectories per GPU (0 = config default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the short cfg2/cfg5 side measurements (N=1 only)")
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


def nuts_side_measurement(dev, chains=128, warmup=300, samples=300, fused=False):
    """cfg 4 in short form: NUTS on the 2-age SIR (tf=100, Poisson incidence), 128 chains on this GPU,
    300 + 300 transitions (BASELINE's cfg 4 runs 1000 + 1000; tools/bench_nuts.py is the full form).
    Unit of work = one gradient-solve (fused solve + tangents for every chain) per sampler iteration."""
    import torch

    from dynode_amd.infer.inference import MCMCProcess
    from examples import sir_infer_parameters as ex

    data = ex.synthetic_incidence(100)
    proc = MCMCProcess(numpyro_model=ex.model_fused if fused else ex.model, num_warmup=warmup, num_samples=samples, num_chains=chains,
                       nuts_max_tree_depth=10, progress_bar=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mcmc = proc.infer(config=ex.get_config(), tf=100, obs_data=data)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    post = proc.get_samples()
    return {"workload": f"cfg4 sir_infer_parameters: NUTS {chains} chains x ({warmup} warm-up + {samples} draws), tree depth 10"
                        + (", Poisson likelihood fused into the solve kernel (examples model_fused)" if fused else ""),
            "seconds": el, "transitions_per_s": chains * (warmup + samples) / el,
            "gradient_solves_per_s": mcmc.nuts.potential_evals / el,
            "chain_gradients_per_s": mcmc.nuts.potential_evals * chains / el,
            "mean_leapfrogs_per_transition": float(mcmc.nuts.num_steps.double().mean()),
            "divergences": int(mcmc.nuts.diverging.sum()),
            "posterior_mean": {k: float(v.mean()) for k, v in post.items()}}


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    from dynode_amd import synthetic
    from dynode_amd.engine import solve_batch

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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    # ---- synthetic workload: same recipe on every rank, rank-offset seed (weak scaling)
    gen = synthetic.WORKLOADS[args.workload]
    base = gen()
    B = args.batch or base.B
    seed = {"cfg2": 0, "cfg3": 1, "cfg3w8": 1, "cfg5": 5, "seip": 7, "seip3": 7}[args.workload] + 1000 * rank
    wl = gen(B, seed)
    m = wl.model
    f32 = torch.float32
    y0 = torch.as_tensor(wl.y0, dtype=f32, device=dev)
    params = torch.as_tensor(wl.params, dtype=f32, device=dev)
    contact = torch.as_tensor(wl.contact, dtype=f32, device=dev)
    ts = torch.as_tensor(wl.save_ts, dtype=f32, device=dev)
    out = torch.empty((B, wl.n_save, m.state_dim), dtype=f32, device=dev)
    stats = torch.empty((3, B), dtype=torch.int32, device=dev)

    def step():
        return solve_batch(m, y0, params, contact, wl.t1, ts, dtype=f32, out=out,
                           stats_out=(stats[0], stats[1], stats[2]))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # HIP events on the launch stream (torch's current stream) around every launch
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e0, e1 in ev:
        e0.record()
        step()
        e1.record()
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))
    ok = int(stats[0].max()) == 0
    steps_mean = float((stats[1] + stats[2]).float().mean())

    if world > 1:
        cdev = torch.device("cpu") if rehearsal else dev
        t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern_ms = float(t[0]), float(t[1])
        okt = torch.tensor([int(ok)], device=cdev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())

    if rank == 0:
        total = B * world * args.steps
        bytes_traj = wl.bytes_per_trajectory(4)
        achieved = bytes_traj * B / (kern_ms * 1e-3) / 1e9  # GB/s per GPU, dominant (only) kernel
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            rec = json.load(open(tpath)).get(f"{args.workload}:{B}")
            if rec:
                traffic = rec["hbm_bytes_per_launch"]
        line = {
            "metric": "trajectories/sec (365-day solve)",
            "value": total / elapsed,
            "unit": "trajectories/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{wl.name} ({args.workload}): A={m.n_age} S={m.n_strain} W={m.n_wane} D={m.state_dim}, "
                            f"{B} parameter samples per GPU, 365 days, Tsit5 rtol=1e-5 atol=1e-6, "
                            f"daily save (n_save={wl.n_save}), all compartments saved",
                "trajectories_per_gpu": B,
                "state_dim": m.state_dim,
                "solver": "tsit5",
                "mean_steps_per_trajectory": steps_mean,
                "all_status_ok": ok,
                "parallelism": f"{world} x independent shards, no data-path collective",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": "dyn::seip_kernel" if m.family == 1 else "dyn::solve_kernel",
                "kernel_ms": kern_ms,
                "algorithmic_bytes_per_trajectory": bytes_traj,
            },
        }
        if world == 1 and not args.no_extra and args.workload == "cfg3":
            # the other single-GPU configs of BASELINE.json, 20 launches each (not the headline value)
            line["other_workloads"] = {}
            for name in ("cfg2", "cfg5", "cfg3w8", "seip", "seip3"):
                w2 = synthetic.WORKLOADS[name]()
                a = [torch.as_tensor(x, dtype=f32, device=dev) for x in (w2.y0, w2.params, w2.contact, w2.save_ts)]
                o2 = torch.empty((w2.B, w2.n_save, w2.model.state_dim), dtype=f32, device=dev)
                st2 = torch.empty((3, w2.B), dtype=torch.int32, device=dev)
                run = lambda: solve_batch(w2.model, a[0], a[1], a[2], w2.t1, a[3], dtype=f32, out=o2,
                                          stats_out=(st2[0], st2[1], st2[2]))
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 20
                gbs = w2.bytes_per_trajectory(4) * w2.B / (ms * 1e-3) / 1e9
                line["other_workloads"][name] = {
                    "workload": (f"{w2.name}: A={w2.model.n_age} S={w2.model.n_strain} W={w2.model.n_wane} D={w2.model.state_dim}, B={w2.B}"
                                 if w2.model.family == 0 else
                                 "seip (ode_model.md): {0} ages x {2} immune histories x {3} vaccination tiers x {4} waning states, "
                                 "{1} strains, D={5}, B={6}".format(*w2.model.seip_dims[:5], w2.model.state_dim, w2.B)),
                    "trajectories_per_s": w2.B / (ms * 1e-3), "ms_per_launch": ms, "hbm_frac": gbs / HBM_PEAK_GBS,
                    "all_status_ok": int(st2[0].max()) == 0}
                del o2
            line["other_workloads"]["cfg4"] = nuts_side_measurement(dev)
            line["other_workloads"]["cfg4_fused_likelihood"] = nuts_side_measurement(dev, fused=True)
        if world == 1 and not args.no_cpu_baseline:
            sample = args.cpu_sample or (1024 if m.family == 1 else 16384 if m.state_dim >= 100 else 65536)
            line["cpu_baseline"] = cpu_baseline(wl, min(sample, B))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
