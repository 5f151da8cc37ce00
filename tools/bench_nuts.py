#!/usr/bin/env python3
"""cfg 4 (BASELINE.json): NUTS on the 2-age SIR, chains sharded over GPUs, one process per GPU.

    python tools/bench_nuts.py [--chains 1024] [--warmup 1000] [--samples 1000]
    torchrun --nproc-per-node N tools/bench_nuts.py ...      (chains / N per GPU, RCCL gather at the end)

Reports transitions/s, gradient-solves/s (the unit of work: one fused solve+tangent launch for
all local chains) and the KS p-values of the pooled marginals against grid quadrature.
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--sampler", default="kernel", choices=["kernel", "graph", "eager", "ensemble"])
    ap.add_argument("--adaptation", default="pooled", choices=["per_chain", "pooled"])
    ap.add_argument("--target-accept", type=float, default=0.8)
    ap.add_argument("--fused-likelihood", action="store_true", help="score the observations inside the solve kernel (examples model_fused)")
    ap.add_argument("--no-fold", action="store_true", help="keep the general torch-autograd potential (infer/folded.py off)")
    args = ap.parse_args()
    import numpy as np, torch, torch.distributed as dist
    from scipy import stats
    from dynode_amd import sharding
    from dynode_amd.infer.inference import MCMCProcess, Potential, log_posterior_grid
    from dynode_amd.simulation import odes
    from examples import sir_infer_parameters as ex

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    data = ex.synthetic_incidence(100)
    proc = MCMCProcess(numpyro_model=ex.model_fused if args.fused_likelihood else ex.model, num_warmup=args.warmup, num_samples=args.samples, num_chains=args.chains,
                       nuts_max_tree_depth=args.depth, progress_bar=(rank == 0),
                       mcmc_kwargs={"sampler": args.sampler, "adaptation": args.adaptation, "fold": not args.no_fold},
                       nuts_kwargs={"target_accept_prob": args.target_accept})
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    t0 = time.perf_counter()
    mcmc = proc.infer(config=ex.get_config(), tf=100, obs_data=data)
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    el = time.perf_counter() - t0
    post = proc.get_samples(group_by_chain=True, gather=world > 1)            # RCCL gather of the posterior
    if rank == 0:
        odes.enable_x64(True)
        pot = Potential(ex.model, dict(config=ex.get_config(), tf=100, obs_data=data), 0, torch.device("cuda"))
        from dynode_amd.infer.inference import marginal_cdfs_by_quadrature
        z0 = torch.linspace(-14.0, 14.0, 1401, dtype=torch.float64); z1 = torch.linspace(-6.0, 6.0, 1001, dtype=torch.float64)
        (g0, c0, m0), (g1, c1, m1) = marginal_cdfs_by_quadrature(pot, [z0, z1]); odes.enable_x64(False)   # unconstrained-space grid
        ks, quad = {}, {}
        for name, grid, cdf, pdf in (("strains_0_r0", g0, c0, m0), ("strains_0_infectious_period", g1, c1, m1)):
            thin = post[name][:, ::20].reshape(-1).cpu().numpy()
            ks[name] = float(stats.kstest(thin, lambda x: np.interp(x, grid, cdf)).pvalue)
            mean = float((grid * pdf).sum())
            quad[name] = {"mean": mean, "sd": float(np.sqrt(((grid - mean) ** 2 * pdf).sum())),
                          "sample_mean": float(post[name].mean()), "sample_sd": float(post[name].std()),
                          "max_cdf_gap": float(stats.kstest(thin, lambda x: np.interp(x, grid, cdf)).statistic)}
            allx = post[name].reshape(-1).cpu().numpy()
            for k in (-3.0, -2.0, 2.0, 3.0):                    # tail masses beyond mean + k sd: quadrature vs draws
                cut = mean + k * quad[name]["sd"]
                pq = float(np.interp(cut, grid, cdf))
                quad[name][f"tail_{k:+.0f}sd"] = [pq if k < 0 else 1.0 - pq, float((allx < cut).mean() if k < 0 else (allx > cut).mean())]
        n_trans = args.chains * (args.warmup + args.samples)
        print(json.dumps({
            "workload": "cfg4 sir_infer_parameters: NUTS, 2-age SIR, tf=100, Poisson incidence",
            "sampler": args.sampler, "adaptation": args.adaptation, "fused_likelihood": args.fused_likelihood, "target_accept": args.target_accept, "n_gpus": world, "chains": args.chains, "warmup": args.warmup, "samples": args.samples,
            "seconds": el, "transitions_per_s": n_trans / el,
            "gradient_solves_per_s_per_gpu": mcmc.nuts.potential_evals / el,
            "chain_gradient_evals_per_s": mcmc.nuts.potential_evals * (args.chains / world) * world / el,
            "mean_leapfrogs_per_transition": float(mcmc.nuts.num_steps.double().mean()),
            "divergences_rank0": int(mcmc.nuts.diverging.sum()), "ks_pvalues_vs_quadrature": ks, "moments_vs_quadrature": quad,
            "posterior_mean": {k: float(v.mean()) for k, v in post.items()},
        }), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
