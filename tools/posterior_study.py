#!/usr/bin/env python3
"""cfg 4 posterior study: is the sampler's low standard deviation bias or a realization?

Measurements on the reference's inference example (examples/sir_infer_parameters.py; reference
src/dynode/infer/inference.py:149-163, examples/sir_infer_parameters.py:21-58), all against tensor-grid quadrature of the
two-parameter posterior (float64 solves), with the tools of dynode_amd/infer/checks.py:

  control     i.i.d. draws FROM the quadrature posterior, many times: what the statistics of a run's 12,800 thinned draws
              (KS p, sd ratio) look like when nothing is wrong.
  seeds       production runs, 128 x (1000 + 1000), S sampler seeds x variants (model_fused / model with numpyro's per-chain
              adaptation; model_fused with pooled windows; model_fused on the torch-op sampler `GraphNUTS` -- an independent
              implementation of the same algorithm with torch's generator instead of Philox): pooled across-chain statistics,
              and the tail occupancy of the chains by quartile of their own adapted z0 variance.
  stationary  the exactly calibrated test of the transition kernel (checks.stationarity).
  decoupled   the production kernels started at independent exact draws, 1000 draws each: the production statistic without
              the coupling between a chain's adapted kernel and the state its own warm-up left it in.
  long        128 x (1000 + 10000).

Writes one JSON (default gpurun_out/posterior_study.json); docs/perf-log.md (round 4) has the findings.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TAILS = ((0, 1.0), (0, 2.0), (0, 3.0), (0, 4.0))          # thresholds in the unconstrained r0 coordinate z0
VARIANTS = {"model_fused": ("model_fused", {}), "model": ("model", {}), "model_fused_pooled": ("model_fused", {"adaptation": "pooled"}),
            "model_fused_graph": ("model_fused", {"sampler": "graph"})}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "posterior_study.json"))
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--variants", default="model_fused,model,model_fused_pooled")
    ap.add_argument("--stationary-chains", type=int, default=102400)
    ap.add_argument("--stationary-steps", type=int, default=100)
    ap.add_argument("--stationary-kinds", default="model_fused,model")
    ap.add_argument("--control-reps", type=int, default=400)
    ap.add_argument("--long-samples", type=int, default=10000)
    ap.add_argument("--skip", default="", help="comma list of: control,seeds,stationary,decoupled,long")
    args = ap.parse_args()
    skip = set(filter(None, args.skip.split(",")))
    import torch

    import bench
    from dynode_amd.infer import checks
    from dynode_amd.infer.folded import discover
    from dynode_amd.infer.inference import MCMCProcess, Potential
    from dynode_amd.infer.nuts import KernelNUTS
    from examples import sir_infer_parameters as ex

    truth, kw = bench.cfg4_truth()
    names = truth.names
    res = {"quadrature": {"mean": truth.mean, "sd": truth.sd, "tail_mass": {f"z0>{t:g}": truth.tail_mass(k, t) for k, t in TAILS}}}
    print("[study] quadrature", res["quadrature"], flush=True)
    rng = np.random.default_rng(20261004)

    def flush():
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f)

    def sampler_for(kind, seed):
        pot = Potential(getattr(ex, kind), kw, seed, torch.device("cuda"))
        folded = discover(pot, seed=seed)
        return KernelNUTS(folded if folded is not None else pot.potential_and_grad, max_tree_depth=10, seed=seed)

    if "control" not in skip:
        res["control"] = checks.iid_control(truth, args.control_reps, 12800, rng)
        print("[study] control", res["control"], flush=True)
        flush()

    adapted = {}
    if "seeds" not in skip:
        res["seeds"] = {}
        for variant in args.variants.split(","):
            kind, mk = VARIANTS[variant]
            runs, eps_all, imm_all, occ, v00, z0_all = [], [], [], [], [], []
            for s in range(args.seeds):
                seed = 8675314 if s == 0 else 1000 + s
                proc = MCMCProcess(numpyro_model=getattr(ex, kind), num_warmup=1000, num_samples=1000, num_chains=128, nuts_max_tree_depth=10,
                                   progress_bar=False, inference_prngkey=seed, mcmc_kwargs=dict(mk))
                torch.cuda.synchronize()
                t = time.time()
                mc = proc.infer(**kw)
                torch.cuda.synchronize()
                el = time.time() - t
                st = checks.run_statistics(truth, mc.nuts.samples.cpu().numpy(), tails=TAILS)
                st.update(seed=seed, seconds=el, gradient_solves=int(mc.nuts.potential_evals), divergences=int(mc.nuts.diverging.sum()),
                          step_size_median=float(mc.nuts.step_size.median()))
                z0_all.append(mc.nuts.samples[:, :, 0].cpu().numpy())
                occ.append((mc.nuts.samples[:, :, 0] > 2.0).double().mean(1).cpu().numpy())
                v00.append(mc.nuts.inverse_mass[:, 0, 0].cpu().numpy())
                runs.append(st)
                eps_all.append(mc.nuts.step_size.cpu())
                imm_all.append(mc.nuts.inverse_mass.cpu())
                print(f"[study] {variant} seed {seed}: {el:.2f} s, div {st['divergences']}, tails {st['tail_ratio']}, " + ", ".join(
                    f"{n[10:]}: sd {st[n]['sd_ratio']:.4f} core {st[n]['core_sd_ratio']:.4f} ks {st[n]['ks_p']:.3f} (thin {st[n]['thin']}) zc {st[n]['chain_mean_z']:.2f}/{st[n]['chain_var_z']:.2f}/{st[n]['chain_core_z']:.2f}"
                    for n in names), flush=True)
            adapted[variant] = (torch.cat(eps_all), torch.cat(imm_all))
            pooled = checks.pool_runs(truth, runs)
            occ, v00 = np.concatenate(occ), np.concatenate(v00)
            order, q = np.argsort(v00), len(v00) // 4
            pooled["tail2_ratio_by_quartile_of_chain_z0_variance"] = [float(occ[order[i * q:(i + 1) * q]].mean() / truth.tail_mass(0, 2.0)) for i in range(4)]
            pooled["chain_z0_variance_quartile_means"] = [float(v00[order[i * q:(i + 1) * q]].mean()) for i in range(4)]
            pooled["chains_never_beyond_z0_2"] = float((occ == 0).mean())
            pooled["excursions"] = {f"z0>{t:g}": checks.excursions(np.concatenate(z0_all), t) for t in (2.0, 4.0)}
            res["seeds"][variant] = {"pooled": pooled, "runs": runs}
            print(f"[study] {variant} pooled", pooled, flush=True)
            flush()

    def kernels(kind):
        if kind not in adapted:      # (seeds skipped: adapt one run here)
            mc = MCMCProcess(numpyro_model=getattr(ex, kind), num_warmup=1000, num_samples=10, num_chains=128, nuts_max_tree_depth=10,
                             progress_bar=False).infer(**kw)
            adapted[kind] = (mc.nuts.step_size.cpu(), mc.nuts.inverse_mass.cpu())
        return adapted[kind]

    if "stationary" not in skip:
        res["stationary"] = {}
        for kind in args.stationary_kinds.split(","):
            eps, imm = kernels(kind)
            C = args.stationary_chains if kind == "model_fused" else min(args.stationary_chains, 25600)
            t = time.time()
            rep = checks.stationarity(truth, sampler_for(kind, 4242), eps, imm, C, args.stationary_steps, rng, tails=TAILS,
                                      at=(1, 2, 5, 10, 20, 50, 100, 150, 200, args.stationary_steps))
            rep["seconds"] = time.time() - t
            res["stationary"][kind] = rep
            print(f"[study] stationary {kind}: {C} chains x {args.stationary_steps}: {rep['seconds']:.1f} s, div {rep['divergences']}, last:", rep["last"], flush=True)
            torch.cuda.empty_cache()
            flush()

    if "decoupled" not in skip:
        res["decoupled"] = {}
        eps, imm = kernels("model_fused")
        reps = 8
        pick = torch.arange(eps.shape[0]).repeat(reps)
        z0 = torch.from_numpy(truth.draws(pick.numel(), rng)).cuda()
        out = sampler_for("model_fused", 777).run(z0, 0, 1000, step_size=eps[pick].cuda(), inverse_mass=imm[pick].cuda())
        st = checks.run_statistics(truth, out.samples.cpu().numpy(), thin=50, tails=TAILS)
        rep = {"chains": int(pick.numel()), "draws": 1000, "divergences": int(out.diverging.sum()), "pooled": checks.pool_runs(truth, [st])}
        occ = (out.samples[:, :, 0] > 2.0).double().mean(1).cpu().numpy().reshape(reps, -1).mean(0)
        v00 = imm[:, 0, 0].numpy()
        order, q = np.argsort(v00), len(v00) // 4
        rep["tail2_ratio_by_quartile_of_kernel_z0_variance"] = [float(occ[order[i * q:(i + 1) * q]].mean() / truth.tail_mass(0, 2.0)) for i in range(4)]
        rep["excursions"] = {f"z0>{t:g}": checks.excursions(out.samples[:, :, 0].cpu().numpy(), t) for t in (2.0, 4.0)}
        res["decoupled"]["model_fused"] = rep
        print("[study] decoupled model_fused:", rep, flush=True)
        del out
        torch.cuda.empty_cache()
        flush()

    if "long" not in skip:
        res["long"] = {}
        for seed in (8675314, 1001):
            proc = MCMCProcess(numpyro_model=ex.model_fused, num_warmup=1000, num_samples=args.long_samples, num_chains=128, nuts_max_tree_depth=10,
                               progress_bar=False, inference_prngkey=seed)
            t = time.time()
            mc = proc.infer(**kw)
            torch.cuda.synchronize()
            st = checks.strip(checks.run_statistics(truth, mc.nuts.samples.cpu().numpy(), tails=TAILS))
            st.update(seconds=time.time() - t, divergences=int(mc.nuts.diverging.sum()))
            res["long"][str(seed)] = st
            print(f"[study] long seed {seed}: tails {st['tail_ratio']} " + ", ".join(
                f"{n[10:]}: sd {st[n]['sd_ratio']:.4f} zc {st[n]['chain_mean_z']:.2f}/{st[n]['chain_var_z']:.2f}" for n in names), flush=True)
            flush()
    flush()
    print("[study] wrote", args.out, flush=True)


if __name__ == "__main__":
    main()
