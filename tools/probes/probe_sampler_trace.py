"""dev probe: the sampler kernel's state after every single launch, written to an .npz -- run once per library build
(`--lib path/to/other/libdynode_hip.so`) and compare the two traces with `--compare a.npz b.npz`: the first launch at which
any field of any chain differs beyond rounding names a defect (two builds that differ only in summation order stay within
1e-9 of each other for hundreds of launches before the chaos of the sampler takes over).
    python tools/probes/probe_sampler_trace.py gauss|mapped out.npz [--lib LIB]"""
import os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
FIELDS = ("z_eval", "z", "eps", "eps_avg", "e0", "s_weight", "weight", "r_half", "r_sum", "s_rsum", "imm", "mm_sqrt", "wf_mean", "wf_m2",
          "it", "depth", "leaf", "n_prop", "right", "s_turn", "s_div", "rng_ctr", "zl", "zr", "zp", "rl", "rr", "gl", "gr", "gp", "up", "u", "da_xbar")


def compare(a, b):
    A, B = np.load(a), np.load(b)
    n = min(A["it"].shape[0], B["it"].shape[0])
    for k in range(n):
        worst = (0.0, None)
        for f in FIELDS:
            x, y = A[f][k].astype(np.float64), B[f][k].astype(np.float64)
            both_nan = np.isnan(x) & np.isnan(y)
            same_inf = np.isinf(x) & (x == y)
            d = np.where(both_nan | same_inf, 0.0, np.abs(x - y) / (1.0 + np.abs(x)))
            d = np.nan_to_num(d, nan=np.inf)
            if d.max() > worst[0]:
                worst = (float(d.max()), f)
        if worst[0] > 1e-7:
            print(f"launch {k}: field {worst[1]} differs by {worst[0]:.3e} (relative); iteration counters {A['it'][k].tolist()} / {B['it'][k].tolist()}")
            for f in FIELDS:
                x, y = A[f][k].astype(np.float64), B[f][k].astype(np.float64)
                d = np.nan_to_num(np.abs(x - y) / (1.0 + np.abs(x)), nan=0.0, posinf=0.0)
                if d.max() > 1e-7:
                    ch = np.unique(np.argwhere(d > 1e-7)[:, 0])
                    print(f"   {f}: max {d.max():.3e}, chains {ch.tolist()[:8]}")
            return
        if k % 100 == 0:
            print(f"launch {k}: worst {worst[0]:.2e} ({worst[1]})")
    print(f"{n} launches, no difference beyond 1e-7")


def main():
    if sys.argv[1] == "--compare":
        return compare(sys.argv[2], sys.argv[3])
    kind, out = sys.argv[1], sys.argv[2]
    import torch
    from dynode_amd import _abi
    if "--lib" in sys.argv:
        _abi.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
    from dynode_amd.infer import nuts as N
    trace = {f: [] for f in FIELDS}
    limit = 1200

    def monitor(S):
        if len(trace["it"]) < limit:
            for f in FIELDS:
                trace[f].append(S[f].detach().cpu().numpy().copy())

    orig = N.KernelNUTS.__init__

    def init(self, *a, **kw):
        kw["use_graph"] = False
        kw["block"] = 1
        orig(self, *a, **kw)
        self.unroll, self.monitor, self.recheck_blocks = 1, monitor, ()
    N.KernelNUTS.__init__ = init
    dev = torch.device("cuda")
    if kind == "gauss":
        g = torch.Generator().manual_seed(5)
        A = torch.randn(12, 12, generator=g, dtype=torch.float64)
        cov = (A @ A.T / 12.0 + torch.diag(torch.linspace(0.2, 2.0, 12, dtype=torch.float64))).to(dev)
        prec = torch.linalg.inv(cov)

        def pg(z):
            gr = z @ prec
            return 0.5 * (z * gr).sum(-1), gr
        z0 = torch.randn(8, 12, generator=g, dtype=torch.float64).to(dev)
        N.KernelNUTS(pg, max_tree_depth=6, seed=2).run(z0, num_warmup=150, num_samples=10)
    else:
        from dynode_amd.infer.inference import MCMCProcess
        from examples import infer_multi_strain as ex_m
        torch.manual_seed(0)
        kw = dict(config=ex_m.get_config(9), tf=120, obs_data=ex_m.synthetic_incidence(120))
        MCMCProcess(numpyro_model=ex_m.model, num_warmup=150, num_samples=10, num_chains=8, nuts_max_tree_depth=6, progress_bar=False).infer(**kw)
    np.savez(out, **{f: np.stack(v) for f, v in trace.items()})
    print(kind, "launches recorded:", len(trace["it"]), "library:", _abi.LIB_PATH)


main()
