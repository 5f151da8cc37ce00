"""dev probe: every launch of the sampler kernel of a run repeated, from the same state, by ANOTHER build of the library
(`--lib`): the two results must agree to rounding in every field, launch by launch, through warm-up windows and their
Cholesky factors, transition ends and recorded draws (a run-against-run comparison drifts apart chaotically after a few
hundred launches; this one does not).  `--wall`: the potential is +inf (and its gradient NaN) beyond |z_i| = 2.5, the
non-finite branch of the state machine; `--chains N`, `--depth N`.
    python tools/probes/probe_sampler_step_ab.py --lib tools/probes/_lib_OLD.so [--wall] [--chains 7] [--depth 10] [dim]"""
import ctypes, os, sys
import numpy as np, torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dynode_amd import _abi
from dynode_amd.infer import nuts as N

other = ctypes.CDLL(os.path.abspath(sys.argv[sys.argv.index("--lib") + 1]))
D = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 12
made = []
Orig = _abi.NutsStateC


def factory():
    made.append(Orig())
    return made[-1]


_abi.lib()                     # (its argument types name the class: load before the factory stands in for it)
_abi.NutsStateC = factory
dev = torch.device("cuda")
g = torch.Generator().manual_seed(5)
A = torch.randn(D, D, generator=g, dtype=torch.float64)
cov = (A @ A.T / D + torch.diag(torch.linspace(0.2, 2.0, D, dtype=torch.float64))).to(dev)
prec = torch.linalg.inv(cov)


WALL = "--wall" in sys.argv
CHAINS = int(sys.argv[sys.argv.index("--chains") + 1]) if "--chains" in sys.argv else 8
DEPTH = int(sys.argv[sys.argv.index("--depth") + 1]) if "--depth" in sys.argv else 6


def pg(z):
    gr = z @ prec
    u = 0.5 * (z * gr).sum(-1)
    if WALL:
        out = (z.abs() > 2.5).any(-1)
        u = torch.where(out, torch.full_like(u, float("inf")), u)
        gr = torch.where(out[:, None], torch.full_like(gr, float("nan")), gr)
    return u, gr


state = {"prev": None, "k": 0, "worst": 0.0, "bad": 0, "events": {"window_end": 0, "transition_end": 0, "draw": 0}}
SKIP = ("pool", "pool_ro", "pend")


def monitor(S):
    prev = state["prev"]
    if prev is not None:
        T = {k: v.clone() for k, v in prev.items()}
        u, gr = pg(T["z_eval"])
        T["u_new"].copy_(u); T["g_new"].copy_(gr)
        st = Orig()
        ctypes.memmove(ctypes.byref(st), ctypes.byref(made[-1]), ctypes.sizeof(st))
        for name in _abi.NUTS_POINTER_FIELDS:
            setattr(st, name, T[name].data_ptr())
        rc = other.dyn_nuts_advance(ctypes.byref(st), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
        torch.cuda.synchronize()
        if bool((T["wi"] != prev["wi"]).any()): state["events"]["window_end"] += int((T["wi"] != prev["wi"]).sum())
        state["events"]["transition_end"] += int((T["it"] != prev["it"]).sum())
        for k in S:
            if k in SKIP: continue
            x, y = T[k].double(), S[k].double()
            same = (x == y) | (x.isnan() & y.isnan())
            d = torch.where(same, torch.zeros_like(x), (x - y).abs() / (1.0 + x.abs()))
            d = torch.nan_to_num(d, nan=float("inf"))
            m = float(d.max()) if d.numel() else 0.0
            state["worst"] = max(state["worst"], m)
            if m > 1e-9:
                state["bad"] += 1
                if state["bad"] <= 12:
                    ch = torch.unique(torch.nonzero(d > 1e-9)[:, 0]).tolist()[:6]
                    print(f"launch {state['k']}: {k} differs by {m:.3e}, chains {ch}, it {prev['it'].tolist()}", flush=True)
    state["prev"] = {k: v.clone() for k, v in S.items()}
    state["k"] += 1


orig = N.KernelNUTS.__init__


def init(self, *a, **kw):
    kw["use_graph"] = False; kw["block"] = 1
    orig(self, *a, **kw)
    self.unroll, self.monitor, self.recheck_blocks = 1, monitor, ()


N.KernelNUTS.__init__ = init
z0 = (0.3 * torch.randn(CHAINS, D, generator=g, dtype=torch.float64)).to(dev)
res = N.KernelNUTS(pg, max_tree_depth=DEPTH, seed=2).run(z0, num_warmup=220, num_samples=12)
print(f"dim {D}, {CHAINS} chains, depth {DEPTH}, wall {WALL}, divergent draws {int(res.diverging.sum())}: {state['k']} launches, worst relative difference {state['worst']:.3e}, fields beyond 1e-9: {state['bad']}, events {state['events']}")
