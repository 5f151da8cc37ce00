"""The N > 1 code paths on the GPU box (one GPU): `bench.py --gpus 2` as the driver launches it (torch.distributed.run, one
process per rank) and `MCMCProcess` with its chains sharded over two ranks.  Both ranks share cuda:0 here, so the collectives
run over gloo (`DYNODE_BENCH_REHEARSAL=1`; RCCL refuses two ranks on one device) -- everything else is the code an 8-GPU node
runs: shard bounds, rank-offset seeds, per-rank solves, the barrier + max-over-ranks timing, the digests rank 0 re-derives.
"""

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import helpers as H
from dynode_amd import sharding, synthetic

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench_two_ranks(*args):
    env = dict(os.environ, DYNODE_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(H.ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", *args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=H.ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line
    return json.loads(lines[0])


def _digest(out, stats):
    idx = torch.arange(1, stats.shape[1] + 1, device=stats.device, dtype=torch.int64)
    return [float(stats[0].sum()), float((stats[1].long() * idx).sum()), float((stats[2].long() * idx).sum()),
            float(out.sum(dtype=torch.float64)), float(out[:, -1].abs().sum(dtype=torch.float64))]


def test_bench_two_ranks_weak_scaling_rehearsal():
    """Weak scaling: every rank solves its own B draws (rank-offset seed); rank 0 repeats the other rank's shard and compares."""
    d = _bench_two_ranks("--scaling", "weak", "--workload", "cfg3d136")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["config"]["shards_match_single_process"] is True and d["config"]["all_status_ok"] is True
    assert len(d["config"]["shard_digests"]) == 2 and d["config"]["shard_digests"][0] != d["config"]["shard_digests"][1]
    assert d["value"] > 0 and d["config"]["trajectories_per_gpu"] == 16384
    # value = trajectories of BOTH ranks over the max-over-ranks wall clock
    assert abs(d["value"] - 2 * 16384 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-9
    # the line checks itself: the process group's own size and backend, and every rank's clock (the line's is their maximum)
    assert d["config"]["ranks_seen"] == 2 and d["config"]["backend"] == "gloo"
    per_rank = d["config"]["per_rank"]
    assert [r["rank"] for r in per_rank] == [0, 1] and max(r["ms_per_step"] for r in per_rank) <= d["ms_per_step"] * (1 + 1e-9)


def test_bench_rccl_process_group_with_one_rank():
    """The backend the 8-GPU node runs, as far as a one-GPU box can: `bench.py` under `torch.distributed.run` with ONE rank and
    `DYNODE_BENCH_SINGLE_RANK_GROUP=1` initialises the RCCL ("nccl") process group on its device and runs every collective of
    the N > 1 path on it -- the barrier around the timed region, the all-gather of the ranks' own clocks, the max / min
    all-reduces, the all-gather of the shard digests (float64 on the device)."""
    env = dict(os.environ, DYNODE_BENCH_SINGLE_RANK_GROUP="1", MASTER_ADDR="127.0.0.1")
    env.pop("DYNODE_BENCH_REHEARSAL", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(H.ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-extra", "--workload", "cfg3d136"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=H.ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["backend"] == "nccl" and d["config"]["ranks_seen"] == 1
    assert [q["rank"] for q in d["config"]["per_rank"]] == [0] and d["config"]["per_rank"][0]["ms_per_step"] == pytest.approx(d["ms_per_step"], rel=1e-9)
    assert len(d["config"]["shard_digests"]) == 1 and d["config"]["all_status_ok"] is True and d["value"] > 0


def test_bench_two_ranks_strong_scaling_equals_one_process():
    """Strong scaling: ONE global batch of 65536 draws split by shard_bounds.  Each rank's digests must be those of the same
    rows of a single-process solve of the whole batch, bit for bit (the kernel is deterministic and batch-position invariant,
    so any difference is a sharding bug)."""
    from dynode_amd.engine import solve_batch

    d = _bench_two_ranks("--scaling", "strong", "--workload", "cfg2", "--batch", "65536")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["shards_match_single_process"] is True
    assert d["config"]["trajectories_per_gpu"] == 32768
    wl = synthetic.WORKLOADS["cfg2"](65536, 0)                       # bench.SEEDS["cfg2"] == 0
    r = solve_batch(wl.model, wl.y0, wl.params, wl.contact, wl.t1, wl.save_ts, dtype=torch.float32)
    stats = torch.stack([r.status, r.n_accept, r.n_reject])
    for rank in range(2):
        lo, hi = sharding.shard_bounds(65536, rank, 2)
        assert _digest(r.ys[lo:hi], stats[:, lo:hi]) == d["config"]["shard_digests"][rank]


# ------------------------------------------------------------------ MCMCProcess: chains sharded over two ranks, GPU kernels
def _mcmc_rank(rank, size, port, chains, out_dir):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        from dynode_amd.infer.inference import MCMCProcess
        from examples import sir_infer_parameters as ex

        proc = MCMCProcess(numpyro_model=ex.model_fused, num_warmup=60, num_samples=40, num_chains=chains, nuts_max_tree_depth=8,
                           progress_bar=False)
        proc.infer(config=ex.get_config(), tf=100, obs_data=ex.synthetic_incidence(100))
        local = {k: v.cpu() for k, v in proc.get_samples(group_by_chain=True).items()}
        lo, hi = sharding.shard_bounds(chains, rank, size)
        assert all(v.shape[:2] == (hi - lo, 40) for v in local.values())           # a rank holds its own chains only
        gathered = proc.get_samples(group_by_chain=True, gather=True)
        torch.save({"local": local, "all": {k: v.cpu() for k, v in gathered.items()}}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_mcmc_chains_sharded_on_the_gpu(tmp_path, monkeypatch):
    """2 ranks x 8 chains of the fused SIR model (the default sampler: HIP sampler kernel + fused gradient-solve).  The gather
    returns the chains in rank order on rank 0 only, and what a rank draws is what a single process draws for that block."""
    import torch.multiprocessing as mp

    from dynode_amd.infer.inference import MCMCProcess
    from examples import sir_infer_parameters as ex

    chains = 16
    mp.spawn(_mcmc_rank, args=(2, _free_port(), chains, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert r1["all"] == {}
    for name in r0["local"]:
        assert r0["all"][name].shape[:2] == (chains, 40)
        assert torch.equal(r0["all"][name][:8], r0["local"][name]) and torch.equal(r0["all"][name][8:], r1["local"][name])
        assert not torch.equal(r0["local"][name], r1["local"][name])
    for rank, got in ((0, r0["local"]), (1, r1["local"])):
        monkeypatch.setattr(sharding, "world", lambda rank=rank: (rank, 2))
        proc = MCMCProcess(numpyro_model=ex.model_fused, num_warmup=60, num_samples=40, num_chains=chains, nuts_max_tree_depth=8,
                           progress_bar=False)
        proc.infer(config=ex.get_config(), tf=100, obs_data=ex.synthetic_incidence(100))
        mine = proc.get_samples(group_by_chain=True)
        for name in got:
            assert torch.equal(mine[name].cpu(), got[name])          # (rank, world size) alone decide a rank's draws
    monkeypatch.undo()
