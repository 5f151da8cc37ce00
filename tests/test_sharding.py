"""The N > 1 path on CPU: two gloo ranks shard a batch, solve their blocks, gather/all-reduce.

The GPU solver is replaced by an oracle-backed stand-in (tests may use the oracle as the checker);
what is under test is dynode_amd.sharding: block bounds, ragged gathers in batch order, ensemble
moments -- the only collectives the hot path uses (after the solve, never inside it).
"""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H
from dynode_amd import sharding, synthetic


def test_shard_bounds_cover_the_batch_exactly():
    for total in (0, 1, 7, 8, 1024, 16385):
        for size in (1, 2, 3, 8):
            blocks = [sharding.shard_bounds(total, r, size) for r in range(size)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            assert max(h - l for l, h in blocks) - min(h - l for l, h in blocks) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(8, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_solver(model, y0, params, contact, t1, save_ts, **kw):
    """CPU stand-in for engine.solve_batch with the same result fields."""
    from dynode_amd.engine import BatchResult

    ys, st, na, nr = H.O.solve(H.omodel(model), np.asarray(y0), np.asarray(params), contact, t1, save_ts,
                               dtype=np.float64)
    t = torch.as_tensor
    return BatchResult(t(ys), t(st), t(na), t(nr), model.compartment_names, model.compartment_sizes)


def _worker(rank, size, port, B, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        wl = synthetic.sir_age_stratified(B, seed=3, A=4, t1=60.0)
        res = sharding.solve_sharded(wl.model, wl.y0, wl.params, wl.contact, wl.t1, wl.save_ts, solver=_oracle_solver)
        assert (res.lo, res.hi) == sharding.shard_bounds(B, rank, size) and res.total == B
        assert res.local.ys.shape[0] == res.hi - res.lo                      # results stay sharded
        status = sharding.gather_rows(res.local.status, B)
        final = sharding.gather_rows(res.local.ys[:, -1, :], B)             # small per-trajectory output
        mean, var, n = sharding.allreduce_ensemble_moments(res.local.ys)
        if rank == 0:
            torch.save({"status": status, "final": final, "mean": mean, "var": var, "n": n}, os.path.join(out_dir, "r0.pt"))
        else:
            assert status is None and final is None
            torch.save({"mean": mean, "var": var}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [10, 13])  # even and ragged split over 2 ranks
def test_two_rank_gloo_shard_gather_and_moments(tmp_path, B):
    mp.spawn(_worker, args=(2, _free_port(), B, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(tmp_path / "r0.pt")
    other = torch.load(tmp_path / "r1.pt")
    wl = synthetic.sir_age_stratified(B, seed=3, A=4, t1=60.0)
    full = _oracle_solver(wl.model, wl.y0, wl.params, wl.contact, wl.t1, wl.save_ts)
    assert torch.equal(got["status"], full.status)
    assert torch.equal(got["final"], full.ys[:, -1, :])                      # batch order preserved, bit for bit
    assert got["n"] == B
    torch.testing.assert_close(got["mean"], full.ys.mean(0), rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(got["var"], full.ys.var(0, unbiased=False), rtol=1e-8, atol=1e-10)
    assert torch.equal(got["mean"], other["mean"]) and torch.equal(got["var"], other["var"])   # all-reduce: same on all ranks


def test_single_process_is_a_no_op_group():
    assert sharding.world() == (0, 1)
    t = torch.arange(6.0).reshape(3, 2)
    assert sharding.gather_rows(t, 3) is t
    mean, var, n = sharding.allreduce_ensemble_moments(t.reshape(3, 1, 2))
    assert n == 3 and torch.allclose(mean, t.mean(0, keepdim=True).double())


# ------------------------------------------------------------------ MCMCProcess: chains sharded over ranks
def _normal_model(obs):
    """The reference's inference smoke model (tests/test_infer/test_inference_processes.py:15-24): one Normal site."""
    from dynode_amd.infer import distributions as dist_
    from dynode_amd.infer import handlers

    mu = handlers.sample("mu", dist_.Normal(0.0, 5.0))
    handlers.sample("obs", dist_.Normal(mu[..., None], 1.0), obs=obs)


def _mcmc(num_chains):
    from dynode_amd.infer.inference import MCMCProcess

    return MCMCProcess(numpyro_model=_normal_model, num_warmup=60, num_samples=40, num_chains=num_chains,
                       nuts_max_tree_depth=6, progress_bar=False, mcmc_kwargs={"sampler": "eager"})


_OBS = torch.tensor([0.3, 1.1, 0.8, 1.6, 0.9, 1.3], dtype=torch.float64)


def _mcmc_worker(rank, size, port, chains, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        proc = _mcmc(chains)
        proc.infer(obs=_OBS)
        local = proc.get_samples(group_by_chain=True)["mu"]
        lo, hi = sharding.shard_bounds(chains, rank, size)
        assert local.shape == (hi - lo, 40)                                   # every rank holds its own chains only
        gathered = proc.get_samples(group_by_chain=True, gather=True)         # the one collective of the inference path
        flat = proc.get_samples(gather=True)
        if rank == 0:
            assert flat["mu"].shape == (chains * 40,)
            torch.save({"all": gathered["mu"], "local": local}, os.path.join(out_dir, "r0.pt"))
        else:
            assert gathered == {} and flat == {}
            torch.save({"local": local}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("chains", [6, 7])   # even and ragged split
def test_two_rank_gloo_mcmc_chains_are_sharded_gathered_in_order_and_seeded_per_rank(tmp_path, chains, monkeypatch):
    mp.spawn(_mcmc_worker, args=(2, _free_port(), chains, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    lo1, hi1 = sharding.shard_bounds(chains, 1, 2)
    assert r0["all"].shape == (chains, 40)
    assert torch.equal(r0["all"][:lo1], r0["local"]) and torch.equal(r0["all"][lo1:hi1], r1["local"])   # chain order = rank order
    assert not torch.equal(r0["local"][:2], r1["local"][:2])                   # ranks do not repeat each other's streams
    # what a rank draws depends on (its rank, the world size) only: repeat each rank's share in this process
    for rank, got in ((0, r0["local"]), (1, r1["local"])):
        monkeypatch.setattr(sharding, "world", lambda rank=rank: (rank, 2))
        proc = _mcmc(chains)
        proc.infer(obs=_OBS)
        assert torch.equal(proc.get_samples(group_by_chain=True)["mu"], got)
    monkeypatch.undo()
    post = r0["all"].reshape(-1)
    want_mean = float(_OBS.sum() / (len(_OBS) + 1 / 25.0))                     # conjugate posterior: N(sum / (n + 1/25), 1 / (n + 1/25))
    assert abs(float(post.mean()) - want_mean) < 0.15 and 0.25 < float(post.std()) < 0.6
