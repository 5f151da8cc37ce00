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
