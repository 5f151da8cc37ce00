"""bench.py's one-line JSON contract (the driver parses it) and its refusal to run without a GPU."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, timeout=600):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout,
                          cwd=ROOT)


def test_bench_refuses_to_run_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = _run("--steps", "1", "--warmup", "0", timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_line_has_every_field_of_the_contract():
    r = _run("--steps", "3", "--warmup", "1", "--no-extra", "--cpu-sample", "512")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "trajectories" in d["metric"] and d["unit"] == "trajectories/s" and d["value"] > 1e6
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and 0.2 < roof["frac"] < 0.8
    assert roof["traffic"] is None or roof["traffic"] > 3e9
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "trajectories/s" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]
    # value = trajectories of all steps over the wall clock; the kernel-only rate cannot be lower
    assert d["value"] <= 16384 / (roof["kernel_ms"] * 1e-3) * 1.001
    # the dispatch order behind the number: learned on draws that are not the timed batch, given-order time beside it
    order = roof["dispatch_order"]
    assert "forecast" in order["kind"] and "other draws" in order["trained_on"] and order["forecast_correlation_on_this_batch"] > 0.8


def test_other_draws_share_the_constants_and_nothing_else():
    """bench.other_draws: the training batch of the dispatch-order forecast -- same contact matrix, age shares and save grid as
    the timed workload, different parameter rows (rows B..2B of the same generator)."""
    import numpy as np

    sys.path.insert(0, ROOT)
    import bench
    from dynode_amd import synthetic

    for name, B in (("cfg3", 256), ("cfg3d136", 256), ("cfg5", 128), ("cfg2", 128), ("seip", 8)):
        gen = synthetic.WORKLOADS[name]
        wl = gen(B, bench.SEEDS[name])
        other = bench.other_draws(gen, wl, bench.SEEDS[name])
        assert other is not None and other.params.shape == wl.params.shape and other.model == wl.model
        assert np.array_equal(other.contact, wl.contact) and np.array_equal(other.save_ts, wl.save_ts)
        assert not np.array_equal(other.params, wl.params)
        if wl.y0.ndim == 2:
            assert other.y0.shape == wl.y0.shape and float(np.abs(other.y0.sum(1) - wl.y0.sum(1)).max()) < 1e-9      # same population
        else:
            assert np.array_equal(other.y0, wl.y0)
