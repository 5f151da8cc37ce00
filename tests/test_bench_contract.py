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
    # profiled traffic is attached only for the instance AND the device code that were profiled: the line says which it compared
    from dynode_amd import _abi

    prov = roof["traffic_provenance"]
    assert prov["kernel_source_hash"] == _abi.kernel_source_hash()
    if roof["traffic"] is not None:
        assert prov["profiled"]["kernel_source_hash"] == prov["kernel_source_hash"] and "stale" not in prov
    elif prov["profiled"] is not None:
        assert prov.get("stale")
    # lock-step cost of the static grid, reported next to the mean: a wave loops as long as its slowest trajectory needs
    cfg = d["config"]
    assert cfg["trajectories_per_wave"] == 2 and cfg["mean_steps_per_trajectory"] <= cfg["mean_loop_iterations_per_wave"] < 1.2 * cfg["mean_steps_per_trajectory"]
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "trajectories/s" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]
    # value = trajectories of all steps over the wall clock; the kernel-only rate cannot be lower
    assert d["value"] <= 16384 / (roof["kernel_ms"] * 1e-3) * 1.001
    # the headline is the batch in its GIVEN order: no forecast, nothing carried over from earlier launches
    order = roof["dispatch_order"]
    assert order["kind"].startswith("none") and "trained_on" not in order
