"""dev probe: cfg 4 posterior z-scores (128 chains x 1000 + 1000) for one library build (DYNODE_HIP_LIB) and adaptation mode."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
for adaptation in sys.argv[1:] or ["per_chain", "pooled"]:
    r = bench.nuts_side_measurement(adaptation=adaptation)
    print(os.environ.get("DYNODE_HIP_LIB", "default")[-24:], adaptation, round(r["seconds"], 2), "div", r["divergences"], "leapfrogs", round(r["mean_leapfrogs_per_transition"], 2),
          {s: (round(q["ks_p"], 3), round(q["mean_z"], 2), round(q["sd_z"], 2), round(q["sd"] / q["quad_sd"], 4)) for s, q in r["posterior_vs_quadrature"].items()}, flush=True)
