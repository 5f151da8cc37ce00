import sys, json
sys.path.insert(0, "/root/repo")
import bench
print(json.dumps({k: v for k, v in bench.nuts_multi_strain_side(9).items() if k in ("seconds", "us_per_gradient_solve", "gradient_solves", "divergences")}))
