"""dev probe: the nine-site NUTS run of bench.py's side figures alone (128 chains x (300 + 300)), e.g. under
`rocprofv3 --kernel-trace --stats -- python3 tools/probes/probe_nine_sites.py` for the per-kernel split of its iteration."""
import json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

print(json.dumps({k: v for k, v in bench.nuts_multi_strain_side(9).items() if k in ("seconds", "us_per_gradient_solve", "gradient_solves", "divergences")}))
