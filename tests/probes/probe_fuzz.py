"""dev probe: the randomized parity sweep of tests/test_gpu_parity.py (fuzz_case / fuzz_compare) at scale.
    python tests/probes/probe_fuzz.py [n_cases] [first_seed] [f32]"""
import os, sys

import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/probes/x.py -> repo root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import fuzz_case, fuzz_compare

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dtype = torch.float32 if len(sys.argv) > 3 and sys.argv[3] == "f32" else torch.float64
bad = ran = 0
worst = 0.0
for seed in range(seed0, seed0 + N):
    case = fuzz_case(seed)
    if case is None:
        continue
    ran += 1
    # float64: rounding level.  float32: the two runs may take different accept / reject decisions where the error
    # estimate sits at 1, so they agree to the solver tolerance of the case (a few rtol), or to rounding for constant steps
    rtol = case[6].get("rtol", 0.0) if "constant_dt" not in case[6] else 0.0
    bound = 1e-10 if dtype == torch.float64 else 2e-5 + 30.0 * rtol
    try:
        err, same = fuzz_compare(case, dtype)
    except Exception as e:  # noqa: BLE001
        err, same = repr(e)[:200], False
    if not same or err >= bound:
        bad += 1
        print("MISMATCH seed", seed, case[0], "t1", case[4], "n_save", len(case[5]), case[6], "err", err, flush=True)
    elif isinstance(err, float):
        worst = max(worst, err)
print(f"{N} cases drawn, {ran} run (the rest: shape not compiled for float64 / that method), {bad} mismatches, worst error {worst:.2e}")
