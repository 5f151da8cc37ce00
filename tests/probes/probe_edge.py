"""dev probe: `edge_sweep` of tests/test_gpu_parity.py (discontinuity points and save times in special position), every
mismatch printed.    python tests/probes/probe_edge.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import edge_sweep

ran, bad = edge_sweep(report=lambda b: print("MISMATCH", *b, flush=True))
print(f"{ran} cases run, {len(bad)} mismatches")
