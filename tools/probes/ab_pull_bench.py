"""dev probe: bench.py's headline with work pulling off / on (DYNODE_HIP_PULL), alternately on ONE box."""
import os, sys, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wl = sys.argv[1:] or ["cfg3"]
for rep in range(3):
    for mode in ("0", "1"):
        env = dict(os.environ, DYNODE_HIP_PULL=mode)
        for w in wl:
            out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-extra", "--no-cpu-baseline", "--steps", "100", "--workload", w],
                                 env=env, capture_output=True, text=True).stdout
            d = json.loads(out.strip().splitlines()[-1])
            print(w, "pull" if mode == "1" else "static", rep, round(d["roofline"]["kernel_ms"], 4), round(d["roofline"]["frac"], 4),
                  "ordered", d["roofline"]["dispatch_order"].get("with_caller_supplied_order_ms_per_launch"), flush=True)
