"""dev probe: bench.py's headline under two settings of one environment variable, alternately on ONE box.
Usage: python tools/probes/ab_env_bench.py VAR value_a value_b [workload]   ("-" = unset)"""
import os, sys, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
var, va, vb = sys.argv[1:4]
wl = sys.argv[4] if len(sys.argv) > 4 else "cfg3"
for rep in range(3):
    for v in (va, vb):
        env = dict(os.environ)
        env.pop(var, None)
        if v != "-":
            env[var] = v
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-extra", "--no-cpu-baseline", "--steps", "100", "--workload", wl],
                             env=env, capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        print(var, v, rep, round(d["roofline"]["kernel_ms"], 4), round(d["roofline"]["frac"], 4), d["roofline"]["kernel"][-12:], flush=True)
