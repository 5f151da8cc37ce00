"""dev probe: A/B of library builds on ONE box (box-to-box spread is +-3 %): tools/probes/_lib_<name>.so variants
(tools/diag_build.sh, or copies of dynode_amd/lib/libdynode_hip.so) run bench.py alternately, three times each.
Usage: python tools/probes/ab_bench.py A B"""
import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for rep in range(3):
    for v in sys.argv[1:]:
        env = dict(os.environ, DYNODE_HIP_LIB=os.path.join(root, "tools", "probes", f"_lib_{v}.so"))
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-extra", "--no-cpu-baseline", "--steps", "100"],
                             env=env, capture_output=True, text=True).stdout
        import json
        d = json.loads(out.strip().splitlines()[-1])
        print(v, rep, round(d["roofline"]["kernel_ms"], 4), round(d["roofline"]["frac"], 4), flush=True)
