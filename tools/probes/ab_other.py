"""dev probe: A/B of library builds on ONE box over the side workloads (cfg3d136, cfg2, cfg5 share): HIP-event ms per launch.
Usage: python tools/probes/ab_other.py A B [workload ...]   (tools/probes/_lib_<name>.so)"""
import os, sys, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, root)
    import torch
    from dynode_amd import synthetic
    from dynode_amd.engine import solve_batch
    res = {}
    for name in sys.argv[2:]:
        wl = synthetic.WORKLOADS[name]()
        m = wl.model
        y0, p, C, ts = (torch.as_tensor(x, dtype=torch.float32, device="cuda") for x in (wl.y0, wl.params, wl.contact, wl.save_ts))
        r = solve_batch(m, y0, p, C, wl.t1, ts)
        out, st = r.ys, (r.status, r.n_accept, r.n_reject)
        for _ in range(12):
            solve_batch(m, y0, p, C, wl.t1, ts, out=out, stats_out=st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            solve_batch(m, y0, p, C, wl.t1, ts, out=out, stats_out=st)
        e1.record()
        torch.cuda.synchronize()
        res[name] = round(e0.elapsed_time(e1) / 40, 4)
    print(json.dumps(res))
    sys.exit(0)
known = ("cfg3", "cfg3d136", "cfg2", "cfg5", "seip", "seip3", "seip83", "seip84")
names = [a for a in sys.argv[1:] if a in known] or ["cfg3", "cfg3d136", "cfg2", "cfg5", "seip", "seip83"]
for rep in range(3):
    for v in [a for a in sys.argv[1:] if a not in known]:
        env = dict(os.environ, DYNODE_HIP_LIB=os.path.join(root, "tools", "probes", f"_lib_{v}.so"))
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + names, env=env, capture_output=True, text=True)
        print(v, rep, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:], flush=True)
