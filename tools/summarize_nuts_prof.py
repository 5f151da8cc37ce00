#!/usr/bin/env python3
"""Condense a rocprofv3 kernel trace of tools/bench_nuts.py into profiles/<tag>_nuts_*."""
import csv, glob, json, os, re, sys

out_dir, tag, log = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats = glob.glob(os.path.join(out_dir, "**", "*kernel_stats.csv"), recursive=True)[0]
trace = glob.glob(os.path.join(out_dir, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(stats)))


def short(name):
    name = re.sub(r"at::native::|\(anonymous namespace\)::", "", name)
    name = re.sub(r"\s+", " ", name)
    return name[:110]


with open(os.path.join(root, "profiles", f"{tag}_nuts_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:30]:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

tr = list(csv.DictReader(open(trace)))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(tr) if "nuts_advance" in r["Kernel_Name"]]
fused = False
if len(idx) < 8:   # one launch per iteration (dyn_solver_opts::nuts_tail): the iteration ends with the fused gradient-solve
    idx = [i for i, r in enumerate(tr) if "solve_kernel_fused" in r["Kernel_Name"]]
    fused = True
# steady state: iterations in the last quarter of the run
sel = idx[len(idx) * 3 // 4:]
per_iter, span, solve, adv = [], [], [], []
for a, b in zip(sel[:-1], sel[1:]):
    per_iter.append(b - a)
    span.append((int(tr[b]["End_Timestamp"]) - int(tr[a]["End_Timestamp"])) / 1e3)
    solve += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr[a + 1:b] if "solve_kernel" in r["Kernel_Name"]]
    adv.append((int(tr[b]["End_Timestamp"]) - int(tr[b]["Start_Timestamp"])) / 1e3)
mean = lambda x: sum(x) / max(len(x), 1)
total = sum(int(r["TotalDurationNs"]) for r in rows)
get = lambda pat: next((r for r in rows if pat in r["Name"]), None)
sk, na = get("solve_kernel_fused<float") or get("solve_kernel<float"), get("nuts_advance")
bench = {}
for line in open(log):
    if line.startswith("{"):
        bench = json.loads(line)
with open(os.path.join(root, "profiles", f"{tag}_nuts_iteration.md"), "w") as f:
    f.write(f"# cfg 4 sampler iteration under rocprofv3 --kernel-trace ({tag})\n\n")
    f.write("Command: `rocprofv3 --kernel-trace --stats -- python3 tools/bench_nuts.py --chains 128 --warmup 300 --samples 300 "
            + " ".join(sys.argv[4:]) + "` (`tools/profile_nuts.sh`)\n")
    f.write("(kernel-trace inflates every tiny launch to about 4.4 us).\n\n")
    f.write(f"- sampler iterations ({'fused gradient-solve + sampler launches, `solve_kernel_fused`' if fused else '`dyn_nuts_advance` launches'}): {len(idx)}\n")
    f.write(f"- kernels per iteration (steady state): {mean(per_iter):.1f}; iteration span under trace {mean(span):.1f} us\n")
    if sk:
        f.write(f"- gradient-solve kernel `{short(sk['Name'])}`: {sk['Calls']} calls, avg {float(sk['AverageNs']) / 1e3:.1f} us ({sk['Percentage']} % of GPU time)\n")
    if na and not fused:
        f.write(f"- `dynnuts::nuts_advance`: {na['Calls']} calls, avg {float(na['AverageNs']) / 1e3:.1f} us, min {float(na['MinNs']) / 1e3:.1f}, max {float(na['MaxNs']) / 1e3:.1f} ({na['Percentage']} %); steady state {mean(adv):.1f} us\n")
    f.write(f"- total GPU kernel time {total / 1e9:.2f} s; kernels not named above belong to the model's torch program (none when the potential is folded, infer/folded.py)\n")
    if bench:
        f.write(f"- traced run: {bench.get('seconds', 0):.2f} s, {bench.get('transitions_per_s', 0):.0f} transitions/s, KS p {bench.get('ks_pvalues_vs_quadrature')}\n")
    a, b = sel[len(sel) // 2], sel[len(sel) // 2 + 1]
    f.write("\nKernel sequence of one steady-state iteration (start us, duration us, kernel):\n\n```\n")
    t0 = int(tr[a]["End_Timestamp"])
    for r in tr[a + 1:b + 1]:
        f.write("%8.1f %6.1f  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                       short(r["Kernel_Name"])[:90]))
    f.write("```\n")
print(open(os.path.join(root, "profiles", f"{tag}_nuts_iteration.md")).read())
