"""dev probe: mean per launch of every counter in a `rocprofv3 --pmc ... --output-format csv -d DIR` output directory, solve
kernels only, over the later half of the launches (the timed ones of a bench.py run).    python tools/probes/pmc_mean.py DIR"""
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "solve_kernel" in r["Kernel_Name"] or "seip" in r["Kernel_Name"]:
            acc[(r["Dispatch_Id"], r["Counter_Name"])].append(float(r["Counter_Value"]))
per = collections.defaultdict(list)
for (d, c), v in acc.items():
    per[c].append(sum(v))
for c, v in sorted(per.items()):
    v = v[len(v) // 2:]      # (the later launches: the timed ones)
    print(f"{c:28s} {sum(v) / len(v):.4g}  (n={len(v)})")
