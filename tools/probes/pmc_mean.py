"""dev probe: mean of every counter per dispatch of the solve / SEIP kernels in a rocprofv3 --pmc output directory."""
import csv, glob, os, sys, collections
for d in sys.argv[1:]:
    c = collections.defaultdict(list)
    grid = None
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "solve_kernel" in r["Kernel_Name"] or "seip_kernel" in r["Kernel_Name"]:
                c[r["Counter_Name"]].append(float(r["Counter_Value"]))
                grid = (r["Grid_Size"], r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
    print(d, "grid/vgpr/sgpr/scratch", grid, " ".join(f"{k}={sum(v) / len(v):.4g}" for k, v in sorted(c.items())))
