#!/usr/bin/env python3
"""Condense a tools/profile.sh run into profiles/<tag>_rocprof_summary.md (+ traffic.json entry).

FETCH_SIZE / WRITE_SIZE are in KiB.  Per MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reads
exactly half of a wide coalesced streaming read, so reads are doubled; WRITE_SIZE is exact for
16-byte-per-lane streaming stores (this kernel's store shape).
"""
import csv, glob, json, os, sys, collections

out_dir, tag = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "seip_kernel" if any(a.startswith("seip") for a in extra) else "solve_kernel"

stats, side = None, []
for f in glob.glob(os.path.join(out_dir, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Name"]:
            stats = r
        elif "dynord::" in r["Name"]:      # dispatch order: forecast + counting sort in front of every ordered launch
            side.append(r)
counters = collections.defaultdict(list)
meta = {}
for f in glob.glob(os.path.join(out_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size")}
mean = {k: sum(v) / len(v) for k, v in counters.items()}

def build_rev() -> str:
    """Source revision the profiled library was built from (tools/stamp_rev.sh writes it before a gpurun call:
    the GPU box has no .git)."""
    try:
        return open(os.path.join(root, "dynode_amd", "lib", "BUILD_REV")).read().strip()
    except OSError:
        return "unknown"


def kernel_source_hash() -> str:
    sys.path.insert(0, root)
    from dynode_amd import _abi

    return _abi.kernel_source_hash()


def instance_name(raw: str) -> str:
    """rocprofv3 prints `void dyn::solve_kernel<...>(dyn::KArgs<float>)`; dyn_last_kernel_name() returns the middle part."""
    name = raw.strip().strip('"')
    if name.startswith("void "):
        name = name[5:]
    cut = name.rfind(">(")
    return name[:cut + 1] if cut > 0 else name


workload, batch = "cfg3", None
for i, a in enumerate(extra):
    if a == "--workload": workload = extra[i + 1]
    if a == "--batch": batch = int(extra[i + 1])
defaults = {"cfg2": 4096, "cfg3": 16384, "cfg3d136": 16384, "cfg5": 8192, "seip": 4096, "seip3": 4096, "seip83": 4096, "seip84": 2048}
batch = batch or defaults[workload]

kernel = instance_name(stats["Name"]) if stats else None
lines = [f"# rocprofv3 summary `{tag}` -- bench.py --workload {workload} (B={batch} per GPU)", "",
         f"- source revision: `{build_rev()}` (kernel sources: `{kernel_source_hash()}`)", f"- dispatched instance: `{kernel}`", ""]
if stats:
    lines += ["## kernel-trace --stats (bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra: 50 timed launches behind the untimed ones -- settle phase of >= 40 ms of work + 5 warm-up: `calls` counts them all --, the batch "
              "in its given order -- no forecast, nothing carried over between launches)", "",
              "| kernel | calls | avg ns | min ns | max ns | % of GPU time |", "|---|---|---|---|---|---|",
              f"| `{stats['Name'][:120]}` | {stats['Calls']} | {float(stats['AverageNs']):.0f} | {stats['MinNs']} | {stats['MaxNs']} | {stats['Percentage']} |"]
    for r in side:
        lines.append(f"| `{r['Name'][:120]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |")
    lines.append("")
if meta:
    lines += ["## dispatch", "", "| " + " | ".join(meta) + " |", "|" + "---|" * len(meta), "| " + " | ".join(meta.values()) + " |", ""]
if mean:
    lines += ["## PMC (mean per launch, separate passes)", "", "| counter | value |", "|---|---|"]
    for k in sorted(mean):
        lines.append(f"| {k} | {mean[k]:.6g} |")
    lines.append("")
hbm = None
if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
    rd = mean["FETCH_SIZE"] * 1024 * 2      # gfx950 correction: x2 for wide coalesced reads
    wr = mean["WRITE_SIZE"] * 1024
    hbm = rd + wr
    lines += ["## HBM traffic per launch", "",
              f"- reads  = FETCH_SIZE x 1024 x 2 (gfx950 half-count correction) = {rd/1e6:.2f} MB",
              f"- writes = WRITE_SIZE x 1024 = {wr/1e6:.2f} MB",
              f"- total  = {hbm/1e6:.2f} MB", ""]
if "SQ_WAVE_CYCLES" in mean and "SQ_WAVES" in mean:
    wc = mean["SQ_WAVE_CYCLES"]
    lines += ["## derived", ""]
    for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_SCA"):
        if k in mean:
            lines.append(f"- {k} / SQ_WAVE_CYCLES = {mean[k]/wc:.3f}")
    lines.append(f"- VALU instructions per wave = {mean.get('SQ_INSTS_VALU',0)/mean['SQ_WAVES']:.0f}; SALU per wave = {mean.get('SQ_INSTS_SALU',0)/mean['SQ_WAVES']:.0f}; LDS per wave = {mean.get('SQ_INSTS_LDS',0)/mean['SQ_WAVES']:.0f}")
    lines.append(f"- waves per launch = {mean['SQ_WAVES']:.0f} for {batch} trajectories (a work-pulling launch is a resident grid: every wave integrates several "
                 f"trajectories one after the other); VALU instructions per TRAJECTORY-lane-group = {mean.get('SQ_INSTS_VALU',0)/batch:.0f} x (trajectories per wave)")
    try:   # the bench line of the traced run: step attempts per trajectory and loop iterations per wave (lock-step cost)
        log = open(os.path.join(out_dir, "trace.log")).read().splitlines()
        cfg = next(json.loads(l) for l in reversed(log) if l.startswith("{") and '"metric"' in l)["config"]
        if cfg.get("mean_loop_iterations_per_wave") is not None:
            lines.append(f"- step attempts per trajectory (mean) = {cfg['mean_steps_per_trajectory']:.1f}; loop iterations per wave (mean over waves of "
                         f"the most attempts among its {cfg['trajectories_per_wave']} trajectories, static grid in the given order) = "
                         f"{cfg['mean_loop_iterations_per_wave']:.1f}")
    except (OSError, StopIteration, KeyError, ValueError):
        pass
    if "GRBM_GUI_ACTIVE" in mean and stats:
        lines.append(f"- effective clock ~ GRBM_GUI_ACTIVE / 8 / kernel time = {mean['GRBM_GUI_ACTIVE']/8/float(stats['AverageNs']):.2f} GHz (profiled pass)")
    lines.append("")
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
open(os.path.join(root, "profiles", f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines))
if hbm:
    tp = os.path.join(root, "profiles", "traffic.json")
    rec = json.load(open(tp)) if os.path.exists(tp) else {}
    rec[f"{workload}:{batch}"] = {"hbm_bytes_per_launch": hbm, "read_bytes": rd, "write_bytes": wr, "source": f"profiles/{tag}_rocprof_summary.md",
                                  "kernel": kernel, "rev": build_rev(), "kernel_source_hash": kernel_source_hash(),
                                  "valu_insts_per_wave": (mean["SQ_INSTS_VALU"] / mean["SQ_WAVES"]) if "SQ_INSTS_VALU" in mean and mean.get("SQ_WAVES") else None,
                                  "waves_per_launch": mean.get("SQ_WAVES"),
                                  "clock_ghz": (mean["GRBM_GUI_ACTIVE"] / 8 / float(stats["AverageNs"])) if "GRBM_GUI_ACTIVE" in mean and stats else None,
                                  "kernel_avg_ns": float(stats["AverageNs"]) if stats else None}
    json.dump(rec, open(tp, "w"), indent=1, sort_keys=True)
print("\n".join(lines))
