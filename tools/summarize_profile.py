"""Development aid: condense a tools/profile.sh output directory (rocprofv3 CSVs under
gpurun_out/prof_<tag>) into the small files committed under profiles/.

Kernels are told apart by name; the scatter pass (onesweep_kernel) is launched once per level and sort, so its
launches are also split by level (launch index modulo LEVELS): on the hybrid route only the two top levels move
keys, the others return at once."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 4
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
KERNELS = ("msd_scatter_kernel", "msd_finish_kernel", "presample_kernel", "onesweep_kernel", "hist16_kernel", "hist_kernel", "local_count_sort_kernel", "local_wide_sort_kernel", "local_sort_kernel",
           "route_kernel", "scan_kernel", "clear_unless_hybrid_kernel", "copyback_kernel")


def short(name):
    """kernel base name: `void (anonymous namespace)::local_wide2_sort_kernel<true>(...)` -> local_wide2_sort_kernel; torch's own
    kernels (fills, copies, the generator) are left out"""
    import re
    m = re.match(r"^(?:void )?\(anonymous namespace\)::([A-Za-z0-9_]+)", name)
    return m.group(1) if m else None


stats = glob.glob(f"{src}/trace/*/*_kernel_stats.csv")
if stats:
    shutil.copy(stats[0], f"profiles/{tag}_kernel_stats.csv")
out = {}
for d in sorted(glob.glob(f"{src}/pmc_*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    seen = collections.Counter()
    rows = sorted(csv.DictReader(open(d)), key=lambda r: int(r["Dispatch_Id"]))
    per_dispatch = collections.OrderedDict()
    for r in rows:
        per_dispatch.setdefault(r["Dispatch_Id"], (r["Kernel_Name"], {}))[1][r["Counter_Name"]] = float(r["Counter_Value"])
    for _, (kname, counters) in per_dispatch.items():
        k = short(kname)
        if not k:
            continue
        label = k
        if k == "onesweep_kernel":
            label = f"onesweep_kernel.level{seen[k] % levels}"
        if k == "msd_scatter_kernel":
            label = "msd_scatter_kernel.pass_" + "ab"[seen[k] % 2]
        seen[k] += 1
        for c, x in counters.items():
            agg[label][c].append(x)
    for k, v in agg.items():
        for c, x in v.items():
            out.setdefault(k, {})[c] = {"mean_per_launch": sum(x) / len(x), "launches": len(x)}
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)

# HBM traffic per launch (MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half
# of the bytes of a coalesced streaming read — TCC_EA0_RDREQ counted at 64 B although requests are 128 B —: double it)
traffic = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline`, profile tag {tag}",
           "correction": "FETCH_SIZE doubled (gfx950 counts 128-B read requests as 64 B); WRITE_SIZE as reported; both KiB -> bytes",
           "kernels": {}}
for k, v in sorted(out.items()):
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        f, w = v["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2, v["WRITE_SIZE"]["mean_per_launch"] * 1024
        if f + w > 1e8:  # launches that moved keys
            traffic["kernels"][k] = {"fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w}
moving = [v["hbm_bytes_per_launch"] for k, v in traffic["kernels"].items() if k.startswith("onesweep_kernel.") or k.startswith("msd_scatter_kernel.")]
if moving:
    traffic["scatter_pass_hbm_bytes_per_launch"] = sum(moving) / len(moving)   # mean over the scatter launches that moved keys (bench.py: roofline.traffic)
    traffic["scatter_launches_that_moved_keys"] = len(moving)
    traffic["algorithmic_bytes_per_launch"] = 8_000_000_000 if levels == 4 else 16_000_000_000
    if levels == 4 and "f32" not in tag and "u64" not in tag:   # the headline u32 command: what bench.py's roofline.traffic cites
        json.dump(traffic, open("profiles/pmc_traffic.json", "w"), indent=1)
    json.dump(traffic, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
    print(json.dumps(traffic, indent=1))

traces = glob.glob(f"{src}/trace/*/*_kernel_trace.csv")
if traces:
    # per-launch durations in the traced bench run (--steps 3 --warmup 1: the first sort is the untimed warm-up)
    rows = sorted(csv.DictReader(open(traces[0])), key=lambda r: int(r["Start_Timestamp"]))
    dur = collections.defaultdict(list)
    seen = collections.Counter()
    for r in rows:
        k = short(r["Kernel_Name"])
        if not k:
            continue
        label = f"onesweep_kernel.level{seen[k] % levels}" if k == "onesweep_kernel" else ("msd_scatter_kernel.pass_" + "ab"[seen[k] % 2] if k == "msd_scatter_kernel" else k)
        seen[k] += 1
        dur[label].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    bench_line = None
    for line in open(f"{src}/trace.log"):
        if line.startswith("{") and "roofline" in line:
            bench_line = json.loads(line)
    summ = {"per_kernel_ms": {}}
    for k, v in sorted(dur.items()):
        per_sort = max(1, len(v) // 4)  # launches of one sort (4 sorts: 1 warm-up + 3 timed)
        timed = v[per_sort:]
        summ["per_kernel_ms"][k] = {"launches": len(v), "avg_ms_all": round(sum(v) / len(v), 4),
                                    "avg_ms_timed_steps": round(sum(timed) / max(1, len(timed)), 4),
                                    "launch_ms": [round(x, 4) for x in v]}
    moved = [v["avg_ms_timed_steps"] for k, v in summ["per_kernel_ms"].items()
             if (k.startswith("onesweep_kernel.") or k.startswith("msd_scatter_kernel.")) and v["avg_ms_timed_steps"] > 0.3]
    if moved:
        summ["scatter_avg_ms_timed_steps_launches_that_moved_keys"] = round(sum(moved) / len(moved), 4)
    # the kernel bench.py's roofline object is about (pass A of the atomic route, else a K3 pass): trace against the run's events
    dom = summ["per_kernel_ms"].get("msd_scatter_kernel.pass_a")
    if dom and dom["avg_ms_timed_steps"] > 0.3 and bench_line and "pass A" in bench_line["roofline"].get("kernel", ""):
        summ["roofline_kernel_avg_ms_timed_steps_trace"] = dom["avg_ms_timed_steps"]
        print("roofline kernel (msd_scatter_kernel, pass A): trace %.4f ms; bench events in the same run: %s ms" % (
            dom["avg_ms_timed_steps"], bench_line["roofline"]["avg_launch_ms"]))
    summ["bench_line_of_the_same_run"] = bench_line
    json.dump(summ, open(f"profiles/{tag}_kernel_trace_summary.json", "w"), indent=1)
    print("scatter pass avg over the launches that moved keys (timed steps, trace): %s ms; bench events in the same run: %s ms" % (
        summ.get("scatter_avg_ms_timed_steps_launches_that_moved_keys"), bench_line and bench_line["roofline"]["avg_launch_ms"]))
    for k, v in summ["per_kernel_ms"].items():
        print("  %-32s %3d launches  avg(timed) %.4f ms" % (k, v["launches"], v["avg_ms_timed_steps"]))
