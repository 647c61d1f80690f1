"""Development aid: condense a tools/profile.sh output directory (rocprofv3 CSVs under
gpurun_out/prof_<tag>) into the small files committed under profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
stats = glob.glob(f"{src}/trace/*/*_kernel_stats.csv")
if stats:
    shutil.copy(stats[0], f"profiles/{tag}_kernel_stats.csv")
out = {}
for d in sorted(glob.glob(f"{src}/pmc_*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        k = r["Kernel_Name"]
        k = "onesweep_kernel" if "onesweep" in k else ("hist_kernel" if "hist_kernel" in k else None)
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        for c, x in v.items():
            out.setdefault(k, {})[c] = {"mean_per_launch": sum(x) / len(x), "launches": len(x)}
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
one = out.get("onesweep_kernel", {})
if "FETCH_SIZE" in one and "WRITE_SIZE" in one:
    # MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of
    # the bytes of a coalesced streaming read (TCC_EA0_RDREQ counted at 64 B although requests are 128 B): double it
    fetch = one["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
    write = one["WRITE_SIZE"]["mean_per_launch"] * 1024
    t = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python bench.py --steps 3 --warmup 1, profile tag {tag}",
         "correction": "FETCH_SIZE doubled (gfx950 counts 128-B read requests as 64 B); WRITE_SIZE as reported; both KiB -> bytes",
         "onesweep_pass_fetch_bytes_per_launch": fetch, "onesweep_pass_write_bytes_per_launch": write,
         "onesweep_pass_hbm_bytes_per_launch": fetch + write,
         "algorithmic_bytes_per_launch": 8_000_000_000}
    if "hist_kernel" in out and "FETCH_SIZE" in out["hist_kernel"]:
        t["hist_fetch_bytes_per_launch"] = out["hist_kernel"]["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
    json.dump(t, open("profiles/pmc_traffic.json", "w"), indent=1)
    print(json.dumps(t, indent=1))
traces = glob.glob(f"{src}/trace/*/*_kernel_trace.csv")
if traces:
    # per-launch durations of the scatter pass in the traced bench run (--steps 3 --warmup 1: the first
    # 4 launches belong to the untimed warm-up step and run slower; kernel_stats.csv averages them in)
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(traces[0]))
         if "onesweep" in r["Kernel_Name"]]
    h = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(traces[0]))
         if "hist_kernel" in r["Kernel_Name"]]
    bench_line = None
    for line in open(f"{src}/trace.log"):
        if line.startswith("{") and "roofline" in line:
            bench_line = json.loads(line)
    summ = {"onesweep_launch_ms": [round(x, 4) for x in d], "onesweep_avg_ms_all_launches": sum(d) / len(d),
            "onesweep_avg_ms_timed_steps": sum(d[4:]) / max(1, len(d[4:])), "hist_launch_ms": [round(x, 4) for x in h],
            "bench_line_of_the_same_run": bench_line}
    json.dump(summ, open(f"profiles/{tag}_kernel_trace_summary.json", "w"), indent=1)
    print("onesweep avg (timed steps) %.4f ms, bench events in the same run: %s" % (
        summ["onesweep_avg_ms_timed_steps"], bench_line and bench_line["roofline"]["avg_launch_ms"]))
if stats:
    for row in list(csv.DictReader(open(stats[0])))[:6]:
        print(row["Name"][:70], row["Calls"], row["AverageNs"], row["Percentage"])
