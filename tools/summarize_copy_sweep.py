"""Development aid: condense the outputs of tools/gpu_jobs/r03_copy.sh and r03_job2.sh (tools/probe/copy_sweep on the GPU
box: timings as JSON lines, rocprofv3 --pmc passes as CSV) into profiles/r03_copy_sweep.json."""
import collections
import csv
import glob
import json
import re

KIND = {"0": "copy", "1": "read", "2": "write"}
LAYOUT = ["grid_stride", "piece_per_block", "xcd_eighth_stride", "xcd_eighth_pieces"]


def lines(path):
    return [json.loads(l) for l in open(path) if l.startswith("{")]


def pmc(pattern, key_of):
    out = collections.defaultdict(dict)
    for d in sorted(glob.glob(pattern)):
        key = key_of(d)
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(d)):
            m = re.search(r"sweep_kernel<(\d), (\d), (\d), (true|false)>", r["Kernel_Name"])
            if m:
                per[KIND[m.group(1)]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for kind, c in per.items():
            for name, v in c.items():
                out[f"{key}_{kind}"][name] = round(sum(v) / len(v), 1)
    return out


sweep = [r for r in lines("gpurun_out/r03_job2/sweep.jsonl")]
table = collections.OrderedDict()
for r in sweep:
    if r.get("tag") != "sweep":
        continue
    f = str(int(r["footprint_MiB"]))
    table.setdefault(f, {}).setdefault(r["kind"], {})[r["layout"] + ("_nt" if r["nt"] else "")] = r["GBps"]
best = {f: {k: max(v.values()) for k, v in kinds.items()} for f, kinds in table.items()}
shapes = lines("gpurun_out/r03_copy/shapes.jsonl")
offsets = lines("gpurun_out/r03_copy/offsets.jsonl")
shape_best = {}
for r in shapes:
    k = f"{int(r['footprint_MiB'])}MiB_{r['kind']}"
    if k not in shape_best or r["GBps"] > shape_best[k]["GBps"]:
        shape_best[k] = {x: r[x] for x in ("layout", "in_flight", "nt", "blocks_per_cu", "GBps")}
off = collections.defaultdict(dict)
for r in offsets:
    off[f"{int(r['footprint_MiB'])}MiB_{r['layout']}"][str(r["dst_gap_bytes"])] = r["GBps"]

c1 = pmc("gpurun_out/r03_copy/pmc_*/*/*_counter_collection.csv", lambda d: re.search(r"pmc_(\d+)_", d).group(1) + "MiB_piece_per_block")
c2 = pmc("gpurun_out/r03_job2/pmc_l*/*/*_counter_collection.csv", lambda d: "4096MiB_" + LAYOUT[int(re.search(r"pmc_l(\d)_", d).group(1))])


def derived(c):
    d = {}
    if c.get("TCC_EA0_RDREQ_sum"):
        d["avg_read_latency_TCC_cycles"] = round(c.get("TCC_EA0_RDREQ_LEVEL_sum", 0) / c["TCC_EA0_RDREQ_sum"])
    if c.get("TCC_EA0_WRREQ_sum"):
        d["avg_write_latency_TCC_cycles"] = round(c.get("TCC_EA0_WRREQ_LEVEL_sum", 0) / c["TCC_EA0_WRREQ_sum"])
    if c.get("TCP_UTCL1_REQUEST_sum"):
        d["utcl1_miss_per_million_requests"] = round(1e6 * (c.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0) + c.get("TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum", 0)) / c["TCP_UTCL1_REQUEST_sum"], 1)
    if c.get("GRBM_GUI_ACTIVE"):
        d["kernel_ms_at_2.4GHz_from_GUI_ACTIVE_over_8_XCDs"] = round(c["GRBM_GUI_ACTIVE"] / 8 / 2.4e6, 4)
    return d


counters = {}
for src in (c1, c2):
    for k, v in src.items():
        counters.setdefault(k, {}).update(v)
for k in counters:
    counters[k]["derived"] = derived(counters[k])

out = {
    "what": "tools/probe/copy_sweep.hip on one MI355X: 16 bytes per lane, 256-thread blocks, mean over back-to-back launches (~48 GB of traffic per case); "
            "GB/s = bytes read + bytes written / time",
    "footprint_sweep_GBps": table,
    "best_per_footprint_GBps": best,
    "runtime_at_4GiB": [r for r in sweep if r.get("tag") == "runtime"],
    "shape_sweep_best": shape_best,
    "dst_offset_sweep_GBps": off,
    "counters_per_launch": counters,
    "findings": [
        "copy (read + write) runs at 6.4-7.0 TB/s while source + destination fit the 256 MiB Infinity Cache (footprint <= 128 MiB each) and at 5.0-5.5 TB/s from 192 MiB to 8 GiB: the 6.3-6.5 TB/s figures are cache-assisted, the HBM figure for a mixed stream is ~5.4",
        "pure streams on HBM: reads 7.0 TB/s (non-temporal, contiguous piece per block), writes 5.5-6.1 TB/s (the runtime's fill: 6.4); a copy takes ~20 % longer than its read and its write would take one after the other",
        "grid-stride layouts lose 10-15 % to the first-level TLB: 260 k UTCL1 misses + 2.3 M misses-under-miss per 4 GiB copy against 819 + 15 k with one contiguous piece per block (every 4 KiB step of a block lands 8 MiB further, in another 2 MiB fragment)",
        "the distance between source and destination does not matter (0 ... 16 MiB + odd: within 1 %)",
        "average read latency at the L2's memory side under load: ~2 500 TCC cycles (pure read), ~3 250 (copy); DRAM-credit stalls 3-12 % of TCC busy cycles: the L2 is not what holds the stream back",
    ],
}
json.dump(out, open("profiles/r03_copy_sweep.json", "w"), indent=1)
print(json.dumps(best, indent=0))
