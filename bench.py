#!/usr/bin/env python3
"""Headline benchmark: device-resident radix sort of 1 B uniform-random u32 keys per GPU
(BASELINE.json configs[1]), Gkeys/s, with the dominant kernel's HBM roofline and a CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one sort of the workload.  N == 1: `rdst_hip_sort_device` on 10^9 keys already in HBM.
N > 1 (weak scaling): every rank holds 10^9 keys; one step = the sharded route (local top-digit
histogram, all-gather of the 256 counts, top-digit scatter, all-to-all over RCCL/xGMI, local LSD
sort) — value = N * 10^9 * K keys / wall time.  Every step sorts a fresh unsorted copy
that was placed in HBM before the timed region (no copies inside it).

The JSON line also carries:
  roofline      the K3 scatter-pass kernel: algorithmic bytes per launch (8 B/key for u32: one
                read + one write of every key) / its average launch duration, taken
                with HIP events recorded between launches on the kernels' stream, inside the
                timed region.
  cpu_baseline  the oracle (C restatement of rdst's StandardTuner route, OpenMP) timed on this
                box's host cores on a bounded sample of the same workload, rank 0, N == 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KEYS_PER_GPU = 1_000_000_000
SEED = 0x5D570002
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
ELEM = 4
LEVELS = 4


def effective_cpus():
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:  # noqa: BLE001
        pass
    return max(1, n)


def cpu_baseline(n_total):
    """Oracle (kind "port") on a bounded sample: ~10-20 s of CPU work."""
    import numpy as np
    from oracle import oracle as O
    O.load()
    threads = effective_cpus()
    rng = np.random.default_rng(SEED)

    def run(n):
        a = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
        t0 = time.perf_counter()
        O.sort(a, tuner="standard", multi_threaded=True, threads=threads)
        dt = time.perf_counter() - t0
        assert bool((a[1:] >= a[:-1]).all())
        return dt

    probe = 50_000_000
    t_probe = run(probe)
    rate = probe / t_probe
    n = int(min(n_total, max(probe, rate * 4.0)))      # ~4 s per run, 3 runs
    n = max(60_000_000, n)                              # stay above the 50 M Scanning threshold of the 1 B route
    n = min(n, n_total)
    times = sorted(run(n) for _ in range(3))
    med = times[1]
    return {"value": round(n / med / 1e9, 4), "unit": "Gkeys/s", "cores": threads, "kind": "port",
            "sample": f"{n} uniform u32 keys (of the {n_total}-key workload), median of 3, "
                      f"oracle/rdst_oracle.c StandardTuner route, {threads} OpenMP threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--keys", type=int, default=KEYS_PER_GPU, help="keys per GPU (default: the BASELINE workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be started with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # rehearsal knobs (not used by the driver): several ranks on ONE GPU with gloo collectives, to
    # exercise the N > 1 code path on a single-GPU box (RCCL refuses two ranks on one device)
    backend = os.environ.get("RDST_BENCH_BACKEND", "nccl")
    if os.environ.get("RDST_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import rdst_amd
    from rdst_amd.sharded import sharded_sort

    n = args.keys
    K, W = args.steps, args.warmup
    g = torch.Generator(device="cuda").manual_seed(SEED + rank)
    src = torch.randint(-(2**31), 2**31, (n,), dtype=torch.int32, device="cuda", generator=g)
    # one unsorted copy per step, resident before the clock starts
    bufs = [src.clone().view(torch.uint32) for _ in range(K + W)]
    tmp = torch.empty(n, dtype=torch.uint32, device="cuda")

    def step(buf):
        if distributed:
            return sharded_sort(buf)
        rdst_amd.sort_device_tensor(buf, tmp, check=False)
        return buf

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for i in range(W):
        out = step(bufs[i])
    fence()
    rdst_amd.set_profiling(True)   # HIP events between the launches of every timed step
    t0 = time.perf_counter()
    for i in range(K):
        out = step(bufs[W + i])
    fence()
    elapsed = time.perf_counter() - t0
    rdst_amd.device_status()

    # per-kernel durations over the timed region: every full-sort pipeline recorded above
    pass_ms, hist_ms, clear_ms = [], [], []
    for r in range(rdst_amd.profile_runs()):
        prof = rdst_amd.profile_run(r, LEVELS)
        if prof and len(prof["passes"]) == LEVELS and min(prof["passes"]) > 0:
            pass_ms += prof["passes"]
            hist_ms.append(prof["histogram"])
            clear_ms.append(prof["clear"])
    rdst_amd.set_profiling(False)

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity of the last result (outside the clock)
    chk = out.view(torch.int32) ^ (-(2**31))
    assert bool((chk[1:] >= chk[:-1]).all()), "output not sorted"

    total_keys = n * world * K
    value = total_keys / elapsed / 1e9
    ms_per_step = elapsed / K * 1e3
    avg_pass = sum(pass_ms) / len(pass_ms) if pass_ms else None
    bytes_per_launch = 2 * ELEM * (n if not distributed else out.numel())
    roof = None
    if avg_pass:
        ach = bytes_per_launch / (avg_pass * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp) and not distributed and n == KEYS_PER_GPU:
            try:
                traffic = json.load(open(tp)).get("onesweep_pass_hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        roof = {"bound": "hbm", "kernel": "onesweep_kernel (K3, one scatter pass)", "achieved": round(ach, 1),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": round(avg_pass, 4),
                "launches_timed": len(pass_ms), "histogram_ms": round(sum(hist_ms) / len(hist_ms), 4),
                "clear_ms": round(sum(clear_ms) / len(clear_ms), 4)}

    if rank == 0:
        line = {
            "metric": "radix_sort_throughput_1B_uniform_u32", "value": round(value, 3), "unit": "Gkeys/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{n} uniform-random u32 keys per GPU, device-resident, LSD 4 passes x 8 bits"
                                   + (", sharded: MSD top byte + RCCL all-to-all + local LSD" if distributed else ""),
                       "keys_per_gpu": n, "total_keys": n * world, "seed": SEED,
                       "parallelism": f"shard{world}" if distributed else "single"},
            "sort_algorithmic_GBps": round(total_keys * ELEM * (2 * LEVELS + 1) / elapsed / 1e9 / world, 1),
            "sort_roofline_frac_per_gpu": round(total_keys * ELEM * (2 * LEVELS + 1) / elapsed / 1e9 / world / HBM_PEAK_GBPS, 4),
            "roofline": roof,
        }
        if not distributed and not args.no_cpu_baseline:
            del bufs, tmp, src
            torch.cuda.empty_cache()
            line["cpu_baseline"] = cpu_baseline(n)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
