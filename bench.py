#!/usr/bin/env python3
"""Headline benchmark: device-resident radix sort of 1 B uniform-random u32 keys per GPU
(BASELINE.json configs[1]), Gkeys/s, with the dominant kernel's HBM roofline and a CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype u32|u64|f32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
A bare `python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) starts its own N ranks as fresh child
processes (torch.distributed.run, 127.0.0.1) BEFORE anything touches the GPU, waits, and exits with their code; rank 0's
JSON line goes to the inherited stdout.

A step = one sort of the workload.  N == 1: `rdst_hip_sort_device` on 10^9 keys already in HBM.
N > 1 (weak scaling, BASELINE configs[4] "C5": u64 keys unless --dtype says otherwise): every rank holds 10^9 keys;
one step = the sharded route (local top-digit histogram, all-gather of the 256 counts, top-digit scatter,
all-to-all over RCCL/xGMI, local sort) — value = N * 10^9 * K keys / wall time.  Every step sorts a fresh unsorted copy that was
placed in HBM before the timed region (no copies inside it).

The JSON line also carries (protocol of SURVEY.md §8(d) / BASELINE.md §3-4):
  median_ms_per_step  median of the K steps (HIP events around each step, same stream) beside the mean
  roofline       the dominant kernel (K3, one scatter pass): algorithmic bytes per launch (2k per key: one
                 read + one write) / its average launch duration over the launches that moved keys, taken
                 with HIP events recorded between launches on the kernels' stream, inside the timed region
  kernels        the same for every stage of the route taken (K1h / K1, K3, K4 ...)
  copy_ceiling   a measured device copy of the same array (read + write), beside the 8 TB/s spec
  configs        N == 1: BASELINE configs[2] and [3] (10^9 u64, 10^9 f32) run after the timed region, same
                 protocol, with their own roofline figures (B = 136 and 36 bytes per key); the inputs on which the
                 default is NOT at its best (u32 on the forced LSD route, the reference's bimodal bench input,
                 f32 normal(0, 1), reverse arange), each checked bit for bit against torch.sort of the mapped keys
                 and timed beside the LSD-only setting; and u64_8e9 — 8 * 10^9 u64 keys on this one GPU, the
                 denominator of C5's scaling target
  end_to_end     `rdst_hip_sort` on a host slice of the same workload (PCIe both ways) — never `value`
  cpu_baseline   the oracle (C restatement of rdst's StandardTuner route, OpenMP) timed on this box's host
                 cores on THE SAME ARRAY the GPU leg sorted; the two outputs are compared bit for bit
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KEYS_PER_GPU = 1_000_000_000
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
# name -> (torch view dtype, numpy dtype, key bytes, levels, seed (C2, C3, C4))
DTYPES = {
    "u32": ("uint32", "uint32", 4, 4, 0x5D570002),
    "u64": ("uint64", "uint64", 8, 8, 0x5D570003),
    "f32": ("float32", "float32", 4, 4, 0x5D570004),
}


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


def cargo_probe():
    """the real rdst needs cargo; record that it is absent (expected)."""
    try:
        out = subprocess.run(["cargo", "--version"], capture_output=True, text=True, timeout=10)
        return out.stdout.strip() or "cargo present but silent"
    except Exception as e:  # noqa: BLE001
        return f"absent ({type(e).__name__})"


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: N ranks as children of torch.distributed.run on 127.0.0.1.
    The parent only waits; rank 0 of the children prints the JSON line to the inherited stdout."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(host_keys, gpu_sorted_host):
    """Oracle (kind "port") on the array the GPU leg sorted: median of 3 runs on fresh copies, and the
    result compared bit for bit with the device's."""
    import numpy as np
    from oracle import oracle as O
    O.load()
    threads = effective_cpus()
    n = host_keys.size
    times = []
    out = None
    for _ in range(3):
        a = host_keys.copy()
        t0 = time.perf_counter()
        O.sort(a, tuner="standard", multi_threaded=True, threads=threads)
        times.append(time.perf_counter() - t0)
        out = a
        if sum(times) > 45:   # slow host: one run is the sample
            break
    med = statistics.median(times)
    w = host_keys.dtype.itemsize
    equal = bool(np.array_equal(out.view(f"u{w}"), gpu_sorted_host.view(f"u{w}")))
    assert equal, "CPU oracle and device outputs differ"
    return {"value": round(n / med / 1e9, 4), "unit": "Gkeys/s", "cores": threads, "kind": "port",
            "sample": f"the full {n}-key array the GPU leg sorted (D2H once, outside every clock), median of {len(times)}, "
                      f"oracle/rdst_oracle.c StandardTuner route, {threads} OpenMP threads",
            "bit_identical_to_device_output": equal, "cargo": cargo_probe()}


def gen_keys(torch, n, name, seed):
    """uniform random bit patterns of the key type, generated on the device"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    if DTYPES[name][2] == 4:
        return torch.randint(-(2**31), 2**31, (n,), dtype=torch.int32, device="cuda", generator=g)
    return torch.randint(-(2**63), 2**63 - 1, (n,), dtype=torch.int64, device="cuda", generator=g)


def mapped_signed(torch, x, name):
    """signed-comparable image of rdst's key order (src/radix_key_impl.rs) of the raw bits in x"""
    mn = torch.iinfo(x.dtype).min
    if name.startswith("u"):
        return x ^ mn
    if name.startswith("f"):
        return torch.where(x < 0, ~x ^ mn, x)
    return x


def is_sorted(torch, m, chunk=1 << 27):
    for s in range(0, m.numel(), chunk):
        e = min(m.numel(), s + chunk + 1)
        if not bool((m[s + 1:e] >= m[s:e - 1]).all()):
            return False
    return True


def kernel_table(rdst_amd, runs, n, kb, levels):
    """per-stage averages over the recorded pipelines: {stage: {avg_ms, launches, GBps, frac}}.  Launches that return at once
    (a level the plan turned off; K1 on the hybrid / atomic routes; the LSD fallback behind a route that succeeded) are
    listed apart as *_skipped."""
    acc = {}
    for r in runs:
        prof = rdst_amd.profile_run(r, levels)
        if not prof:
            continue
        moving = max([ms for nm, _lv, ms in prof["stages"] if nm in ("pass", "msd_pass_a", "msd_pass_b", "local_sort")] or [0.0])
        for name, _level, ms in prof["stages"]:
            if name in ("pass", "histogram", "histogram16", "msd_pass_a", "msd_pass_b", "local_sort") and ms < 0.25 * moving and ms < 0.1:
                name += "_skipped"
            acc.setdefault(name, []).append(ms)
    # (4-byte keys travel from the last scatter pass to K4 as their low 16-bit halves: 4 + 2 and 2 + 4 bytes per key)
    half = kb == 4
    alg = {"pass": 2 * kb * n, "msd_pass_a": 2 * kb * n, "msd_pass_b": (6 if half else 2 * kb) * n,
           "local_sort": (6 if half else 2 * kb) * n, "histogram": kb * n, "histogram16": kb * n}
    out = {}
    for name, v in acc.items():
        avg = sum(v) / len(v)
        e = {"avg_ms": round(avg, 4), "launches": len(v)}
        if name in alg and avg > 0:
            gbps = alg[name] / (avg * 1e-3) / 1e9
            e.update({"algorithmic_bytes_per_launch": alg[name], "GBps": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBPS, 4)})
        out[name] = e
    return out


def route_bytes_per_key(route, kb, levels):
    """HBM bytes the route moves per key when everything goes to plan: atomic: pass A (read + write), pass B
    (read + write; 4-byte keys leave it as 16-bit halves), K4 (read + write); hybrid: K1h's read on top, both passes whole
    keys but for the last one's halves; LSD: K1's read and a read + write per level."""
    if route == "atomic":
        return {4: 4 + 4 + 4 + 2 + 2 + 4, 8: 6 * 8}.get(kb)
    if route == "hybrid":
        return {4: 4 + 4 + 4 + 4 + 2 + 2 + 4, 8: 7 * 8}.get(kb)
    return kb * (2 * levels + 1)


def sort_level_fields(route, kb, levels, keys_per_gpu_per_s):
    """the sort as a whole against HBM: by SURVEY.md §8(d)'s convention (the bytes an LSD sort of this key width moves — the
    figure round 1's target was set in; a route that moves fewer bytes can pass 1.0 here) and by the bytes the route
    that ran really moves"""
    b_lsd, b_route = kb * (2 * levels + 1), route_bytes_per_key(route, kb, levels)
    out = {"sort_algorithmic_bytes_per_key": b_lsd,
           "sort_algorithmic_GBps": round(keys_per_gpu_per_s * b_lsd / 1e9, 1),
           "sort_roofline_frac_per_gpu": round(keys_per_gpu_per_s * b_lsd / 1e9 / HBM_PEAK_GBPS, 4),
           "sort_roofline_note": "LSD-equivalent bytes (SURVEY §8(d): k(2L+1) per key) / time / 8 TB/s; see route_* for the bytes this route moves"}
    if b_route:
        out.update({"route_bytes_per_key": b_route, "route_GBps": round(keys_per_gpu_per_s * b_route / 1e9, 1),
                    "route_hbm_frac_per_gpu": round(keys_per_gpu_per_s * b_route / 1e9 / HBM_PEAK_GBPS, 4)})
    return out


def other_inputs(torch, n):
    """The inputs on which the default route choice is not at its best: name -> (dtype name, maker).
    bimodal = the reference's own bench input, gen_inputs(n, 16) (benches/bench_utils.rs:43-53, benches/full_sort.rs:68-70):
    the first half of a uniform vector shifted right by 16, the second half left."""
    def bimodal():
        x = gen_keys(torch, n, "u32", 0x5D570006)
        h = n // 2
        x[:h] = (x[:h] >> 16) & 0xFFFF   # logical shift of the u32 bit pattern held in int32
        x[h:] = x[h:] << 16
        return x

    def normal():
        g = torch.Generator(device="cuda").manual_seed(0x5D570007)
        return torch.randn(n, generator=g, device="cuda", dtype=torch.float32).view(torch.int32)

    def reverse():
        return torch.arange(n - 1, -1, -1, dtype=torch.int32, device="cuda")

    def sum4():   # a smooth bell over the whole range: most keys in 16-bit prefixes of one to four K4 tiles
        acc = torch.zeros(n, dtype=torch.int64, device="cuda")
        for i in range(4):
            acc += gen_keys(torch, n, "u32", 0x5D570010 + i).to(torch.int64) & 0xFFFFFFFF
        return (acc >> 2).to(torch.int32)   # (wraps: the bit pattern is what counts)

    return {"u32_bimodal_shift16": ("u32", bimodal), "f32_normal": ("f32", normal), "u32_reverse_arange": ("u32", reverse),
            "u32_sum_of_4_uniforms": ("u32", sum4)}


def exact_against_torch_sort(torch, src, out, name):
    """bit-exact: rdst's key map is a bijection, so mapped(out) must equal torch.sort(mapped(src)) element for element"""
    want = torch.sort(mapped_signed(torch, src, name)).values
    got = mapped_signed(torch, out.view(src.dtype), name)
    return bool(torch.equal(want, got))


def mix_checksum(torch, x, chunk=1 << 27):
    """(sum, sum of a bit-mixed image) of an integer tensor, both mod 2^64, in bounded temporaries"""
    a = b = 0
    for s in range(0, x.numel(), chunk):
        c = x[s:s + chunk].to(torch.int64)
        a = (a + int(c.sum())) & (2**64 - 1)
        m = (c * -7046029254386353131) ^ (c >> 29)
        b = (b + int(m.sum())) & (2**64 - 1)
    return a, b


def big_u64_config(torch, rdst_amd, n_big):
    """8 * 10^9 u64 keys on ONE GPU: the denominator of C5's scaling target; 64 GB + 64 GB tmp + the
    unsorted source + the workspace (24 GB of 64-bit status rows) ~ 220 GB.  No clones: the buffer is refilled from the
    source before every sort, outside the events."""
    free, _total = torch.cuda.mem_get_info()
    need = 3 * 8 * n_big + 40 * 2**30
    if free < need:
        return {"skipped": f"needs ~{need / 2**30:.0f} GiB of free HBM, {free / 2**30:.0f} available"}
    src = gen_keys(torch, n_big, "u64", 0x5D570005)   # C5, rank 0's seed
    buf = torch.empty_like(src).view(torch.uint64)
    tmp = torch.empty_like(buf)
    ms = []
    for _ in range(3):   # the first one also allocates the workspace
        buf.view(torch.int64).copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rdst_amd.sort_device_tensor(buf, tmp, check=False)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    rdst_amd.device_status()
    route = rdst_amd.last_route()
    out = buf.view(torch.int64)
    ok_sorted = is_sorted(torch, mapped_signed(torch, out, "u64"))
    ok_sum = mix_checksum(torch, out) == mix_checksum(torch, src)
    assert ok_sorted and ok_sum, f"u64_8e9: sorted={ok_sorted} checksums_equal={ok_sum}"
    best = min(ms[1:])
    res = {"keys": n_big, "route": route, "ms_per_step": round(best, 3), "all_ms": [round(x, 3) for x in ms],
           "Gkeys_per_s": round(n_big / best / 1e6, 3), "checked": "sortedness + two checksums against the input (outside the clock)",
           **sort_level_fields(route, 8, 8, n_big / (best * 1e-3)),
           "what": "BASELINE configs[4]'s single-GPU leg: 8e9 uniform u64 keys on one MI355X (best of 2 after a warm-up)"}
    del src, buf, tmp, out
    torch.cuda.empty_cache()
    return res


def size_points(torch, rdst_amd):
    """The rate outside the 10^9-key sweet spot (the whole sweep: profiles/r03_size_sweep.json): short slices,
    where K4's 65 536 workgroups are a floor, and slices beyond the byte-saving routes' window, which are split on their top
    byte.  Best of 3 after a warm-up, sortedness and a checksum checked outside the clock."""
    out = {}
    for name, n in (("u32", 100_000_000), ("u32", 250_000_000), ("u32", 2_000_000_000), ("u64", 100_000_000), ("u64", 1 << 31)):
        free, _total = torch.cuda.mem_get_info()
        kb = DTYPES[name][2]
        if free < 3 * kb * n + 30 * 2**30:
            out[f"{name}_{n}"] = {"skipped": "not enough free HBM"}
            continue
        src = gen_keys(torch, n, name, 0x5D570020 + len(out))
        buf = torch.empty_like(src).view(getattr(torch, DTYPES[name][0]))
        tmp = torch.empty_like(buf)
        ms = []
        for _ in range(4):
            buf.view(src.dtype).copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rdst_amd.sort_device_tensor(buf, tmp, check=False)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        rdst_amd.device_status()
        res = buf.view(src.dtype)
        assert is_sorted(torch, mapped_signed(torch, res, name)) and mix_checksum(torch, res) == mix_checksum(torch, src), f"size point {name} {n}"
        best = min(ms[1:])
        out[f"{name}_{n}"] = {"keys": n, "route": rdst_amd.last_route(), "ms_per_step": round(best, 3), "Gkeys_per_s": round(n / best / 1e6, 2)}
        del src, buf, tmp, res
        torch.cuda.empty_cache()
        rdst_amd.release_workspace()
    return out


ROUTE_TEXT = {
    "atomic": "two MSD scatter passes that claim space with atomics (no counting read) + in-LDS sort of every bucket",
    "hybrid": "K1h (counts of the top 16 bits) + 2 scatter passes on them + in-LDS sort of every bucket",
}


def timed_sorts(torch, rdst_amd, src, name, K, W, step=None, fence=None, in_place=True):
    """W warm-up + K timed sorts of fresh copies of `src` (already in HBM).  Returns (elapsed s, per-step ms,
    profiled run indices, last output).  in_place=False: `step` leaves its input alone (the sharded sort returns a new
    tensor), so every step reads `src` itself — 25 copies of a 10^9-key u64 shard would be 200 GB per GPU."""
    view_dt = getattr(torch, DTYPES[name][0])
    bufs = [src.clone().view(view_dt) for _ in range(K + W)] if in_place else [src.view(view_dt)] * (K + W)
    tmp = torch.empty_like(bufs[0])
    if step is None:
        def step(buf):  # noqa: E306
            rdst_amd.sort_device_tensor(buf, tmp, check=False)
            return buf
    if fence is None:
        fence = torch.cuda.synchronize
    out = None
    for i in range(W):
        out = step(bufs[i])
    fence()
    rdst_amd.set_profiling(True)   # HIP events between the launches of every timed step
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(K):
        out = step(bufs[W + i])
        ev[i + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    rdst_amd.device_status()       # any kernel failure of any step (the error word is sticky) raises here
    runs = list(range(rdst_amd.profile_runs()))
    per_step = [ev[i].elapsed_time(ev[i + 1]) for i in range(K)]
    return elapsed, per_step, runs, out, bufs, tmp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--keys", type=int, default=KEYS_PER_GPU, help="keys per GPU (default: the BASELINE workload)")
    ap.add_argument("--dtype", choices=sorted(DTYPES), default=None,
                    help="key type of the timed region (default: the headline u32 at N == 1; u64 — BASELINE configs[4], C5 — at N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the other configs, the copy ceiling and the end-to-end leg")
    ap.add_argument("--no-big", action="store_true", help="skip the 8 * 10^9-key u64 config (needs ~220 GB of HBM)")
    ap.add_argument("--route", choices=("auto", "lsd"), default="auto", help="lsd: never take the hybrid route (A/B)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one.  Nothing in this process has touched the GPU (torch is not even imported).
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if args.dtype is None:
        args.dtype = "u32" if world == 1 else "u64"
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

    import numpy as np
    import rdst_amd
    from rdst_amd.sharded import sharded_sort

    if args.route == "lsd":
        rdst_amd.set_hybrid(False)
    n = args.keys
    K, W = args.steps, args.warmup
    name = args.dtype
    _vdt, npdt, kb, levels, seed = DTYPES[name]
    B_key = kb * (2 * levels + 1)   # algorithmic bytes per key of the whole sort

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    src = gen_keys(torch, n, name, seed + rank)
    step = (lambda buf: sharded_sort(buf)) if distributed else None
    elapsed, per_step, runs, out, bufs, tmp = timed_sorts(torch, rdst_amd, src, name, K, W, step=step, fence=fence, in_place=not distributed)
    route = rdst_amd.last_route()
    kernels = kernel_table(rdst_amd, runs, n if not distributed else out.numel(), kb, levels)
    rdst_amd.set_profiling(False)

    breakdown = None
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # outside the clock: where a sharded step's time goes (two more steps with events between the stages; max over ranks), and
        # what one rank's shard costs on its own GPU without the exchange (the same dtype: the N = 1 line of this bench is u32)
        stages = ("split", "counts_and_plan", "exchange", "local_sort")
        tm = {}
        for _ in range(2):
            tm = {}
            sharded_sort(bufs[0], timings=tm)   # (leaves its input alone)
        alone = []
        work = torch.empty_like(bufs[0])
        for _ in range(3):
            work.view(src.dtype).copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rdst_amd.sort_device_tensor(work, tmp, check=False)
            e1.record()
            torch.cuda.synchronize()
            alone.append(e0.elapsed_time(e1))
        t = torch.tensor([tm.get(k, 0.0) for k in stages] + [min(alone)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        vals = [round(float(x), 3) for x in t.tolist()]
        breakdown = {"stages_ms_max_over_ranks": dict(zip(stages, vals[:4])), "one_shard_sorted_alone_ms": vals[4],
                     "one_shard_sorted_alone_Gkeys_per_s": round(n / vals[4] / 1e6, 2),
                     "what": "events between the stages of one more sharded step (outside the clock); the last two figures: this "
                             "dtype's single-GPU sort of one rank's shard, no exchange"}

    # sanity of the last result (outside the clock): sorted, and (single GPU) the same multiset as its input
    itype = torch.int32 if kb == 4 else torch.int64
    assert is_sorted(torch, mapped_signed(torch, out.view(itype), name)), "output not sorted"
    if not distributed:
        assert int(out.view(itype).sum()) == int(src.sum()), "checksum differs: not a permutation of the input"

    total_keys = n * world * K
    value = total_keys / elapsed / 1e9
    ms_per_step = elapsed / K * 1e3
    roof = None
    # the dominant kernel: the scatter pass — K3 (onesweep_kernel) on the LSD and hybrid routes, msd_scatter_kernel on the atomic one
    dom = [k for k in ("msd_pass_a", "pass") if k in kernels and "GBps" in kernels[k]][:1]
    if dom:
        launches = sum(kernels[k]["launches"] for k in dom)
        avg_ms = sum(kernels[k]["avg_ms"] * kernels[k]["launches"] for k in dom) / launches
        bytes_per_launch = kernels[dom[0]]["algorithmic_bytes_per_launch"]
        gbps = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp) and not distributed and n == KEYS_PER_GPU and name == "u32":
            try:
                tj = json.load(open(tp))
                if "msd_pass_a" in dom:
                    traffic = tj["kernels"]["msd_scatter_kernel.pass_a"]["hbm_bytes_per_launch"]
                else:
                    traffic = tj.get("onesweep_pass_hbm_bytes_per_launch", tj.get("scatter_pass_hbm_bytes_per_launch"))
                traffic_src = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; not measured in this run)"
            except Exception:  # noqa: BLE001
                traffic = None
        roof = {"bound": "hbm", "kernel": ("msd_scatter_kernel, pass A (whole keys in, whole keys out: the launch that moves the most bytes)" if "msd_pass_a" in dom
                                           else "onesweep_kernel (K3, one scatter pass)"),
                "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "traffic_source": traffic_src, "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": round(avg_ms, 4),
                "launches_timed": launches}

    line = None
    if rank == 0:
        line = {
            "metric": f"radix_sort_throughput_1B_uniform_{name}", "value": round(value, 3), "unit": "Gkeys/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(ms_per_step, 4),
            "median_ms_per_step": round(statistics.median(per_step), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": name, "data": "synthetic",
            "config": {"workload": (f"C5 (BASELINE configs[4]) at {world} GPUs, weak: " if distributed and name == "u64" and n == KEYS_PER_GPU else "")
                                   + f"{n} uniform-random {name} keys per GPU, device-resident; route '{route}': "
                                   + ROUTE_TEXT.get(route, f"LSD, {levels} passes x 8 bits")
                                   + (f", sharded: MSD top byte + all-gather of the 256 counts + all-to-all ({backend}) + local sort" if distributed else ""),
                       "keys_per_gpu": n, "total_keys": n * world, "seed": seed, "route": route,
                       "parallelism": f"shard{world}" if distributed else "single"},
            **sort_level_fields(route, kb, levels, total_keys / elapsed / world),
            "roofline": roof,
            "kernels": kernels,
        }
        if breakdown:
            line["sharded_breakdown"] = breakdown

    if rank == 0 and not distributed:
        # ---- outside the timed region: copy ceiling, the other BASELINE configs, end to end, CPU leg
        gpu_sorted_host = None
        host_keys = None
        if not args.no_cpu_baseline:
            host_keys = src.cpu().numpy().view(npdt)          # D2H once, outside every clock
            gpu_sorted_host = out.view(itype).cpu().numpy().view(npdt)
        if not args.no_extras:
            # measured streaming ceilings on the same array, 16 B per lane (the library's own yardstick kernels):
            # a copy (read + write of n*k bytes each) and a read-only sweep, beside the 8 TB/s spec
            import ctypes
            from rdst_amd import _lib
            lib = _lib.load()
            a, b = bufs[0], tmp
            nbytes = (kb * n) // 16 * 16
            sh = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            ceil = {}
            for what, call, moved in (("copy", lambda: lib.rdst_hip_stream_copy(ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(a.data_ptr()), nbytes, sh), 2 * nbytes),
                                      ("read", lambda: lib.rdst_hip_stream_read(ctypes.c_void_p(a.data_ptr()), nbytes, sh), nbytes),
                                      ("write", lambda: lib.rdst_hip_stream_fill(ctypes.c_void_p(b.data_ptr()), nbytes, sh), nbytes)):
                for _ in range(2):
                    _lib.check(call())
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    _lib.check(call())
                e1.record()
                torch.cuda.synchronize()
                cms = e0.elapsed_time(e1) / 10
                ceil[what] = {"ms": round(cms, 4), "GBps": round(moved / (cms * 1e-3) / 1e9, 1),
                              "frac_of_spec": round(moved / (cms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
            ceil["what"] = (f"16-bytes-per-lane streaming kernels over the same {kb * n / 1e9:.1f} GB array (one contiguous piece per block, non-temporal "
                            "loads: the fastest shapes of profiles/r03_copy_sweep.json), mean of 10 launches each")
            line["copy_ceiling"] = ceil
        del bufs, tmp, out
        torch.cuda.empty_cache()
        if not args.no_extras:
            configs = {}
            for other in ("u64", "f32"):
                if other == name or n != KEYS_PER_GPU:
                    continue
                _v2, _np2, kb2, lv2, seed2 = DTYPES[other]
                src2 = gen_keys(torch, n, other, seed2)
                k2 = max(3, min(K, 5))
                el2, ps2, runs2, out2, bufs2, tmp2 = timed_sorts(torch, rdst_amd, src2, other, k2, 1)
                it2 = torch.int32 if kb2 == 4 else torch.int64
                assert is_sorted(torch, mapped_signed(torch, out2.view(it2), other)), f"{other}: output not sorted"
                assert int(out2.view(it2).sum()) == int(src2.sum()), f"{other}: checksum differs"
                med = statistics.median(ps2)
                configs[other] = {"keys": n, "steps": k2, "route": rdst_amd.last_route(), "ms_per_step": round(el2 / k2 * 1e3, 4),
                                  "median_ms_per_step": round(med, 4), "Gkeys_per_s": round(n * k2 / el2 / 1e9, 3),
                                  **sort_level_fields(rdst_amd.last_route(), kb2, lv2, n * k2 / el2),
                                  "kernels": kernel_table(rdst_amd, runs2, n, kb2, lv2)}
                rdst_amd.set_profiling(False)
                del src2, out2, bufs2, tmp2
                torch.cuda.empty_cache()
            # the inputs on which the default is not at its best, each beside the LSD-only setting of the same box
            if n == KEYS_PER_GPU and name == "u32":
                for cname, (dt, make) in other_inputs(torch, n).items():
                    srcx = make()
                    _vx, _npx, kbx, lvx, _sx = DTYPES[dt]
                    el, ps, _runs, outx, bufsx, tmpx = timed_sorts(torch, rdst_amd, srcx, dt, 3, 1)
                    rdst_amd.set_profiling(False)
                    rt = rdst_amd.last_route()
                    exact = exact_against_torch_sort(torch, srcx, outx, dt)
                    assert exact, f"{cname}: device output differs from torch.sort of the mapped keys"
                    del outx, bufsx, tmpx
                    rdst_amd.set_hybrid(False)
                    try:
                        el_l, ps_l, _r, outl, bufsl, tmpl = timed_sorts(torch, rdst_amd, srcx, dt, 3, 1)
                    finally:
                        rdst_amd.set_profiling(False)
                        rdst_amd.set_hybrid(True, 0)
                    del outl, bufsl, tmpl
                    configs[cname] = {"keys": n, "dtype": dt, "route": rt, "ms_per_step": round(statistics.median(ps), 4),
                                      "Gkeys_per_s": round(n / statistics.median(ps) / 1e6, 3), "bit_exact_vs_torch_sort": exact,
                                      "lsd_only_ms_per_step": round(statistics.median(ps_l), 4),
                                      "default_over_lsd_only": round(statistics.median(ps) / statistics.median(ps_l), 4)}
                    del srcx
                    torch.cuda.empty_cache()
                # the headline input on the forced LSD route (BASELINE configs[1] words it "LSD 4-pass")
                rdst_amd.set_hybrid(False)
                try:
                    el_l, ps_l, runs_l, outl, bufsl, tmpl = timed_sorts(torch, rdst_amd, src, "u32", 3, 1)
                    exact = exact_against_torch_sort(torch, src, outl, "u32")
                    assert exact, "u32 on the LSD route: device output differs from torch.sort"
                    configs["u32_forced_lsd"] = {"keys": n, "route": rdst_amd.last_route(), "ms_per_step": round(statistics.median(ps_l), 4),
                                                 "Gkeys_per_s": round(n / statistics.median(ps_l) / 1e6, 3), "bit_exact_vs_torch_sort": exact,
                                                 **sort_level_fields("lsd", 4, 4, n / (statistics.median(ps_l) * 1e-3)),
                                                 "kernels": kernel_table(rdst_amd, runs_l, n, 4, 4)}
                finally:
                    rdst_amd.set_profiling(False)
                    rdst_amd.set_hybrid(True, 0)
                del outl, bufsl, tmpl
                torch.cuda.empty_cache()
            line["configs"] = configs
            # end to end through the reference-shaped entry point: host slice in, host slice out (PCIe both ways)
            if host_keys is None:
                host_keys = src.cpu().numpy().view(npdt)
            e2e = []
            for _ in range(2):
                h = host_keys.copy()
                t0 = time.perf_counter()
                rdst_amd.radix_sort_unstable(h)
                e2e.append(time.perf_counter() - t0)
            if gpu_sorted_host is not None:
                assert np.array_equal(h.view(f"u{kb}"), gpu_sorted_host.view(f"u{kb}")), "host entry point and device path differ"
            import ctypes
            from rdst_amd import _lib
            t3 = [ctypes.c_float(0) for _ in range(3)]
            _lib.check(_lib.load().rdst_hip_host_timing(*[ctypes.byref(x) for x in t3]))
            line["end_to_end"] = {"ms": round(min(e2e) * 1e3, 2), "Gkeys_per_s": round(n / min(e2e) / 1e9, 3),
                                  "last_call_breakdown_ms": {"h2d": round(t3[0].value, 2), "sort_and_status": round(t3[1].value, 2), "d2h": round(t3[2].value, 2)},
                                  "link_GBps": {"h2d": round(kb * n / (t3[0].value * 1e-3) / 1e9, 1), "d2h": round(kb * n / (t3[2].value * 1e-3) / 1e9, 1)},
                                  "what": "rdst_hip_sort on a host numpy slice of the same keys: H2D + sort + D2H, best of 2 (never `value`)"}
            del h
        del src
        torch.cuda.empty_cache()
        if not args.no_extras and not args.no_big and n == KEYS_PER_GPU:
            line.setdefault("configs", {})["other_sizes"] = size_points(torch, rdst_amd)
            line.setdefault("configs", {})["u64_8e9"] = big_u64_config(torch, rdst_amd, 8 * KEYS_PER_GPU)
        line["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(host_keys, gpu_sorted_host)
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
