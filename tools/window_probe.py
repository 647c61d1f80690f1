"""Development aid: the routes outside the default window — short slices with the threshold lowered, and u32 slices of 1.6 / 2 / 3.2
x 10^9 keys on the hybrid route against LSD-only, with stage times."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rdst_amd


def run(it, n, mode, min_len, stages=False):
    rdst_amd.set_hybrid(mode, min_len)
    info = torch.iinfo(it)
    g = torch.Generator(device="cuda").manual_seed(n & 0xFFFF)
    src = torch.randint(info.min, info.max, (n,), dtype=it, device="cuda", generator=g)
    keys, tmp = torch.empty_like(src), torch.empty_like(src)
    vdt = torch.uint32 if it == torch.int32 else torch.uint64
    times = []
    for _ in range(4):
        keys.copy_(src)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rdst_amd.sort_device_tensor(keys.view(vdt), tmp.view(vdt), check=False)
        e1.record()
        torch.cuda.synchronize()
        rdst_amd.device_status()
        times.append(e0.elapsed_time(e1))
    k = keys ^ info.min
    ok = bool((k[1:] >= k[:-1]).all())
    ms = min(times[1:])
    line = f"{'u32' if it == torch.int32 else 'u64'} n={n:>11d} mode={mode} min_len={min_len}: {ms:8.3f} ms {n / ms / 1e6:7.1f} Gkeys/s route={rdst_amd.last_route()} ok={ok}"
    if stages:
        keys.copy_(src)
        rdst_amd.set_profiling(True)
        rdst_amd.sort_device_tensor(keys.view(vdt), tmp.view(vdt), check=False)
        p = rdst_amd.profile_run(-1, 4)
        rdst_amd.set_profiling(False)
        line += "\n      " + "  ".join(f"{nm}{'' if lv is None else lv}={t:.3f}" for nm, lv, t in p["stages"] if t >= 0.02)
    print(line, flush=True)
    del src, keys, tmp, k
    torch.cuda.empty_cache()
    rdst_amd.release_workspace()


what = sys.argv[1] if len(sys.argv) > 1 else "small"
if what == "small":
    for it in (torch.int32, torch.int64):
        for k in (24, 25, 26, 27):
            for n in (1 << k, 3 << (k - 1)):
                run(it, n, 1, 0)
                run(it, n, 1, 1, stages=True)
else:
    for n in (1_610_612_736, 2_000_000_000, 3_221_225_472):
        run(torch.int32, n, 1, 0, stages=True)
        run(torch.int32, n, 0, 0)
