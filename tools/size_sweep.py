"""Device sort rate against the length, 2^24 .. 2^32 uniform u32 keys (and u64 to 2^31) on the default route settings:
which route each length takes and how far its Gkeys/s lies from the 10^9-key rate.
    python tools/size_sweep.py [out.json]"""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rdst_amd

out = sys.argv[1] if len(sys.argv) > 1 else None
rows = []
for name, it, top in (("u32", torch.int32, 32), ("u64", torch.int64, 31)):
    lengths = sorted({1 << k for k in range(24, top + 1)} | {3 << k for k in range(23, top - 1)} | {10**8, 10**9, 2 * 10**9 if top == 32 else 10**9})
    info = torch.iinfo(it)
    ref = None
    for n in lengths:
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
        k = keys ^ info.min   # unsigned order as signed order
        ok = bool((k[1:] >= k[:-1]).all()) and int(keys.sum()) == int(src.sum())
        ms = min(times[1:])
        row = {"dtype": name, "n": n, "ms": round(ms, 4), "gkeys_per_s": round(n / ms / 1e6, 2), "route": rdst_amd.last_route(), "ok": ok}
        rows.append(row)
        print(f"{name} n={n:>11d}  {ms:9.3f} ms  {row['gkeys_per_s']:7.1f} Gkeys/s  route={row['route']}  ok={ok}", flush=True)
        del src, keys, tmp, k
        torch.cuda.empty_cache()
        rdst_amd.release_workspace()
    at_1e9 = next(r["gkeys_per_s"] for r in rows if r["dtype"] == name and r["n"] == 10**9)
    for r in rows:
        if r["dtype"] == name:
            r["vs_1e9"] = round(r["gkeys_per_s"] / at_1e9, 3)
if out:
    with open(out, "w") as f:
        json.dump({"what": "uniform keys, device-resident, min of 3 timed sorts after one warm-up, default route settings", "rows": rows}, f, indent=1)
