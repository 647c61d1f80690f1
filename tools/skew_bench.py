"""Development aid: device sort time on non-uniform inputs (1e9 u32 keys unless argv[1])."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rdst_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
if os.environ.get("RDST_MODE"):   # rdst_hip_set_hybrid mode: 0 LSD only, 7 hybrid first, 11 no giant kernels ...
    rdst_amd.set_hybrid(int(os.environ["RDST_MODE"]) or False)
g = torch.Generator(device="cuda").manual_seed(5)
rnd = torch.randint(-(2**31), 2**31, (n,), dtype=torch.int32, device="cuda", generator=g)
cases = {
    "uniform random": rnd,
    "already sorted (arange)": torch.arange(n, dtype=torch.int32, device="cuda"),
    "reverse sorted": torch.arange(n - 1, -1, -1, dtype=torch.int32, device="cuda"),
    "sorted random": None,
    "16-bit values (two empty levels)": rnd & 0xFFFF,
    "bimodal (gen_inputs shift 16)": torch.cat([(rnd[: n // 2] >> 16) & 0xFFFF, rnd[n // 2:] << 16]),
    "all equal": torch.full((n,), 1234567, dtype=torch.int32, device="cuda"),
    "256 distinct values": rnd & 0xFF00,
    "low byte in one digit group (one chain in pass 1)": rnd & ~0xE0,
    "byte 1 in two digits": rnd & ~0xFE00,
    "byte 1 constant (skipped level)": (rnd & ~0xFF00) | 0x4200,
    "gaussian-ish (sum of 4 uniforms)": ((rnd >> 2) + (torch.roll(rnd, 1) >> 2) + (torch.roll(rnd, 2) >> 2) + (torch.roll(rnd, 3) >> 2)),
    "ids below 2^30 (top 2 bits shared)": rnd & 0x3FFFFFFF,
    "one rank's share of 8 (top byte in [32, 64))": (rnd & 0x1FFFFFFF) | 0x20000000,
    "f32 normal(0, 1) [sorted as f32]": torch.randn(n, dtype=torch.float32, device="cuda", generator=g).view(torch.int32),
    "f32 uniform [0, 1) [sorted as f32]": torch.rand(n, dtype=torch.float32, device="cuda", generator=g).view(torch.int32),
}
if len(sys.argv) > 2:
    cases = {k: v for k, v in cases.items() if any(w in k for w in sys.argv[2].split(","))}
tmp = torch.empty(n, dtype=torch.uint32, device="cuda")
keys = torch.empty(n, dtype=torch.int32, device="cuda")
for name, src in cases.items():
    if src is None:
        src = rnd.clone().view(torch.uint32)
        rdst_amd.sort_device_tensor(src, tmp)
        src = src.view(torch.int32)
    times = []
    vdt = torch.float32 if "as f32" in name else torch.uint32
    for _ in range(3):
        keys.copy_(src)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rdst_amd.sort_device_tensor(keys.view(vdt), tmp.view(vdt), check=False)
        e1.record()
        torch.cuda.synchronize()
        rdst_amd.device_status()
        times.append(e0.elapsed_time(e1))
    k = torch.where(keys < 0, ~keys ^ (-(2**31)), keys) if vdt == torch.float32 else keys ^ (-(2**31))
    ok = bool((k[1:] >= k[:-1]).all()) and int(keys.sum()) == int(src.sum())
    print(f"{name:36s}: {min(times):8.3f} ms  {n / min(times) / 1e6:7.1f} Gkeys/s  ok={ok}  route={rdst_amd.last_route()}", flush=True)
    if os.environ.get("RDST_STAGES"):
        keys.copy_(src)
        rdst_amd.set_profiling(True)
        rdst_amd.sort_device_tensor(keys.view(vdt), tmp.view(vdt), check=False)
        p = rdst_amd.profile_run(-1, 4)
        rdst_amd.set_profiling(False)
        print("      " + "  ".join(f"{nm}{'' if lv is None else lv}={ms:.3f}" for nm, lv, ms in p["stages"] if ms >= 0.02), flush=True)
