"""Development aid: the atomic route's lowered bucket window on keys that share s top bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rdst_amd
n = 400_000_000
g = torch.Generator(device="cuda").manual_seed(5)
r = torch.randint(-(2**31), 2**31, (n,), dtype=torch.int32, device="cuda", generator=g)
for s in range(0, 10):
    mask = (1 << (32 - s)) - 1
    top = (0x5A5A5A5A >> (32 - s)) << (32 - s) if s else 0
    src = (r & mask) | top
    if src.dtype != torch.int32: src = src.to(torch.int32)
    keys = src.clone()
    rdst_amd.set_profiling(True)
    rdst_amd.sort_device_tensor(keys.view(torch.uint32))
    p = rdst_amd.profile_run(-1, 4)
    rdst_amd.set_profiling(False)
    k = keys ^ (-(2**31))
    ok = bool((k[1:] >= k[:-1]).all()) and int(keys.sum()) == int(src.sum())
    print(f"s={s} route={rdst_amd.last_route()} ok={ok} " + " ".join(f"{nm}{'' if lv is None else lv}={ms:.3f}" for nm, lv, ms in p["stages"] if ms >= 0.02), flush=True)
