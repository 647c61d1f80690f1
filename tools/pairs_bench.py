"""Development aid: device time of key-value sorts (rdst_hip_sort_pairs_device)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rdst_amd

g = torch.Generator(device="cuda"); g.manual_seed(3)
for kt, vt, n in ((torch.int32, torch.int32, 10**9), (torch.int32, torch.int64, 5 * 10**8), (torch.int64, torch.int32, 5 * 10**8),
                  (torch.int64, torch.int64, 5 * 10**8)):
    info = torch.iinfo(kt)
    src = torch.randint(info.min, info.max, (n,), dtype=kt, device="cuda", generator=g)
    ut = torch.uint32 if kt == torch.int32 else torch.uint64
    keys = torch.empty_like(src); tk = torch.empty_like(src)
    vals = torch.empty(n, dtype=vt, device="cuda"); tv = torch.empty_like(vals)
    base = torch.arange(n, dtype=vt, device="cuda")
    times = []
    for _ in range(4):
        keys.copy_(src); vals.copy_(base)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rdst_amd.sort_pairs_device_tensor(keys.view(ut), vals, tk.view(ut), tv, check=False)
        e1.record(); torch.cuda.synchronize()
        rdst_amd.device_status()
        times.append(e0.elapsed_time(e1))
    t = min(times[1:])
    kb, vb = src.element_size(), vals.element_size()
    L = kb
    alg = n * (kb * (2 * L + 1) + vb * 2 * L)
    print(f"{kb}-byte keys + {vb}-byte values, n={n:.0e}: {t:8.3f} ms  {n / t / 1e6:6.1f} Gpairs/s  {alg / t / 1e9:6.2f} TB/s algorithmic", flush=True)
    del src, keys, tk, vals, tv, base
