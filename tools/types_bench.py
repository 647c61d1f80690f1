"""Development aid: device sort time of 10^9 keys for every headline key type."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rdst_amd
g = torch.Generator(device="cuda"); g.manual_seed(2)
for name, it, n in (("uint32", torch.int32, 10**9), ("int32", torch.int32, 10**9), ("float32", torch.int32, 10**9),
                    ("uint64", torch.int64, 10**9), ("int64", torch.int64, 10**9), ("float64", torch.int64, 10**9),
                    ("uint16", torch.int16, 10**9), ("uint8", torch.int8, 10**9)):
    info = torch.iinfo(it)
    src = torch.randint(info.min, info.max, (n,), dtype=it, device="cuda", generator=g)
    keys = torch.empty_like(src); tmp = torch.empty_like(src)
    ts = []
    for _ in range(4):
        keys.copy_(src); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rdst_amd.sort_device_tensor(keys.view(getattr(torch, name)), tmp.view(getattr(torch, name)), check=False); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    rdst_amd.device_status()
    t = min(ts[1:]); k = src.element_size()
    print(f"{name:8s} {t:8.3f} ms  {n / t / 1e6:7.1f} Gkeys/s  {n * k * (2 * k + 1) / t / 1e9:5.2f} TB/s algorithmic", flush=True)
    del src, keys, tmp
