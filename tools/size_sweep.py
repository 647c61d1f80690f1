"""Development aid: per-pass time against n (does a working set that fits the Infinity Cache run faster?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rdst_amd import radix_sort as rs

g = torch.Generator(device="cuda"); g.manual_seed(1)
for n in [1 << 20, 1 << 22, 1 << 23, 1 << 24, 1 << 25, 1 << 26, 1 << 27, 1 << 28, 10**9]:
    src = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device="cuda", generator=g).view(torch.uint32)
    keys = torch.empty_like(src); tmp = torch.empty_like(src)
    rs.set_profiling(True)
    for it in range(6):
        keys.copy_(src)
        rs.sort_device_tensor(keys, tmp, check=False)
    torch.cuda.synchronize()
    pr = rs.profile_run(-1, 4)
    rs.set_profiling(False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(5): keys.copy_(src)
    e1.record(); torch.cuda.synchronize()
    cp = e0.elapsed_time(e1) / 5
    p = sum(pr["passes"]) / 4
    print(f"n={n:>11d} ({n*4/2**20:8.1f} MiB)  hist {pr['histogram']:.4f} ms  pass {p:.4f} ms = {n*8/p/1e9:7.1f} GB/s  copy {cp:.4f} ms = {n*8/cp/1e9:7.1f} GB/s  clear {pr['clear']:.4f}", flush=True)
    del src, keys, tmp
