"""Development aid: numbers quoted in DESIGN.md that bench.py does not print — the host entry
point's end-to-end rate (PCIe included) and the other key types at 1 B keys."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import rdst_amd
from gpu_sanity import time_sort

rng = np.random.default_rng(3)
a = rng.integers(0, 1 << 32, size=1_000_000_000, dtype=np.uint32)
rdst_amd.radix_sort_unstable(a[:1000].copy())
for _ in range(2):
    b = a.copy()
    t0 = time.perf_counter()
    rdst_amd.radix_sort_unstable(b)          # rdst_hip_sort: hipMalloc + H2D + sort + D2H
    dt = time.perf_counter() - t0
    print(f"host entry point, 1e9 u32 (pageable numpy buffer): {dt*1e3:.1f} ms = {1e9/dt/1e9:.2f} Gkeys/s", flush=True)
assert (b[1:] >= b[:-1]).all()
del a, b
for dtype in (np.int32, np.int64, np.float32, np.float64, np.uint64):
    time_sort(1_000_000_000, dtype, iters=3)
