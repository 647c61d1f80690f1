"""Development aid: host entry point (rdst_hip_sort: alloc + H2D + sort + D2H) against the CPU oracle
(StandardTuner route, all host cores) over n — where a device-aware tuner should switch."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch  # noqa: F401  (one HIP runtime)
import rdst_amd
import oracle as orc

threads = len(os.sched_getaffinity(0))
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()
    if q != "max": threads = max(1, min(threads, int(q) // int(p)))
except OSError:
    pass
rng = np.random.default_rng(1)
print(f"threads={threads}")
for dtype in ("uint32", "uint64"):
    for n in (10_000, 30_000, 100_000, 300_000, 1_000_000, 3_000_000, 10_000_000, 30_000_000, 100_000_000):
        a = rng.integers(0, np.iinfo(dtype).max, size=n, dtype=dtype)
        tg, tc = [], []
        for _ in range(4):
            b = a.copy(); t = time.perf_counter(); rdst_amd.radix_sort_unstable(b); tg.append(time.perf_counter() - t)
            c = a.copy(); t = time.perf_counter(); orc.sort(c, threads=threads); tc.append(time.perf_counter() - t)
        assert np.array_equal(b, c)
        print(f"{dtype} n={n:>11d}: device route {min(tg[1:])*1e3:9.3f} ms   cpu oracle {min(tc[1:])*1e3:9.3f} ms   ratio {min(tc[1:])/min(tg[1:]):6.2f}", flush=True)
