"""Development aid: whole-sort device time for small and mid-size inputs, chain split on and off."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rdst_amd import radix_sort as rs
g = torch.Generator(device="cuda"); g.manual_seed(1)
for n in [1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24, 1 << 26]:
    src = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device="cuda", generator=g).view(torch.uint32)
    keys = torch.empty_like(src); tmp = torch.empty_like(src)
    out = []
    for split in (True, False):
        rs.set_tuning(chain_split=split)
        ts = []
        for it in range(12):
            keys.copy_(src); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rs.sort_device_tensor(keys, tmp, check=False); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        rs.set_profiling(True); keys.copy_(src); rs.sort_device_tensor(keys, tmp, check=False); torch.cuda.synchronize()
        pr = rs.profile_run(-1, 4) if rs.profile_runs() else None; rs.set_profiling(False)
        det = f"(clear {pr['clear']*1e3:.0f} hist {pr['histogram']*1e3:.0f} scan {pr['scan']*1e3:.0f} passes {sum(pr['passes'])*1e3:.0f})" if pr else "(one-workgroup sort)"
        out.append(f"split={split!s:5s} {min(ts[2:])*1e3:8.1f} us {det}")
    print(f"n={n:>9d}: " + " | ".join(out), flush=True)
rs.set_tuning()
