import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rdst_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.join(ROOT, sys.argv[1])
import torch, rdst_amd
from rdst_amd import radix_sort as rs
n = 10**7
g = torch.Generator(device="cuda"); g.manual_seed(3)
src = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device="cuda", generator=g)
fails = 0
for it in range(12):
    keys = src.clone(); vals = torch.arange(n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize(); t0 = time.time()
    rs.set_profiling(True)
    try:
        rdst_amd.sort_pairs_device_tensor(keys.view(torch.uint32), vals)
        msg = "ok"
    except Exception as e:
        msg = str(e)[-70:]
    torch.cuda.synchronize()
    pr = rs.profile_run(-1, 4)
    rs.set_profiling(False)
    fails += msg != "ok"
print(sys.argv[1] if len(sys.argv) > 1 else "default", "failures:", fails, "of 12", flush=True)
