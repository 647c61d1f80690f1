"""Development aid: per-stage device times of a 10^9-key sort, chain split on and off."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rdst_amd
from rdst_amd import radix_sort as rs

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
dt = sys.argv[2] if len(sys.argv) > 2 else "u32"
g = torch.Generator(device="cuda"); g.manual_seed(1)
if dt == "u32":
    src = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device="cuda", generator=g).view(torch.uint32)
else:
    src = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device="cuda", generator=g).view(torch.uint64)
dist = os.environ.get("RDST_DIST", "uniform")
if dt == "u32" and dist != "uniform":
    r = src.view(torch.int32)
    if dist == "bimodal":
        src = torch.cat([(r[: n // 2] >> 16) & 0xFFFF, r[n // 2:] << 16]).view(torch.uint32)
    elif dist == "twodigit":
        src = (r & ~0xFE00).view(torch.uint32)
    elif dist == "onegroup":
        src = (r & ~0xE0).view(torch.uint32)
    del r
keys = torch.empty_like(src); tmp = torch.empty_like(src)
ref = None
cfgs = [int(x) for x in os.environ.get("RDST_CFGS", "-1").split(",")]
vary = os.environ.get("RDST_VARY", "split")  # which knob the True/False column toggles: chain split or fast ranking
for cfg, split in [(c, sp) for c in cfgs for sp in (True, False, True, False)]:
    if vary == "fast":
        rs.set_tuning(cfg, 0, chain_split=True, fast_rank=split)
    else:
        rs.set_tuning(cfg, 0, chain_split=split)
    rs.set_profiling(True)
    for it in range(4):
        keys.copy_(src)
        rs.sort_device_tensor(keys, tmp, check=True)
    torch.cuda.synchronize()
    lv = 4 if dt == "u32" else 8
    pr = rs.profile_run(-1, lv)
    prof = [pr["clear"], pr["histogram"], pr["scan"]] + list(pr["passes"]) + [pr["copy_back"]]
    rs.set_profiling(False)
    k = keys.view(torch.int32 if dt == "u32" else torch.int64)
    chk = int(k[::997].sum().item())
    if ref is None: ref = chk
    print(f"cfg={cfg:2d} split={split!s:5s} total={sum(prof):7.3f} ms stages=" + " ".join(f"{x:.3f}" for x in prof) + ("" if chk == ref else "  CHECKSUM DIFFERS"), flush=True)
