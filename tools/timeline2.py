"""Development aid: where a workgroup's life goes in the 8-byte K4 (which = 1), msd_scatter_kernel pass A / B (2 / 3) or the
expanding 4-byte K4 (5): thread 0's shader-clock stamps, from a -DRDST_EXPERIMENTS build (tools/_build/librdst_hip_exp.so).
    python tools/timeline2.py <uint32|uint64|float32n|uint32g> <which> [n]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rdst_amd import _lib
_lib.LIB_PATH = os.environ.get("RDST_HIP_LIB") or os.path.join(ROOT, "tools", "_build", "librdst_hip_exp.so")
import rdst_amd
lib = _lib.load()
name, which = sys.argv[1], int(sys.argv[2])
n = int(float(sys.argv[3])) if len(sys.argv) > 3 else 10**9
g = torch.Generator(device="cuda").manual_seed(1)
if name == "float32n":   # a float column: normal(0, 1), sorted as f32 (the hybrid route's exact MSD passes)
    name, it = "float32", torch.int32
    src = torch.randn(n, dtype=torch.float32, device="cuda", generator=g).view(torch.int32)
elif name == "uint32g":  # a bell over the whole range (sum of four uniforms): three quarters of the keys in buckets of 17 409 .. 45 000
    name, it = "uint32", torch.int32
    r0 = torch.randint(-(2**31), 2**31, (n,), dtype=torch.int32, device="cuda", generator=g)
    src = (r0 >> 2) + (torch.roll(r0, 1) >> 2) + (torch.roll(r0, 2) >> 2) + (torch.roll(r0, 3) >> 2)
    del r0
else:
    it = torch.int32 if name == "uint32" else torch.int64
    info = torch.iinfo(it)
    src = torch.randint(info.min, info.max, (n,), dtype=it, device="cuda", generator=g)
keys, tmp = src.clone(), torch.empty_like(src)
view, tview = keys.view(getattr(torch, name)), tmp.view(getattr(torch, name))
rdst_amd.sort_device_tensor(view, tview)
rows = 400_000
keys.copy_(src)
lib.rdst_hip_exp_timeline_select(ctypes.c_uint32(which))
lib.rdst_hip_exp_timeline(None, ctypes.c_uint64(rows))
rdst_amd.sort_device_tensor(view, tview)
rec = np.zeros((rows, 12), dtype=np.uint32)
lib.rdst_hip_exp_timeline(rec.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ctypes.c_uint64(rows))
last = 9 if which == 1 else 8
r = rec[rec[:, last] != 0].astype(np.int64)
names = {1: ["loads issued + tables zeroed", "barrier (loads land)", "count (returning LDS adds)", "barrier + scan + prefixes + barrier", "place (slots, staged bits)",
             "barrier + ties", "barrier + stage half 0 + barrier", "store half 0 + barrier", "stage + store half 1"],
         2: ["loads issued", "loads land + inversion test + count", "barrier", "digit sums + scan + claims", "barrier", "place into LDS", "barrier", "scatter stores issued"],
         3: ["loads issued", "loads land + count", "barrier", "digit sums + scan + claims", "barrier", "place into LDS", "barrier", "scatter stores issued"],
         5: ["table zeroed + barrier", "loads + count", "barrier", "scan + prefixes + barrier", "expand (windows)", "barrier", "-", "-"]}[which]
d = (r[:, 1:last + 1] - r[:, 0:last]) & 0xFFFFFFFF
print(f"{name} kernel {which}: {len(r)} workgroups; shader clocks per phase as thread 0 sees them (mean / p50 / p90), us at 2.4 GHz:")
for k, nm in enumerate(names):
    print(f"  {nm:40s} {d[:, k].mean():8.0f} {np.median(d[:, k]):8.0f} {np.percentile(d[:, k], 90):8.0f}   {d[:, k].mean() / 2400:6.2f} us")
tot = (r[:, last] - r[:, 0]) & 0xFFFFFFFF
print(f"  {'whole workgroup':40s} {tot.mean():8.0f} {np.median(tot):8.0f} {np.percentile(tot, 90):8.0f}   {tot.mean() / 2400:6.2f} us")
