"""Development aid: where a tile's life goes (thread 0's shader-clock stamps along pass 0), from a
-DRDST_EXPERIMENTS build."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rdst_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_build", "librdst_hip_exp.so")
import rdst_amd
lib = _lib.load()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
g = torch.Generator(device="cuda").manual_seed(1)
src = torch.randint(-(2**31), 2**31, (n,), dtype=torch.int32, device="cuda", generator=g)
keys, tmp = src.clone(), torch.empty_like(src)
rdst_amd.sort_device_tensor(keys.view(torch.uint32), tmp.view(torch.uint32))
tiles = n // 16896 + 12
keys.copy_(src)
lib.rdst_hip_exp_timeline(None, ctypes.c_uint64(tiles))
rdst_amd.sort_device_tensor(keys.view(torch.uint32), tmp.view(torch.uint32))
rec = np.zeros((tiles, 12), dtype=np.uint32)
lib.rdst_hip_exp_timeline(rec.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ctypes.c_uint64(tiles))
r = rec[rec[:, 9] != 0].astype(np.int64)
names = ["entry->ticket barrier", "ticket->loads issued", "loads+count", "barrier 1", "scan (2 barriers)", "rank", "look-back", "barrier 4", "scatter"]
d = (r[:, 1:10] - r[:, 0:9]) & 0xFFFFFFFF
print(f"{len(r)} tiles; clocks per phase as seen by thread 0 (mean / p50 / p90):")
for k, nm in enumerate(names):
    print(f"  {nm:24s} {d[:, k].mean():8.0f} {np.median(d[:, k]):8.0f} {np.percentile(d[:, k], 90):8.0f}")
tot = (r[:, 9] - r[:, 0]) & 0xFFFFFFFF
print(f"  {'whole tile':24s} {tot.mean():8.0f} {np.median(tot):8.0f} {np.percentile(tot, 90):8.0f}")
