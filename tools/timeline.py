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
# how often is "tile = blockIdx / 8 of chain blockIdx % 8" what the ticket hands out?  (rows of chain c start at c * nt)
rows = np.nonzero(rec[:, 9] != 0)[0]
nt = (n // 8 + 16895) // 16896
c, t = rows // nt, rows % nt
b = rec[rows, 11].astype(np.int64)
hit = (b % 8 == c) & (b // 8 == t)
near = (b % 8 == c) & (np.abs(b // 8 - t) <= 2)
print(f"ticket == block index / 8: {hit.mean() * 100:.1f} % of tiles; within 2: {near.mean() * 100:.1f} %; same chain: {(b % 8 == c).mean() * 100:.1f} %")
d2 = (b // 8 - t)[b % 8 == c]
print("offset histogram (block/8 - ticket):", {int(k): int(v) for k, v in zip(*np.unique(np.clip(d2, -6, 6), return_counts=True))})
x = rec[rows, 10].astype(np.int64) & 7
off = (x - b) % 8
print("XCC_ID - blockIdx mod 8:", {int(k): int(v) for k, v in zip(*np.unique(off, return_counts=True))})
order = np.argsort(r[:, 0] if False else rec[rows, 0])
# tickets against start order inside chain 0
m0 = c == 0
print("chain 0: first 24 (block/8, ticket) by ticket:", [(int(bb // 8), int(tt)) for bb, tt in sorted(zip(b[m0], t[m0]), key=lambda p: p[1])[:24]])
print("chain 0: (block/8, ticket) around 3000:", [(int(bb // 8), int(tt)) for bb, tt in sorted(zip(b[m0], t[m0]), key=lambda p: p[1])[3000:3012]])
