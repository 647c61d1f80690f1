import sys, numpy as np, torch
sys.path.insert(0,'.')
import rdst_amd
sys.path.insert(0,'tools')
from gpu_sanity import time_sort
for dtype in (np.uint32, np.uint64):
    for cfg in (0,1,2,3):
        rdst_amd.set_tuning(cfg,0)
        print("cfg",cfg, end=" ")
        time_sort(1_000_000_000, dtype, iters=3)
