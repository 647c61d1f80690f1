"""The sharded route end to end on real hardware: two ranks that share the one GPU of the test box
run rdst_amd.sharded.sharded_sort with the product engine (HipEngine: K6 histogram, K3 top-digit
split, device LSD sort through the C ABI).  RCCL refuses two ranks on one device, so the two
collectives go through gloo (staged via the host inside sharded_sort); on a multi-GPU node the same
code runs over RCCL (bench.py --gpus N)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, dtype_name, per_rank, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import random_bits, to_device, to_host
        from rdst_amd.sharded import sharded_sort
        a = random_bits(per_rank + 1000 * rank, dtype_name, seed=0x5D570005 + rank).copy()
        np.save(os.path.join(out_dir, f"in{rank}.npy"), a.view(f"u{a.dtype.itemsize}"))
        out, info = sharded_sort(to_device(a), return_info=True)   # product engine
        np.save(os.path.join(out_dir, f"out{rank}.npy"), to_host(out, dtype_name).view(f"u{a.dtype.itemsize}"))
        assert sum(info["recv"]) == out.numel()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype_name", ["uint32", "float32", "uint64"])
def test_two_ranks_one_gpu(tmp_path, gpu, dtype_name):
    import torch.multiprocessing as mp
    from helpers import reference_sorted, same_bits
    world, per_rank = 2, 3_000_000
    port = 29500 + (os.getpid() + len(dtype_name)) % 2000
    mp.spawn(_worker, args=(world, port, dtype_name, per_rank, str(tmp_path)), nprocs=world, join=True)
    ins = [np.load(tmp_path / f"in{r}.npy").view(dtype_name) for r in range(world)]
    outs = [np.load(tmp_path / f"out{r}.npy").view(dtype_name) for r in range(world)]
    assert same_bits(np.concatenate(outs), reference_sorted(np.concatenate(ins)))
    assert abs(outs[0].size - outs[1].size) < 0.05 * (outs[0].size + outs[1].size)
