"""The N>1 exchange logic of rdst_amd/sharded.py under gloo, world_size 2 and 3, on CPU.  The
local steps are served by a CPU engine built on the oracle (test infrastructure) so that the
all-gather / split / all-to-all / ordering logic is exercised exactly as on GPUs."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """CPU stand-in for HipEngine — tests only."""

    def __init__(self):
        sys.path.insert(0, ROOT)
        from oracle import oracle as O
        self.O = O

    @staticmethod
    def _np(t):
        return t.numpy()

    def top_level_counts(self, keys):
        a = self._np(keys)
        c, _, _, _ = self.O.get_counts_with_ends(a, a.dtype.itemsize - 1)
        return c.astype(np.int64)

    def scatter_top_level(self, keys):
        a = self._np(keys)
        dst, _ = self.O.out_of_place_sort(a, a.dtype.itemsize - 1)
        return torch.from_numpy(dst)

    def sort(self, keys, tmp=None):
        self.O.sort(self._np(keys), threads=2)
        return keys

    def empty(self, n, like):
        return torch.empty(int(n), dtype=like.dtype)


def _worker(rank, world, port, dtype_name, per_rank, skew, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import random_bits
        from rdst_amd.sharded import sharded_sort
        a = random_bits(per_rank + rank * 17, dtype_name, seed=0x5D570005 + rank).copy()
        if skew:  # every key in one top digit: one rank must receive everything
            u = a.view(f"u{a.dtype.itemsize}")
            u &= np.array((1 << (8 * a.dtype.itemsize - 8)) - 1, dtype=u.dtype)
        np.save(os.path.join(out_dir, f"in{rank}.npy"), a)
        # torch has no uint32/uint64 arithmetic on CPU for gloo; ship the bits as signed ints
        carrier = {4: np.int32, 8: np.int64}[a.dtype.itemsize]

        class Eng(OracleEngine):
            def _np(self, t):
                return t.numpy().view(dtype_name)

            def scatter_top_level(self, keys):
                x = self._np(keys)
                dst, _ = self.O.out_of_place_sort(x, x.dtype.itemsize - 1)
                return torch.from_numpy(dst.view(carrier))

        out, info = sharded_sort(torch.from_numpy(a.view(carrier).copy()), engine=Eng(), return_info=True)
        np.save(os.path.join(out_dir, f"out{rank}.npy"), out.numpy().view(dtype_name))
        assert sum(info["recv"]) == out.numel()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dtype_name,skew", [(2, "uint32", False), (2, "float32", False), (2, "int64", False),
                                                   (3, "uint64", False), (2, "uint32", True)])
def test_sharded_sort_gloo(tmp_path, world, dtype_name, skew):
    from helpers import reference_sorted, same_bits
    port = 29500 + (os.getpid() + hash((world, dtype_name, skew))) % 2000
    per_rank = 60_000
    mp.spawn(_worker, args=(world, port, dtype_name, per_rank, skew, str(tmp_path)), nprocs=world, join=True)
    ins = [np.load(tmp_path / f"in{r}.npy") for r in range(world)]
    outs = [np.load(tmp_path / f"out{r}.npy") for r in range(world)]
    exp = reference_sorted(np.concatenate(ins))
    got = np.concatenate(outs)  # rank order == key order
    assert same_bits(got, exp)
    if not skew:
        sizes = [o.size for o in outs]
        assert max(sizes) - min(sizes) < 0.1 * sum(sizes)  # near-equal ranges on uniform keys
