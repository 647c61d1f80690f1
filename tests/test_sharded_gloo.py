"""The N>1 exchange logic of rdst_amd/sharded.py under gloo, world_size 2 and 3, on CPU.  The
local steps are served by a CPU engine built on the oracle (test infrastructure) so that the
all-gather / split / all-to-all / ordering logic — including the 16-bit split a skewed top byte
falls back to — is exercised exactly as on GPUs."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """CPU stand-in for HipEngine — tests only.  `dtype_name` is the key type; tensors carry the bits as signed
    integers (torch has no unsigned arithmetic for gloo)."""

    def __init__(self, dtype_name):
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle import oracle as O
        self.O = O
        self.dtype_name = dtype_name
        self.carrier = {4: np.int32, 8: np.int64}[np.dtype(dtype_name).itemsize]

    def _np(self, t):
        return t.numpy().view(self.dtype_name)

    def split_top_level(self, keys):
        a = self._np(keys)
        top = a.dtype.itemsize - 1
        dst, _ = self.O.out_of_place_sort(a, top)
        c, _, _, _ = self.O.get_counts_with_ends(a, top)
        return torch.from_numpy(dst.view(self.carrier)), torch.from_numpy(c.astype(np.int64))

    def split_top16(self, keys):
        from helpers import mapped_key
        a = self._np(keys)
        top = a.dtype.itemsize - 1
        mid, _ = self.O.out_of_place_sort(a, top - 1)          # two stable passes: levels L-2, L-1
        dst, _ = self.O.out_of_place_sort(mid, top)
        w = a.dtype.itemsize * 8
        prefix = (mapped_key(a) >> np.array(w - 16, dtype=f"uint{w}")).astype(np.int64)
        return torch.from_numpy(dst.view(self.carrier)), torch.from_numpy(np.bincount(prefix, minlength=65536).astype(np.int64))

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
        u = a.view(f"u{a.dtype.itemsize}")
        bits = 8 * a.dtype.itemsize
        if skew == "one_top_digit":    # every key in one top digit: the 8-bit split would hand one rank everything
            u &= np.array((1 << (bits - 8)) - 1, dtype=u.dtype)
        elif skew == "one_prefix16":   # every key in one 16-bit prefix: nothing 65 536 buckets can balance
            u &= np.array((1 << (bits - 16)) - 1, dtype=u.dtype)
        elif skew == "heavy_digit":    # 70 % of the keys in one top digit, the rest uniform
            heavy = np.random.default_rng(rank).random(a.size) < 0.7
            u[heavy] = (u[heavy] & np.array((1 << (bits - 8)) - 1, dtype=u.dtype)) | np.array(0x37 << (bits - 8), dtype=u.dtype)
        np.save(os.path.join(out_dir, f"in{rank}.npy"), a)
        # torch has no uint32/uint64 arithmetic on CPU for gloo; ship the bits as signed ints
        carrier = {4: np.int32, 8: np.int64}[a.dtype.itemsize]
        out, info = sharded_sort(torch.from_numpy(a.view(carrier).copy()), engine=OracleEngine(dtype_name), return_info=True)
        np.save(os.path.join(out_dir, f"out{rank}.npy"), out.numpy().view(dtype_name))
        assert sum(info["recv"]) == out.numel()
        assert info["split_bits"] == (16 if skew else 8), info["split_bits"]
        assert len(info["owner"]) == (65536 if skew else 256)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dtype_name,skew", [(2, "uint32", None), (2, "float32", None), (2, "int64", None),
                                                   (3, "uint64", None), (2, "uint32", "one_top_digit"),
                                                   (3, "uint32", "heavy_digit"), (2, "int64", "one_top_digit"),
                                                   (2, "uint32", "one_prefix16")])
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
    sizes = [o.size for o in outs]
    if skew != "one_prefix16":
        # near-equal ranges: on uniform keys from the 8-bit split, on a skewed top byte from the 16-bit one
        assert max(sizes) - min(sizes) < 0.1 * sum(sizes), sizes


def test_owner_map_device_form_equals_the_host_definition():
    sys.path.insert(0, ROOT)
    from rdst_amd.sharded import _owners, split_digits
    rng = np.random.default_rng(3)
    for world in (2, 3, 8):
        for buckets in (256, 65536):
            for shape in ("uniform", "skew", "sparse", "empty"):
                t = rng.integers(0, 1000, size=(world, buckets))
                if shape == "skew":
                    t[:, 7] += 10_000_000
                if shape == "sparse":
                    t[:, rng.random(buckets) < 0.95] = 0
                if shape == "empty":
                    t[:] = 0
                got = _owners(torch.from_numpy(t.astype(np.int64)), world).tolist()
                assert got == split_digits(t.sum(axis=0).tolist(), world), (world, buckets, shape)
