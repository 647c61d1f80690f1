"""The sharded route end to end on real hardware: two ranks that share the one GPU of the test box
run rdst_amd.sharded.sharded_sort with the product engine (HipEngine: the non-blocking split entries and the
device sort through the C ABI).  RCCL refuses two ranks on one device, so the two collectives go through gloo
(staged via the host inside sharded_sort); on a multi-GPU node the same code runs over RCCL (bench.py --gpus N).
Unmeasured on RCCL until the driver's multi-GPU run exists."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, dtype_name, per_rank, out_dir, skew=False, through_builder=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rdst_amd
        from helpers import random_bits, to_device, to_host
        from rdst_amd.sharded import sharded_sort
        a = random_bits(per_rank + 1000 * rank, dtype_name, seed=0x5D570005 + rank).copy()
        if skew:  # 80 % of the keys share one top byte: the 8-bit split cannot balance, the 16-bit one must
            u = a.view(f"u{a.dtype.itemsize}")
            bits = 8 * a.dtype.itemsize
            heavy = np.random.default_rng(rank).random(a.size) < 0.8
            u[heavy] = (u[heavy] & np.array((1 << (bits - 8)) - 1, dtype=u.dtype)) | np.array(0x42 << (bits - 8), dtype=u.dtype)
        np.save(os.path.join(out_dir, f"in{rank}.npy"), a.view(f"u{a.dtype.itemsize}"))
        if through_builder:  # the reference-shaped surface: a tuner that answers GpuSharded
            from rdst_amd.tuner import Algorithm, Tuner

            class Sharded(Tuner):
                def pick_algorithm(self, p, counts):
                    return Algorithm.GpuSharded

            out = rdst_amd.radix_sort_builder(to_device(a)).with_tuner(Sharded()).sort()
        else:
            out, info = sharded_sort(to_device(a), return_info=True)   # product engine
            assert sum(info["recv"]) == out.numel()
            assert info["split_bits"] == (16 if skew else 8)
            rdst_amd.device_status()
        np.save(os.path.join(out_dir, f"out{rank}.npy"), to_host(out, dtype_name).view(f"u{a.dtype.itemsize}"))
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


@pytest.mark.parametrize("dtype_name,through_builder", [("uint32", False), ("int64", False), ("float32", True)])
def test_two_ranks_one_gpu_skewed_top_byte_uses_the_16_bit_split(tmp_path, gpu, dtype_name, through_builder):
    import torch.multiprocessing as mp
    from helpers import reference_sorted, same_bits
    world, per_rank = 2, 2_000_000
    port = 29500 + (os.getpid() + 7 * len(dtype_name) + 3) % 2000
    mp.spawn(_worker, args=(world, port, dtype_name, per_rank, str(tmp_path), True, through_builder), nprocs=world, join=True)
    ins = [np.load(tmp_path / f"in{r}.npy").view(dtype_name) for r in range(world)]
    outs = [np.load(tmp_path / f"out{r}.npy").view(dtype_name) for r in range(world)]
    assert same_bits(np.concatenate(outs), reference_sorted(np.concatenate(ins)))
    assert abs(outs[0].size - outs[1].size) < 0.05 * (outs[0].size + outs[1].size)   # balanced although one top byte holds 80 %


def test_split_entries_match_the_oracle(gpu, oracle):
    """rdst_hip_split_top_level_device: one stable pass on the top level + its 256 counts on the device;
    rdst_hip_split_top16_device: ordered by the top 16 bits of the mapped key, stably, + the 65 536 bucket lengths."""
    from helpers import mapped_key, random_bits, same_bits, to_device, to_host
    from rdst_amd.sharded import HipEngine
    eng = HipEngine()
    for dtype in ("uint32", "float32", "int64", "uint16"):
        for n in (1, 77, 300_001):
            a = random_bits(n, dtype, seed=n).copy()
            top = a.dtype.itemsize - 1
            t = to_device(a)
            dst, counts = eng.split_top_level(t)
            exp, _ = oracle.out_of_place_sort(a, top)
            assert same_bits(to_host(dst, dtype), exp), (dtype, n)
            oc, _, _, _ = oracle.get_counts_with_ends(a, top)
            assert np.array_equal(counts.cpu().numpy().astype(np.uint64), oc)
            assert same_bits(to_host(t, dtype), a)   # source untouched
            srt, c16 = eng.split_top16(t)
            w = a.dtype.itemsize * 8
            prefix = (mapped_key(a) >> np.array(w - 16, dtype=f"uint{w}")).astype(np.int64)
            order = np.argsort(prefix, kind="stable")
            assert same_bits(to_host(srt, dtype), a[order]), (dtype, n)
            assert np.array_equal(c16.cpu().numpy(), np.bincount(prefix, minlength=65536)), (dtype, n)
    gpu.device_status()


def _nccl_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        import rdst_amd
        from helpers import random_bits, reference_sorted, same_bits, to_device, to_host
        from rdst_amd.sharded import sharded_sort
        for dtype_name in ("uint64", "uint32", "float32", "int64"):
            a = random_bits(2_000_003, dtype_name, seed=0x5D570040).copy()
            tm = {}
            out, info = sharded_sort(to_device(a), return_info=True, force_collectives=True, timings=tm)
            rdst_amd.device_status()
            assert info["recv"] == [a.size] and info["send"] == [a.size], info
            assert same_bits(to_host(out, dtype_name), reference_sorted(a)), dtype_name
            # the stage clock bench.py's N > 1 line reports (`sharded_breakdown`): every stage seen, none negative
            assert set(tm) == {"split", "counts_and_plan", "exchange", "local_sort"} and min(tm.values()) >= 0.0, tm
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_collectives_over_rccl_with_one_rank(tmp_path, gpu):
    """The box has one GPU and RCCL refuses two ranks on one device, so the all-gather and the all-to-all of the sharded route had
    never executed on the `nccl` backend.  A one-rank RCCL group runs them for real — device tensors, the int views,
    the split-size plumbing — and the result must be the plain sort of the shard.  (A single rank never owns more than its
    share, so the 16-bit split's table does not travel here.)"""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() + 77) % 2000
    mp.spawn(_nccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "ok").exists()
