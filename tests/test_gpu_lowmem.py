"""The low-memory device route (rdst_hip_sort_device_lowmem: a Regions-sort level with block swaps, then ordinary sorts
of what fits the scratch; rdst_hip_partition_device: partition_index) against the same unique answer as every other
route.  Reference: src/sorts/regions_sort.rs:51-286, src/sorts/ska_sort.rs:28-113 (the per-tile step),
src/sort_utils.rs:295-331, src/tuners/low_memory_tuner.rs:13-43, src/radix_sort_builder.rs:74-77."""
import numpy as np
import pytest

from helpers import DTYPES, mapped_key, random_bits, reference_sorted, same_bits, to_device, to_host

pytestmark = pytest.mark.gpu
SCRATCH = 65536  # the smallest scratch the entry accepts: many tiles already at a few million keys


def _lowmem(rdst, a, scratch_len=SCRATCH):
    import torch
    t = to_device(a)
    scratch = torch.empty(scratch_len, dtype=t.dtype, device=t.device)
    rdst.sort_device_tensor_lowmem(t, scratch)
    return to_host(t, a.dtype)


@pytest.mark.parametrize("dtype", DTYPES + ("uint16", "int8"))
def test_lowmem_matches_the_reference_order_over_sizes(gpu, dtype):
    for i, n in enumerate((2, 1000, SCRATCH - 1, SCRATCH, SCRATCH + 1, 3 * SCRATCH + 17, 1_000_003, 3_000_001)):
        a = random_bits(n, dtype, seed=500 + i).copy()
        assert same_bits(_lowmem(gpu, a), reference_sorted(a)), (dtype, n)


@pytest.mark.parametrize("dtype", ("uint32", "int64", "float32"))
def test_lowmem_skewed_inputs_recurse_to_lower_levels(gpu, dtype):
    """A digit that holds more keys than the scratch takes another Regions level on the next digit; all-equal keys go all
    the way down and stop."""
    n = 2_000_000
    w = np.dtype(dtype).itemsize * 8
    rng = np.random.default_rng(9)
    base = random_bits(n, f"uint{w}", seed=61).copy()
    top = np.array(0xFF << (w - 8), dtype=base.dtype)
    cases = {
        "90% one top byte": np.where(rng.random(n) < 0.9, (base & ~top) | np.array(0x42 << (w - 8), dtype=base.dtype), base),
        "one top byte": (base & ~top) | np.array(0x17 << (w - 8), dtype=base.dtype),
        "two top bytes values": base & ~(np.array(0xFE << (w - 8), dtype=base.dtype)),
        "top three bytes constant": (base & np.array((1 << (w - 24)) - 1, dtype=base.dtype)) | np.array(0xABCDEF << (w - 24), dtype=base.dtype),
        "all equal": np.full(n, 0x1234567, dtype=base.dtype),
        "sorted": np.sort(base),
        "reverse sorted": np.sort(base)[::-1].copy(),
    }
    for name, u in cases.items():
        a = u.view(dtype)
        assert same_bits(_lowmem(gpu, a), reference_sorted(a)), (dtype, name)


def test_lowmem_scratch_sizes_and_builder(gpu):
    import torch
    a = random_bits(5_000_011, "uint32", seed=71).copy()
    exp = reference_sorted(a)
    for scratch in (65536, 100_000, 1 << 20, 5_000_011, 8_000_000):   # odd sizes are rounded down to whole 4096s; >= n: one ordinary sort
        assert same_bits(_lowmem(gpu, a, scratch), exp), scratch
    t = to_device(a)
    gpu.radix_sort_builder(t).with_low_mem_tuner().sort()              # src/radix_sort_builder.rs:74-77
    assert same_bits(to_host(t, "uint32"), exp)
    h = a.copy()
    gpu.radix_sort_builder(h).with_low_mem_tuner().sort()              # a host slice takes the same route through a device copy
    assert same_bits(h, exp)
    with pytest.raises(gpu.RdstHipError):
        gpu.sort_device_tensor_lowmem(to_device(a), torch.empty(1000, dtype=torch.uint32, device="cuda"))


@pytest.mark.parametrize("dtype", ("uint32", "float64"))
def test_partition_device_is_partition_index(gpu, dtype):
    """src/sort_utils.rs:295-331: elements for which the predicate holds first, returns the split index; here the predicate is
    'digit `level` == `digit`' (what Ska / Scanning / Regions use it for: moving the largest bucket aside)."""
    import torch
    levels = np.dtype(dtype).itemsize
    w = levels * 8
    for n in (1, 70_000, 1_000_003):
        a = random_bits(n, dtype, seed=n).copy()
        for level, digit in ((levels - 1, 0x80), (0, 0x00), (levels // 2, 0xFF)):
            if n > 1:  # make the chosen digit common
                u = a.view(f"uint{w}")
                m = mapped_key(a)
                pick = np.random.default_rng(level).random(n) < 0.3
                # set digit `level` of the MAPPED key to `digit` for the picked ones, then map back (the map is an involution on bits per sign)
                sh = np.array(8 * level, dtype=m.dtype)
                m2 = np.where(pick, (m & ~(np.array(0xFF, dtype=m.dtype) << sh)) | (np.array(digit, dtype=m.dtype) << sh), m)
                a = _unmap(m2, dtype)
            t = to_device(a)
            split = gpu.partition_device(t, level, digit, torch.empty(SCRATCH, dtype=t.dtype, device="cuda"))
            got = to_host(t, dtype)
            dg = (mapped_key(got) >> np.array(8 * level, dtype=f"uint{w}")) & np.array(0xFF, dtype=f"uint{w}")
            want = int(((mapped_key(a) >> np.array(8 * level, dtype=f"uint{w}")) & np.array(0xFF, dtype=f"uint{w}") == digit).sum())
            assert split == want, (dtype, n, level)
            assert (dg[:split] == digit).all() and (dg[split:] != digit).all()
            assert same_bits(reference_sorted(got), reference_sorted(a))   # same multiset


def _unmap(m, dtype):
    """inverse of helpers.mapped_key"""
    dt = np.dtype(dtype)
    w = dt.itemsize * 8
    msb = np.array(1 << (w - 1), dtype=m.dtype)
    if dt.kind == "u":
        return m.view(dtype)
    if dt.kind == "i":
        return (m ^ msb).view(dtype)
    neg = (m >> np.array(w - 1, dtype=m.dtype)) == 0      # mapped keys below the midpoint were negative
    return np.where(neg, ~m, m ^ msb).view(dtype)


def test_lowmem_one_billion_keys_with_a_64th_of_the_memory(gpu):
    import torch
    n = 1_000_000_000
    g = torch.Generator(device="cuda").manual_seed(0x5D570009)
    src = torch.randint(-(2**31), 2**31, (n,), dtype=torch.int32, device="cuda", generator=g)
    keys = src.clone()
    gpu.sort_device_tensor_lowmem(keys.view(torch.uint32))            # scratch: n / 64 elements = 62.5 MB for 4 GB of keys
    assert int(keys.sum()) == int(src.sum())
    assert int((keys ^ (keys >> 11)).sum()) == int((src ^ (src >> 11)).sum())
    m = keys ^ torch.iinfo(torch.int32).min
    for s in range(0, n, 1 << 27):
        e = min(n, s + (1 << 27) + 1)
        assert bool((m[s + 1:e] >= m[s:e - 1]).all())
