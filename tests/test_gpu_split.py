"""4- and 8-byte keys beyond the atomic route's window (run_split_sort, rdst_kernels.hip): one exact scatter pass on the top byte, then
aligned groups of top bytes as slices of their own, each delivered straight into the caller's array.  The reference recurses
the same way where a bucket is too big for the sort at hand (src/sorter.rs:131-138, src/sorts/recombinating_sort.rs:68-88).
Mode 17 forces the split at every length (eight groups of 32 top bytes), so the parts, their offsets and every route a part can
end on are checked bit-exactly at sizes torch.sort handles; the default setting is checked at 2^30 + 12 345 keys by its
size-independent properties."""
import numpy as np
import pytest

from helpers import mapped_key, random_bits, reference_sorted, same_bits, to_device, to_host

pytestmark = pytest.mark.gpu

WIDE = ("uint32", "int32", "float32", "uint64", "int64", "float64")


@pytest.fixture
def split(gpu):
    gpu.set_hybrid(17, 1)
    yield gpu
    gpu.set_hybrid(True, 0)
    gpu.device_status()


@pytest.mark.parametrize("dtype", WIDE)
def test_split_matches_the_reference_order_over_sizes(split, dtype):
    for i, n in enumerate((2, 3, 255, 256, 1_000, 100_003, 1_500_000)):
        a = random_bits(n, dtype, 0x5D570400 + i)
        d = to_device(a)
        split.sort_device_tensor(d)
        assert same_bits(to_host(d, dtype), reference_sorted(a)), (dtype, n)


@pytest.mark.parametrize("dtype", WIDE)
def test_split_with_empty_heavy_and_single_key_groups(split, dtype):
    """groups of top bytes that are empty, hold one key, or hold nearly everything (a part whose areas overflow falls to the
    next route inside the part; the result must still land in the caller's array)"""
    n = 3_000_000
    a = random_bits(n, dtype, 0x5D570410).copy()
    ut = np.dtype(f"u{a.dtype.itemsize}").type
    sh = 8 * a.dtype.itemsize - 8
    u = a.view(ut)
    u[: n - n // 50] = (u[: n - n // 50] & ut((1 << sh) - 1)) | ut(0x47 << sh)   # 98 % on one top byte
    u[n - 5] = ut(0x01 << sh) | ut(5)                                             # a group with one key
    for order in ("random", "sorted", "reversed"):
        b = a if order == "random" else reference_sorted(a) if order == "sorted" else reference_sorted(a)[::-1].copy()
        d = to_device(b)
        split.sort_device_tensor(d)
        assert same_bits(to_host(d, dtype), reference_sorted(a)), (dtype, order)
    c = np.full(100_000, a[0], dtype=a.dtype)                                                         # all equal: one group, sorted already
    d = to_device(c)
    split.sort_device_tensor(d)
    assert same_bits(to_host(d, dtype), c)


def test_split_at_lengths_the_atomic_route_takes_per_part(split):
    """4 x 10^7 u64 keys in eight parts of 5 x 10^6 against torch.sort; then the parts' own thresholds: with the real threshold
    (2^26 keys) the parts are short slices and go the LSD way — delivered into the caller's array all the same"""
    import torch
    n = 40_000_000
    g = torch.Generator(device="cuda").manual_seed(0x5D570420)
    src = torch.randint(-(2**63), 2**63 - 1, (n,), dtype=torch.int64, device="cuda", generator=g)
    want = torch.sort(src ^ torch.iinfo(torch.int64).min).values
    for min_len in (1, 0):
        split.set_hybrid(17, min_len)
        keys = src.clone()
        split.sort_device_tensor(keys.view(torch.uint64))
        assert bool(torch.equal(want, keys ^ torch.iinfo(torch.int64).min)), min_len
        # (parts below 2^26 keys have no sample to lower the window past the group's shared top bits: an area overflows and the
        # part falls to the K1h hybrid route; from 2^26 keys up — every part of a real split — the sample does, see below)
        assert split.last_route() == ("hybrid" if min_len else "lsd")


def test_two_pow_30_u64_keys_take_the_split_by_default(gpu):
    """2^30 + 12 345 uniform u64 keys (a uniform bucket no longer fits K4's tile): the default setting splits on the top byte
    and the parts take the atomic route.  Checked by sortedness, a checksum over the multiset, and against the LSD-only setting
    on a prefix and a suffix."""
    import torch
    n = (1 << 30) + 12_345
    g = torch.Generator(device="cuda").manual_seed(0x5D570430)
    src = torch.randint(-(2**63), 2**63 - 1, (n,), dtype=torch.int64, device="cuda", generator=g)
    keys = src.clone()
    gpu.sort_device_tensor(keys.view(torch.uint64))
    assert gpu.last_route() == "atomic"
    k = keys ^ torch.iinfo(torch.int64).min
    assert bool((k[1:] >= k[:-1]).all())
    del k
    assert int(keys.sum()) == int(src.sum()) and int((keys ^ (keys >> 17)).sum()) == int((src ^ (src >> 17)).sum())
    gpu.set_hybrid(False)
    try:
        lsd = src
        gpu.sort_device_tensor(lsd.view(torch.uint64))
        assert gpu.last_route() == "lsd"
        assert bool(torch.equal(lsd, keys))
    finally:
        gpu.set_hybrid(True, 0)
    gpu.device_status()


def test_u32_keys_beyond_the_window_take_the_split_by_default(gpu):
    """1.4 x 10^9 uniform u32 keys: past 1.3 x 10^9 the split (12 + 20 bytes per key) beats the K1h hybrid route (24 and every
    bucket the expanding K4's)"""
    import torch
    n = 1_400_000_000
    g = torch.Generator(device="cuda").manual_seed(0x5D570440)
    src = torch.randint(-(2**31), 2**31, (n,), dtype=torch.int32, device="cuda", generator=g)
    keys = src.clone()
    gpu.sort_device_tensor(keys.view(torch.uint32))
    assert gpu.last_route() == "atomic"
    k = keys ^ torch.iinfo(torch.int32).min
    assert bool((k[1:] >= k[:-1]).all())
    del k
    assert int(keys.sum()) == int(src.sum()) and int((keys ^ (keys >> 9)).sum()) == int((src ^ (src >> 9)).sum())
    gpu.device_status()
