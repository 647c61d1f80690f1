"""BASELINE.json's full sizes (configs[1..3]: 1 B u32 / u64 / f32 on one MI355X) through
size-independent properties: the output is non-decreasing in mapped-key order, it is the same
multiset as the input (two independent checksums + every level's 256-bin histogram is
unchanged), and sorting is idempotent.  Inputs are generated on the device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N = 1_000_000_000


def _gen(torch, n, itype, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    info = torch.iinfo(itype)
    return torch.randint(info.min, info.max, (n,), dtype=itype, device="cuda", generator=g)


def _mapped(torch, x, kind):
    """signed-comparable image of the key order"""
    mn = torch.iinfo(x.dtype).min
    if kind == "u":
        return x ^ mn
    if kind == "i":
        return x
    return torch.where(x < 0, ~x ^ mn, x)


def _is_sorted(torch, m, chunk=1 << 27):
    for s in range(0, m.numel(), chunk):
        e = min(m.numel(), s + chunk + 1)
        if not bool((m[s + 1:e] >= m[s:e - 1]).all()):
            return False
    return True


@pytest.mark.parametrize("name,itype_name,seed,route", [("uint32", "int32", 0x5D570002, "atomic"), ("uint32", "int32", 0x5D570002, "hybrid"),
                                                         ("uint32", "int32", 0x5D570002, "lsd"), ("uint64", "int64", 0x5D570003, "atomic"),
                                                         ("uint64", "int64", 0x5D570003, "hybrid"),
                                                         ("float32", "int32", 0x5D570004, "atomic"), ("float64", "int64", 0x5D570007, "atomic")])
def test_one_billion_keys(gpu, name, itype_name, seed, route):
    """`route`: uniform 10^9-key slices take the atomic route by the device's own choice (asserted); the K1h hybrid route and
    the LSD route — every skewed input's route — are forced once at full size too."""
    import torch
    itype = getattr(torch, itype_name)
    src = _gen(torch, N, itype, seed)
    keys = src.clone()
    view = keys.view(getattr(torch, name))
    before = gpu.all_level_counts(view)
    gpu.set_hybrid({"lsd": False, "hybrid": 7, "atomic": True}[route])
    try:
        gpu.sort_device_tensor(view)
        assert gpu.last_route() == route
    finally:
        gpu.set_hybrid(True)
    after = gpu.all_level_counts(view)
    assert np.array_equal(before, after)                       # same multiset, digit by digit
    assert int(keys.sum()) == int(src.sum())                   # wrapping checksum
    assert int((keys ^ (keys >> 11)).sum()) == int((src ^ (src >> 11)).sum())
    m = _mapped(torch, keys, np.dtype(name).kind)
    assert _is_sorted(torch, m)
    del m
    once = keys.clone()
    gpu.sort_device_tensor(view)                                # idempotence (no inversion: no pass, no local sort)
    assert bool((keys == once).all())
    del once, src
    # a slice small enough for the oracle-free numpy check, bit-exact against an independent sort
    head = keys[:2_000_000].cpu().numpy().view(name)
    from helpers import mapped_key
    k = mapped_key(head)
    assert (k[1:] >= k[:-1]).all()


def test_half_a_billion_key_value_pairs(gpu):
    """5·10^8 (u32 key, i32 row index) pairs: keys non-decreasing, every value still names the row its key
    came from, the values are a permutation, equal keys keep their input order."""
    import torch
    n = 500_000_000
    src = _gen(torch, n, torch.int32, 0x5D570006) & 0x3FFFFFF   # 2^26 distinct keys: every key repeats
    keys = src.clone()
    idx = torch.arange(n, dtype=torch.int32, device="cuda")
    gpu.sort_pairs_device_tensor(keys.view(torch.uint32), idx)
    assert _is_sorted(torch, keys)                               # non-negative values: signed order == unsigned order
    assert bool((src[idx.long()] == keys).all())
    assert int(idx.sum(dtype=torch.int64)) == n * (n - 1) // 2
    same = keys[1:] == keys[:-1]
    assert bool((idx[1:][same] > idx[:-1][same]).all())          # stable


def test_more_than_2_pow_30_keys_uses_wide_status_words(gpu):
    """n >= 2^30: the look-back prefix no longer fits 30 bits -> 64-bit status words."""
    import torch
    n = (1 << 30) + 12_345
    src = _gen(torch, n, torch.int32, 99)
    keys = src.clone()
    gpu.sort_device_tensor(keys.view(torch.uint32))
    assert int(keys.sum()) == int(src.sum())
    assert _is_sorted(torch, keys ^ torch.iinfo(torch.int32).min)


def test_skewed_full_size_inputs(gpu):
    """The reference's bimodal bench input (gen_inputs with shift 16, src/test_utils.rs:51-61 / benches/full_sort.rs:68-78) at
    5·10^8 keys: half the keys share the 16-bit prefix 0.  The 8 192-key sample sees it: K1h counts the slice exactly (a bucket
    of any size) before the MSD passes, which then run in their exact form for the hybrid route (mode 7: K3's passes); the
    hybrid route's giant kernels sort that bucket and the buckets of one value the other half makes are written at once.  An input with ONE bucket one key over the tile is invisible
    to the sample and is caught by the atomic route's own exact check (a slot claim that does not fit): the hybrid route takes it."""
    import torch
    n = 500_000_000
    src = _gen(torch, n, torch.int32, 0x5D570008)
    bimodal = torch.cat([(src[: n // 2] >> 16) & 0xFFFF, src[n // 2:] << 16])
    borderline = src.clone()
    borderline[:16_897] = (borderline[:16_897] & 0xFFFF) | (0x1234 << 16)   # >= 16 897 keys with prefix 0x1234 (+ ~7 600 random ones)
    for mode, first in ((True, "msd_pass_a"), (7, "histogram16")):
        gpu.set_hybrid(mode)
        try:
            for name, inp, first_runs, route in (("bimodal", bimodal, True, "hybrid"), ("borderline", borderline, True, "hybrid")):
                keys = inp.clone()
                gpu.sort_device_tensor(keys.view(torch.uint32))   # first use of a kernel loads its code object: not timed
                keys.copy_(inp)
                gpu.set_profiling(True)
                gpu.sort_device_tensor(keys.view(torch.uint32))
                prof = gpu.profile_run(-1, 4)
                gpu.set_profiling(False)
                assert gpu.last_route() == route, (mode, name)
                assert (prof[first] > 0.2) == first_runs, (mode, name, prof[first])   # >= 0.4 ms when it reads the slice, microseconds when it returns
                assert int(keys.sum()) == int(inp.sum())
                assert _is_sorted(torch, keys ^ torch.iinfo(torch.int32).min), (mode, name)
        finally:
            gpu.set_hybrid(True)


def test_more_giants_than_tables_take_the_lsd_route(gpu):
    """every eighth 16-bit prefix holds ~73 000 keys: 8 192 giants, twice the count tables the hybrid route's giant kernels
    have (pass B of the atomic route overflows its slots first; the top bytes alone look uniform)."""
    import torch
    n = 600_000_000
    i = torch.arange(n, dtype=torch.int64, device="cuda")
    keys = ((((i * 2654435761) & 0x1FFF) * 8 + 3) << 16) | ((i * 40503) & 0xFFFF)
    keys = keys.to(torch.int32)          # (wraps: the bits are what counts)
    del i
    src_sum, src_x = int(keys.sum()), int((keys ^ (keys >> 11)).sum())
    gpu.sort_device_tensor(keys.view(torch.uint32))
    assert gpu.last_route() == "lsd"
    assert int(keys.sum()) == src_sum and int((keys ^ (keys >> 11)).sum()) == src_x
    assert _is_sorted(torch, _mapped(torch, keys, "u"))


def test_keys_that_share_their_top_bits_take_the_atomic_route_through_a_lowered_window(gpu):
    """Keys below 2^30 (ids), one rank's share of an 8-way sharded sort (top byte in [32, 64)), positive i32, f32 in [0.5, 2):
    the sample's keys share their top 2 / 3 / 1 / 8 bits, the atomic route's 65 536 buckets are taken that many bits lower
    and the slice sorts as fast as uniform keys do.  One stray key the sample does not see is caught by pass A's check of
    every key (the route is given up, the next one sorts)."""
    import torch
    n = 400_000_000
    g = torch.Generator(device="cuda").manual_seed(0x5D57000A)
    r = torch.randint(-(2**31), 2**31, (n,), dtype=torch.int32, device="cuda", generator=g)
    cases = [("uint32", r & 0x3FFFFFFF, "atomic"),
             ("uint32", (r & 0x1FFFFFFF) | 0x20000000, "atomic"),
             ("int32", r & 0x7FFFFFFF, "atomic"),
             ("float32", (r & 0x00FFFFFF) | 0x3F000000, "atomic")]   # f32 in [0.5, 2): sign and seven exponent bits shared
    stray = (r & 0x3FFFFFFF).clone()
    stray[123_456_789] = -1                      # 0xFFFFFFFF (no sampled position: they are multiples of n / 8192)
    cases.append(("uint32", stray, None))
    g8 = torch.Generator(device="cuda").manual_seed(0x5D57000B)
    r8 = torch.randint(-(2**63), 2**63 - 1, (300_000_000,), dtype=torch.int64, device="cuda", generator=g8)
    cases.append(("uint64", (r8 & 0x1FFFFFFFFFFFFFFF) | 0x2000000000000000, "atomic"))   # one rank's share of 8, 8-byte keys
    cases.append(("int64", r8 & 0x7FFFFFFFFFFFFFFF, "atomic"))
    del r8
    for name, src, want in cases:
        keys = src.clone()
        s1, s2 = int(keys.sum()), int((keys ^ (keys >> 11)).sum())
        gpu.sort_device_tensor(keys.view(getattr(torch, name)))
        route = gpu.last_route()
        assert (route == want) if want else (route != "atomic"), (name, route)
        assert int(keys.sum()) == s1 and int((keys ^ (keys >> 11)).sum()) == s2, name
        assert _is_sorted(torch, _mapped(torch, keys, np.dtype(name).kind)), name
        del keys


def _sum_pair(torch, x, chunk=1 << 28):
    """two wrapping checksums in bounded temporaries"""
    a = b = 0
    for s in range(0, x.numel(), chunk):
        c = x[s:s + chunk]
        a = (a + int(c.sum(dtype=torch.int64))) & (2**64 - 1)
        b = (b + int((c ^ (c >> 11)).sum(dtype=torch.int64))) & (2**64 - 1)
    return a, b


def test_eight_billion_u64_keys_on_one_gpu(gpu):
    """BASELINE configs[4]'s single-GPU leg (the 8-GPU figure is divided by "1 GPU, 8·10^9 u64"): n ~ 2^33 —
    element indices and status-word prefixes past 2^32, 24 GB of 64-bit status rows, 64 GB + 64 GB of keys.  Same
    size-independent properties as test_one_billion_keys: per-level histograms unchanged, two checksums, sortedness.  The
    source is not kept (keys + tmp + workspace ~ 155 GB)."""
    import torch
    n = 8_000_000_000
    torch.cuda.empty_cache()          # (earlier tests' blocks sit in torch's caching allocator)
    gpu.release_workspace()           # ... and the 10^9-key routes' 7-19 GB workspace in the library
    free, _ = torch.cuda.mem_get_info()
    assert free > 150 * 2**30, f"needs ~150 GiB of free HBM (64 + 64 + 24 GB), {free / 2**30:.0f} available"
    keys = _gen(torch, n, torch.int64, 0x5D570005)
    view = keys.view(torch.uint64)
    before = gpu.all_level_counts(view)
    sums = _sum_pair(torch, keys)
    gpu.sort_device_tensor(view)
    gpu.device_status()
    after = gpu.all_level_counts(view)
    assert np.array_equal(before, after)
    assert int(before[0].sum()) == n
    assert _sum_pair(torch, keys) == sums
    mn = torch.iinfo(torch.int64).min
    for s in range(0, n, 1 << 27):   # (chunks: no n-sized temporaries)
        e = min(n, s + (1 << 27) + 1)
        c = keys[s:e] ^ mn
        assert bool((c[1:] >= c[:-1]).all()), f"inversion in [{s}, {e})"
    # the same through the numpy statement of the key order, on the slice's two ends and a stretch across index 2^32
    from helpers import mapped_key
    for lo in (0, (1 << 32) - 500_000, n - 1_000_000):
        k = mapped_key(keys[lo:lo + 1_000_000].cpu().numpy().view("uint64"))
        assert (k[1:] >= k[:-1]).all()


def test_more_than_2_pow_32_u32_keys(gpu):
    """4-byte keys past 2^32 ELEMENTS (17 GB + 17 GB): 64-bit destinations in every pass, tile starts past 2^32."""
    import torch
    n = (1 << 32) + 54_321
    torch.cuda.empty_cache()
    gpu.release_workspace()
    keys = _gen(torch, n, torch.int32, 0x5D57000C)
    view = keys.view(torch.uint32)
    before = gpu.all_level_counts(view)
    sums = _sum_pair(torch, keys)
    gpu.sort_device_tensor(view)
    gpu.device_status()
    assert np.array_equal(before, gpu.all_level_counts(view))
    assert _sum_pair(torch, keys) == sums
    assert _is_sorted(torch, _mapped(torch, keys, "u"))
    tail = keys[n - 100_000:].cpu().numpy().view("uint32")   # the last tile lies wholly past element 2^32
    assert (tail[1:] >= tail[:-1]).all() and int(tail[-1]) >= 0xFFFF0000
