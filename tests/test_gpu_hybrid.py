"""The hybrid route (K1h -> route -> two K3 passes on the top 16 bits -> K4 in-LDS sort per bucket) against
the same oracles as the LSD route: bit-exact, every key type it is built for, bucket shapes from empty to
one key short of a tile, the fallbacks (a bucket too large, a K1h counter overflow), skewed low digits
inside the buckets (heavy-digit ranking, trivial levels), and the self-test mode of the fast ranking.
The reference's own route at 10^9 keys has this shape (two MSD levels, then Lsb on ~15 k-key chunks:
src/tuners/standard_tuner.rs:46-62, src/sorts/lsb_sort.rs:39-127)."""
import numpy as np
import pytest

from helpers import DTYPES, mapped_key, random_bits, reference_sorted, same_bits, to_device, to_host

pytestmark = pytest.mark.gpu

TILE = {4: 16896, 8: 16384}  # K4 tile = largest bucket the hybrid route accepts


MODES = {"count": 7, "count_whole_keys": 3, "ranked": 2, "wide_one_block": 6, "atomic": True, "atomic_4_only": 8, "no_expand": 9,
         "atomic_then_lsd": 10, "no_giants": 11}


@pytest.fixture(params=list(MODES))
def hybrid(gpu, request):
    """the routes and the forms of K4.  "count": the K1h hybrid route for every width — 4-byte keys: counting sort by value
    fed with the 16-bit halves pass L-1 leaves in the workspace, refused buckets (a value 16 times, more than a tile) to the
    expanding kernel; 8-byte keys: two 512-thread blocks per CU, or ("wide_one_block") one of 1024.  "count_whole_keys": the
    same fed with whole keys.  "ranked": the generic ranked passes for every bucket.  "no_expand": without the expanding
    kernel (buckets up to one tile only, refused buckets to the ranked kernel).  "atomic": the library's default — 4- and
    8-byte keys try the atomic route (MSD passes that claim space, no counting read), then the hybrid one, then LSD
    ("atomic_4_only": 8-byte keys start at the hybrid route; "atomic_then_lsd": no hybrid route behind a failed atomic one;
    "no_giants": without the giant kernels — a 4-byte bucket of 65 536 keys and more sends the sort down the LSD route)."""
    gpu.set_hybrid(MODES[request.param], min_len=1)   # consider the routes at every length (default: from 2^26 u64 / 3 * 2^26 u32 keys up)
    gpu._test_mode = request.param
    yield gpu
    gpu.set_hybrid(True, 0)
    gpu.set_tuning()


def _takes_atomic(rdst, dtype):
    mode, nb = getattr(rdst, "_test_mode", ""), np.dtype(dtype).itemsize
    return (mode in ("atomic", "atomic_then_lsd", "no_giants") and nb in (4, 8)) or (mode == "atomic_4_only" and nb == 4)


def _giants_ok(rdst, dtype):
    """4-byte keys, the counting and expanding K4 kernels on: buckets of 65 536 keys and more are the giant kernels'"""
    return np.dtype(dtype).itemsize == 4 and getattr(rdst, "_test_mode", "") in ("count", "count_whole_keys", "wide_one_block", "atomic", "atomic_4_only")


def _bucket_cap(rdst, dtype):
    """the largest bucket the hybrid route takes in this mode"""
    mode, nb = getattr(rdst, "_test_mode", ""), np.dtype(dtype).itemsize
    return 65535 if nb == 4 and mode not in ("ranked", "no_expand") else TILE[nb]


def _skewed_route(rdst, dtype, fits):
    """the route of an input the atomic route's areas cannot hold: the hybrid route if every bucket fits its bound"""
    if _takes_atomic(rdst, dtype) and getattr(rdst, "_test_mode", "") == "atomic_then_lsd":
        return "lsd"
    return "hybrid" if fits else "lsd"


def _fast_route(rdst, route, dtype, strict=True):
    """the route a sort whose buckets all fit a tile must have taken: "hybrid" — except in the "atomic" modes of the fixture,
    where the atomic route comes first; an input whose top bytes are far from uniform overflows that route's areas and falls
    to the hybrid route (or, "atomic_then_lsd", the LSD one) (strict=False: these are fine too)"""
    if _takes_atomic(rdst, dtype):
        after = "lsd" if getattr(rdst, "_test_mode", "") == "atomic_then_lsd" else "hybrid"
        return route == "atomic" or (not strict and route == after)
    return route == "hybrid"


def _sort(rdst, a):
    t = to_device(a)
    rdst.radix_sort_unstable(t)
    route = rdst.last_route()
    return to_host(t, a.dtype), route


def _with_prefixes(n, dtype, prefixes, seed, low_mask=None):
    """random keys whose top 16 bits (of the raw pattern) come from `prefixes`; low bits random & low_mask"""
    rng = np.random.default_rng(seed)
    dt = np.dtype(dtype)
    w = dt.itemsize * 8
    u = random_bits(n, f"uint{w}", seed).copy()
    if low_mask is not None:
        u &= np.array(low_mask, dtype=u.dtype)
    low = u & np.array((1 << (w - 16)) - 1, dtype=u.dtype)
    top = rng.choice(np.asarray(prefixes, dtype=np.uint64), size=n).astype(u.dtype)
    return (low | (top << np.array(w - 16, dtype=u.dtype))).view(dt)


@pytest.mark.parametrize("dtype", DTYPES)
def test_hybrid_matches_oracle_over_sizes(hybrid, oracle, dtype):
    for i, n in enumerate((16_385, 70_000, 300_001, 1_000_003, 5_000_011)):
        a = random_bits(n, dtype, seed=900 + i).copy()
        exp = a.copy()
        oracle.sort(exp, threads=4)
        got, route = _sort(hybrid, a)
        assert _fast_route(hybrid, route, dtype), (dtype, n, route)
        assert same_bits(got, exp), (dtype, n)


@pytest.mark.parametrize("dtype", DTYPES)
def test_buckets_up_to_one_tile(hybrid, dtype):
    nb = np.dtype(dtype).itemsize
    tile = TILE[nb]
    rng = np.random.default_rng(17)
    # ~300 buckets of ~13 k keys each: most of a tile, several 64-key rounds short of it
    prefixes = rng.choice(65536, size=300, replace=False)
    a = _with_prefixes(300 * (tile - 3000), dtype, prefixes, seed=18)
    got, route = _sort(hybrid, a)
    assert _fast_route(hybrid, route, dtype, strict=False)
    assert same_bits(got, reference_sorted(a)), dtype
    # exact fits: buckets of tile, tile - 1, 1, 2, 63, 64, 65 keys and empty ones in between
    sizes = [tile, tile - 1, 1, 2, 63, 64, 65, 768, 769, tile - 767, 5]
    w = nb * 8
    parts = []
    for j, sz in enumerate(sizes):
        low = random_bits(sz, f"uint{w}", seed=40 + j) & np.array((1 << (w - 16)) - 1, dtype=f"uint{w}")
        parts.append(low | np.array((3 * j + 1) << (w - 16), dtype=f"uint{w}"))
    b = np.concatenate(parts)
    rng.shuffle(b)
    b = np.concatenate([b, random_bits(20_000, f"uint{w}", seed=77)]).view(dtype)  # plus thin buckets everywhere
    # the exact-fit buckets may have grown by a random key: that is the fallback's business (next test), so only
    # demand a correct result here and check the route only when every bucket still fits
    got, route = _sort(hybrid, b)
    top = (mapped_key(b) >> np.array(w - 16, dtype=f"uint{w}")).astype(np.int64)
    fits = np.bincount(top, minlength=65536).max() <= _bucket_cap(hybrid, dtype) or _giants_ok(hybrid, dtype)
    assert (route == "lsd") if not fits else _fast_route(hybrid, route, dtype, strict=False)
    assert same_bits(got, reference_sorted(b)), dtype


def test_bucket_larger_than_the_routes_bound(hybrid):
    """8-byte keys, and 4-byte keys without the expanding K4: a bucket over one tile sends the sort down the LSD route;
    4-byte keys by default: the hybrid route takes buckets of up to 65 535 keys (the expanding kernel sorts what is
    over a tile), and one key more is the LSD route's."""
    for dtype in ("uint32", "float64"):
        tile = TILE[np.dtype(dtype).itemsize]
        a = _with_prefixes(40 * (tile + 2000), dtype, list(range(1000, 1040)), seed=5)
        got, route = _sort(hybrid, a)
        assert route == _skewed_route(hybrid, dtype, _bucket_cap(hybrid, dtype) > tile), (dtype, route)
        assert same_bits(got, reference_sorted(a))
    for extra in (0, 1):   # exactly the bound / one key too many in one bucket
        cap = _bucket_cap(hybrid, "uint32")
        low = random_bits(cap + extra, "uint32", seed=6) & np.uint32(0xFFFF)
        a = np.concatenate([low | np.uint32(0xABCD0000), random_bits(100_000, "uint32", seed=7) & np.uint32(0x7FFFFFFF)])
        got, route = _sort(hybrid, a)
        assert route == _skewed_route(hybrid, "uint32", not extra or _giants_ok(hybrid, "uint32")), (extra, route)
        assert same_bits(got, reference_sorted(a))


def test_buckets_of_several_tiles_and_of_few_distinct_values(hybrid):
    """what the expanding K4 is for (4-byte keys): buckets between one tile and 65 535 keys, and buckets whose low halves
    repeat (a value 16 times overflows the counting kernel's 4-bit counters) — including buckets of ONE value, whose 64
    equal keys per wave are added once."""
    rng = np.random.default_rng(101)
    for dtype in ("uint32", "int32", "float32"):
        parts = []
        for j, size in enumerate((20_000, 40_000, 65_535, 17_000, 16_897, 3, 700)):
            low = rng.integers(0, 1 << 16, size=size, dtype=np.uint32)
            if j % 3 == 1:
                low &= np.uint32(0x00FF)          # 256 distinct values
            if j % 3 == 2:
                low[:] = 0x4242                   # one value
            parts.append(low | np.uint32((0x0100 + 37 * j) << 16))
        parts.append(rng.integers(0, 1 << 16, size=30_000, dtype=np.uint32) & np.uint32(0xF00F) | np.uint32(0x7001 << 16))
        parts.append(random_bits(200_000, "uint32", seed=102))
        a = np.concatenate(parts)
        rng.shuffle(a)
        a = a.view(dtype)
        got, route = _sort(hybrid, a)
        top = (mapped_key(a) >> np.uint32(16)).astype(np.int64)
        fits = np.bincount(top, minlength=65536).max() <= _bucket_cap(hybrid, dtype) or _giants_ok(hybrid, dtype)
        assert route == _skewed_route(hybrid, dtype, fits), (dtype, route)   # (too skewed for the atomic route's areas in any mode)
        assert same_bits(got, reference_sorted(a)), dtype


def test_counter_overflow_in_k1h_is_detected(hybrid):
    """> 65 535 keys of one bucket inside one block's piece: without the giant kernels the packed 16-bit LDS counter carries
    or wraps and the block-wide sum test must send the sort down the LSD route; with them (4-byte keys by default) a counter
    is 15 bits and a guard, every wrap is moved to the global table as it happens, and the hybrid route sorts the bucket."""
    n = 6_000_000
    want = _skewed_route(hybrid, "uint32", _giants_ok(hybrid, "uint32"))
    for heavy_prefix in (0x1234, 0x1235):   # low and high half of a counter word
        a = random_bits(n, "uint32", seed=8).copy()
        a[: n // 2] = (a[: n // 2] & np.uint32(0xFFFF)) | np.uint32(heavy_prefix << 16)
        np.random.default_rng(9).shuffle(a)
        got, route = _sort(hybrid, a)
        assert route == want
        assert same_bits(got, reference_sorted(a))
    # 65 536 * k keys in one bucket of one piece: a counter that wraps to exactly 0
    a = np.full(65536 * 4, 0x77770000, dtype=np.uint32) | (random_bits(65536 * 4, "uint32", seed=10) & np.uint32(0xFFFF))
    a = np.concatenate([a, random_bits(50_000, "uint32", seed=11)])
    got, route = _sort(hybrid, a)
    assert route == want
    assert same_bits(got, reference_sorted(a))


def test_giant_buckets(hybrid):
    """4-byte keys: 16-bit prefixes holding 65 536 keys and more — dense and sparse in their low halves, of two values, of one
    value, neighbours in one counter word, first and last prefix, mapped key kinds, the reference's bimodal bench shape
    (gen_inputs with shift 16, src/test_utils.rs:51-61) — are sorted by the hybrid route's giant kernels."""
    rng = np.random.default_rng(2024)

    def bucket(prefix, size, kind):
        if kind == "dense":
            low = rng.integers(0, 1 << 16, size=size, dtype=np.uint32)
        elif kind == "sparse":      # few distinct values far apart: long empty stretches of the count table
            low = rng.choice(np.array([0, 1, 300, 30000, 32767, 32768, 65000, 65535], dtype=np.uint32), size=size)
        elif kind == "two":
            low = rng.choice(np.array([0, 65535], dtype=np.uint32), size=size)
        elif kind == "one":
            low = np.full(size, 0x8000, dtype=np.uint32)
        else:                        # "narrow": every value of a small range, many times
            low = rng.integers(1000, 1100, size=size, dtype=np.uint32)
        return low | np.uint32(prefix << 16)

    for dtype in ("uint32", "int32", "float32"):
        parts = [bucket(0x0000, 65_536, "dense"), bucket(0xFFFF, 70_001, "sparse"), bucket(0x1234, 300_000, "dense"),
                 bucket(0x1235, 65_537, "two"), bucket(0x8000, 131_072, "one"), bucket(0x7FFF, 1_100_000, "narrow"),
                 bucket(0x4000, 65_535, "dense"), bucket(0x4001, 20_000, "two"), random_bits(500_000, "uint32", seed=5)]
        a = np.concatenate(parts)
        rng.shuffle(a)
        a = a.view(dtype)
        got, route = _sort(hybrid, a)
        assert route == _skewed_route(hybrid, dtype, _giants_ok(hybrid, dtype)), (dtype, route)
        assert same_bits(got, reference_sorted(a)), dtype
    # bimodal: half the keys below 2^16 (one giant bucket), the other half with their low 16 bits zero (buckets of one value)
    r = random_bits(4_000_000, "uint32", seed=6)
    a = np.concatenate([r[:2_000_000] >> np.uint32(16), r[2_000_000:] << np.uint32(16)])
    got, route = _sort(hybrid, a)
    assert route == _skewed_route(hybrid, "uint32", _giants_ok(hybrid, "uint32"))
    assert same_bits(got, reference_sorted(a))
    # every key in one bucket; and in one bucket and one value but for a few
    for a in (random_bits(3_000_000, "uint32", seed=7) & np.uint32(0xFFFF) | np.uint32(0x00420000),
              np.where(np.arange(2_000_000) % 100_003 == 0, np.uint32(7), np.uint32(0x00420042)).astype(np.uint32)):
        got, route = _sort(hybrid, a)
        assert route == _skewed_route(hybrid, "uint32", _giants_ok(hybrid, "uint32"))
        assert same_bits(got, reference_sorted(a))


def test_pass_b_grid_larger_than_the_areas(gpu):
    """lengths at which pass B's grid (sized for the exact form too: tiles + 256) exceeds 2 048 areas x tiles per area: the
    blocks past the last area must do nothing (a randomized run caught them sorting what lies behind the claim counters)."""
    gpu.set_hybrid(True, 1)
    try:
        for n in (30_405_504, 65_048_690, 99_514_889):
            a = random_bits(n, "uint32", seed=n & 0xFFFF).copy()
            got, route = _sort(gpu, a)
            assert route == "atomic"
            assert same_bits(got, reference_sorted(a)), n
    finally:
        gpu.set_hybrid(True, 0)


def test_two_thousand_giants(hybrid):
    """2 048 prefixes of ~67 000 keys each (what 10^9 normally distributed f32 keys look like to the route): within the 4 096
    count tables; more giants than tables is test_gpu_fullsize.py's (it takes 6·10^8 keys)."""
    if hybrid._test_mode not in ("atomic", "count", "no_giants"):
        pytest.skip("137 M keys: three modes are enough")
    a = (np.arange(2100 * 65_536, dtype=np.uint32) * np.uint32(2654435761)) & np.uint32(0x07FFFFFF)
    top = np.bincount((a >> np.uint32(16)).astype(np.int64), minlength=65536)
    assert (top >= 65536).sum() == 2048
    got, route = _sort(hybrid, a)
    # (137 M keys: from 2^26 keys up there is a sample, it sees the five shared top bits and lowers the window past them — the
    # modes that try the atomic route first need no giants; "count" starts at the K1h hybrid route and has its 2 048)
    assert route == ("atomic" if _takes_atomic(hybrid, "uint32") else "hybrid" if _giants_ok(hybrid, "uint32") else "lsd")
    assert same_bits(got, reference_sorted(a))


@pytest.mark.parametrize("dtype", ("uint32", "int32", "float32", "uint64", "int64"))
def test_skewed_low_digits_inside_buckets(hybrid, dtype):
    nb = np.dtype(dtype).itemsize
    w = nb * 8
    prefixes = list(range(7, 7 + 200))
    n = 200 * 9000
    full = (1 << (w - 16)) - 1
    masks = [0, 0xFF, 0xFF00, 0x1, 0x8000, 0xF0F0, full & ~0xFF, full & ~0xFFFF if w > 32 else 0x00FF, 0x0101]
    for j, m in enumerate(masks):
        a = _with_prefixes(n, dtype, prefixes, seed=300 + j, low_mask=(m | (0xFFFF << (w - 16))))
        got, route = _sort(hybrid, a)
        assert _fast_route(hybrid, route, dtype, strict=False), (dtype, hex(m), route)
        assert same_bits(got, reference_sorted(a)), (dtype, hex(m))
    # every bucket already sorted / reverse sorted inside, buckets interleaved
    a = _with_prefixes(n, dtype, prefixes, seed=400)
    s = reference_sorted(a)
    for arr in (s, s[::-1].copy()):
        arr = arr.copy()
        perm = np.random.default_rng(3).permutation(200)
        arr = arr.reshape(200, -1)[perm].reshape(-1)  # blocks of 9000 in shuffled order: still one inversion at least
        got, route = _sort(hybrid, arr)
        assert same_bits(got, reference_sorted(arr)), dtype


def test_float_specials_through_the_hybrid_route(hybrid):
    rng = np.random.default_rng(12)
    for dtype, ut in (("float32", np.uint32), ("float64", np.uint64)):
        a = random_bits(400_000, dtype, seed=13).copy()
        sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -np.nan, 1.0, -1.0, np.finfo(dtype).tiny, -np.finfo(dtype).tiny], dtype=dtype)
        a[rng.integers(0, a.size, size=5000)] = sp[rng.integers(0, sp.size, size=5000)]
        got, route = _sort(hybrid, a)
        assert _fast_route(hybrid, route, dtype, strict=False)   # (a few thousand specials on a handful of top bytes can overflow an area)
        assert same_bits(got, reference_sorted(a))
        k = mapped_key(got)
        assert (k[1:] >= k[:-1]).all()


def test_fast_rank_selftest_and_ballot_modes_agree(hybrid):
    a = _with_prefixes(150 * 12000, "uint32", list(range(500, 650)), seed=21)
    b = random_bits(1_500_000, "uint64", seed=22).copy()
    exp_a, exp_b = reference_sorted(a), reference_sorted(b)
    for mode in (True, 2, False):          # returning-add ranking, its fallback forced on every round, ballots only
        hybrid.set_tuning(fast_rank=mode)
        got, route = _sort(hybrid, a)
        assert route in ("hybrid", "atomic", "lsd") and same_bits(got, exp_a), mode
        got, route = _sort(hybrid, b)
        assert _fast_route(hybrid, route, "uint64") and same_bits(got, exp_b), mode
    hybrid.set_tuning()


def test_already_sorted_input_runs_no_pass_on_either_route(hybrid):
    a = np.sort(random_bits(2_000_000, "uint32", seed=31))
    got, _ = _sort(hybrid, a)
    assert same_bits(got, a)
    prof_n = 2_000_000
    hybrid.set_profiling(True)
    t = to_device(a)
    hybrid.sort_device_tensor(t)
    p = hybrid.profile_run(-1, 4)
    hybrid.set_profiling(False)
    assert p is not None and ("local_sort" in p or "msd_pass_a" in p) and prof_n == a.size
    assert same_bits(to_host(t, a.dtype), a)


def test_k1h_sees_every_adjacent_pair(hybrid):
    """The already-sorted exit now also rests on K1h's inversion test (a sorted slice makes K1 return at once): one
    swapped pair anywhere — inside a lane's vector, between lanes, between waves, between blocks' pieces, at the ends —
    must be seen, or the slice would come back unsorted."""
    n = 3_000_000
    base = np.sort(random_bits(n, "uint32", seed=77))
    base = np.unique(base)          # strictly increasing: every swap is an inversion
    n = base.size
    piece = -(-n // 256)            # K1h's pieces are multiples of 16 384 keys; probe around plausible boundaries anyway
    spots = [0, 1, 2, 3, 4, 63, 64, 255, 256, 4095, 4096, 16383, 16384, 16385, piece - 1, piece, 2 * piece, n // 2, n - 3, n - 2]
    for i in spots:
        a = base.copy()
        a[i], a[i + 1] = a[i + 1], a[i]
        got, _ = _sort(hybrid, a)
        assert same_bits(got, base), i
    got, _ = _sort(hybrid, base.copy())
    assert same_bits(got, base)


def test_device_error_word_is_sticky_until_checked(gpu):
    """An error raised by an earlier asynchronous sort must still be reported after later sorts were
    enqueued: the word lives outside the per-sort workspace; the check reports it once."""
    import ctypes
    import torch
    from rdst_amd import _lib
    lib = _lib.load()
    a = to_device(random_bits(3_000_000, "uint32", seed=1).copy())
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.rdst_hip_debug_raise_device_error(2, s))      # "sort A failed"
    gpu.sort_device_tensor(a, check=False)                        # sort B, clean, same stream
    big = to_device(random_bits(40_000_000, "uint32", seed=2).copy())
    gpu.sort_device_tensor(big, check=False)                      # grows the workspace
    with pytest.raises(_lib.RdstHipError):
        gpu.device_status()
    gpu.device_status()                                           # reported once, then clean
    # (results enqueued between a failure and its report are suspect by contract: look-back walkers give up
    # early while the word is raised) — after the report the device sorts normally again
    c = to_device(random_bits(3_000_000, "uint32", seed=3).copy())
    gpu.sort_device_tensor(c)
    k = c.view(torch.int32) ^ (-(2**31))
    assert bool((k[1:] >= k[:-1]).all())
