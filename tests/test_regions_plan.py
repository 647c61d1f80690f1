"""The swap plan of the low-memory route's Regions level (rdst_regions_plan, host only): applied to an array whose
tiles are grouped by digit, the rounds of swaps must leave every digit in its region [start[d], start[d+1]), the
ranges of one round must be pairwise disjoint (one kernel launch per round), and the key multiset is untouched.
Reference: src/sorts/regions_sort.rs:51-286 (outbound edges :66-123, swap operations :126-204, rounds :229-261)."""
import ctypes

import numpy as np
import pytest


class SwapOp(ctypes.Structure):
    _fields_ = [("a", ctypes.c_uint64), ("b", ctypes.c_uint64), ("len", ctypes.c_uint64)]


ARGTYPES = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32,
            ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p, ctypes.c_uint32,
            ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p]


def plan(lib, tile_counts, tile_len, n, buckets, col_bucket=None):
    tiles = tile_counts.shape[0]
    cap = 4 * tiles * tile_counts.shape[1] + 4 * buckets * buckets + 16
    ops = (SwapOp * cap)()
    nops, nrounds = ctypes.c_uint64(0), ctypes.c_uint32(0)
    max_rounds = 4096
    rounds = (ctypes.c_uint64 * (max_rounds + 1))()
    starts = (ctypes.c_uint64 * (buckets + 1))()
    tc = np.ascontiguousarray(tile_counts, dtype=np.uint64)
    lib.rdst_regions_plan.argtypes = ARGTYPES
    cb = None
    if col_bucket is not None:
        cb = (ctypes.c_uint32 * len(col_bucket))(*col_bucket)
    rc = lib.rdst_regions_plan(tc.ctypes.data, tiles, tile_len, n, tile_counts.shape[1], cb, buckets, ops, cap, ctypes.byref(nops), rounds,
                               max_rounds, ctypes.byref(nrounds), starts)
    assert rc == 0
    return [(ops[i].a, ops[i].b, ops[i].len) for i in range(nops.value)], list(rounds)[:nrounds.value + 1], list(starts)


def _digits(n, buckets, shape, rng):
    if shape == "uniform":
        return rng.integers(0, buckets, size=n)
    if shape == "skew":       # one digit holds 70 %
        d = rng.integers(0, buckets, size=n)
        d[rng.random(n) < 0.7] = buckets // 3
        return d
    if shape == "two":        # two digits only: most countries are empty
        return rng.choice([1, buckets - 1], size=n)
    if shape == "one":        # nothing to do
        return np.full(n, buckets // 2)
    if shape == "sorted":     # already grouped
        return np.sort(rng.integers(0, buckets, size=n))
    if shape == "reversed":
        return np.sort(rng.integers(0, buckets, size=n))[::-1].copy()
    raise ValueError(shape)


@pytest.mark.parametrize("buckets", [2, 256])
@pytest.mark.parametrize("shape", ["uniform", "skew", "two", "one", "sorted", "reversed"])
def test_plan_groups_every_digit_into_its_region(hiplib, buckets, shape):
    rng = np.random.default_rng(hash((buckets, shape)) % 2**32)
    for n, tile_len in ((1, 5), (1000, 1000), (10_007, 1_000), (200_003, 4_096), (65_536, 1_024)):
        digit = _digits(n, buckets, shape, rng).astype(np.int64)
        ident = np.arange(n, dtype=np.int64)            # every element is unique: the multiset check is exact
        tiles = -(-n // tile_len)
        counts = np.zeros((tiles, buckets), dtype=np.uint64)
        arr_d, arr_i = digit.copy(), ident.copy()
        for t in range(tiles):                          # step (1): every tile grouped by digit (any order inside a run)
            lo, hi = t * tile_len, min(n, (t + 1) * tile_len)
            o = np.argsort(arr_d[lo:hi], kind="stable")
            arr_d[lo:hi], arr_i[lo:hi] = arr_d[lo:hi][o], arr_i[lo:hi][o]
            counts[t] = np.bincount(arr_d[lo:hi], minlength=buckets)
        ops, rounds, starts = plan(hiplib, counts, tile_len, n, buckets)
        assert starts[-1] == n and np.array_equal(np.diff(starts), np.bincount(digit, minlength=buckets))
        for r in range(len(rounds) - 1):
            touched = np.zeros(n, dtype=np.int8)
            for a, b, m in ops[rounds[r]:rounds[r + 1]]:
                assert m > 0 and a + m <= n and b + m <= n
                touched[a:a + m] += 1
                touched[b:b + m] += 1
                for arr in (arr_d, arr_i):
                    tmp = arr[a:a + m].copy()
                    arr[a:a + m] = arr[b:b + m]
                    arr[b:b + m] = tmp
            assert touched.max() <= 1, "ranges of one round overlap"
        for d in range(buckets):
            assert (arr_d[starts[d]:starts[d + 1]] == d).all(), (shape, n, d)
        assert np.array_equal(np.sort(arr_i), ident)
        assert np.array_equal(digit[arr_i], arr_d)      # every element still carries its own digit
        assert len(rounds) - 1 <= 64, len(rounds)       # rounds stay few (uniform: ~10)


def test_plan_rejects_inconsistent_counts(hiplib):
    counts = np.zeros((2, 256), dtype=np.uint64)
    counts[0, 3] = 10
    counts[1, 4] = 9          # second tile should hold 10 (n = 20, tile_len = 10)
    cap = 64
    ops = (SwapOp * cap)()
    nops, nrounds = ctypes.c_uint64(0), ctypes.c_uint32(0)
    rounds = (ctypes.c_uint64 * 9)()
    hiplib.rdst_regions_plan.argtypes = ARGTYPES
    assert hiplib.rdst_regions_plan(counts.ctypes.data, 2, 10, 20, 256, None, 256, ops, cap, ctypes.byref(nops), rounds, 8, ctypes.byref(nrounds), None) == -1


def test_plan_with_three_columns_for_two_buckets_is_a_partition(hiplib):
    """The partition's shape (rdst_hip_partition_device; src/sort_utils.rs:295-331): every tile lies as [digits below D][D][digits
    above D], bucket 0 = D, bucket 1 = everything else."""
    rng = np.random.default_rng(11)
    n, tile_len, D = 100_003, 2_048, 77
    digit = rng.integers(0, 256, size=n)
    digit[rng.random(n) < 0.3] = D
    tiles = -(-n // tile_len)
    arr = digit.copy()
    counts = np.zeros((tiles, 3), dtype=np.uint64)
    for t in range(tiles):
        lo, hi = t * tile_len, min(n, (t + 1) * tile_len)
        arr[lo:hi] = np.sort(arr[lo:hi])
        counts[t] = [(arr[lo:hi] < D).sum(), (arr[lo:hi] == D).sum(), (arr[lo:hi] > D).sum()]
    ops, rounds, starts = plan(hiplib, counts, tile_len, n, 2, col_bucket=[1, 0, 1])
    for a, b, m in ops:
        tmp = arr[a:a + m].copy()
        arr[a:a + m] = arr[b:b + m]
        arr[b:b + m] = tmp
    split = int((digit == D).sum())
    assert starts == [0, split, n]
    assert (arr[:split] == D).all() and (arr[split:] != D).all()
    assert np.array_equal(np.sort(arr), np.sort(digit))
