"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs (bit-exact: integer / byte work), against the committed golden vectors,
and the edge cases the reference tests (empty, one element, tile boundaries, all-equal,
sorted, reverse, masked bit patterns, bimodal shifts, float specials)."""
import json
import os

import numpy as np
import pytest

from helpers import (DTYPES, SMALL_DTYPES, gen_inputs, mapped_key, random_bits, reference_sorted, same_bits, to_device, to_host,
                     u32_patterns)

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _device_sort(rdst, a):
    t = to_device(a)
    rdst.radix_sort_unstable(t)
    return to_host(t, a.dtype)


def test_native_library_is_the_one_in_tree(gpu):
    from rdst_amd import _lib
    assert os.path.dirname(_lib.LIB_PATH) == os.path.join(os.path.dirname(HERE), "rdst_amd")
    assert _lib.load().rdst_hip_abi_version() == 2
    maps = open("/proc/self/maps").read()
    assert "librdst_hip.so" in maps


def test_golden_known_answers_on_device(gpu):
    ka = json.load(open(os.path.join(HERE, "golden", "reference_known_answers.json")))
    for case in ka["sorts"]:
        if case["type"] == "uint32":
            a = np.array(case["input"], dtype=np.uint32)
            assert _device_sort(gpu, a).tolist() == case["expected"], case["source"]
        elif case["type"] == "b4":  # [u8;4] lexicographic == big-endian u32
            a = np.array(case["input"], dtype=np.uint8).view(">u4").astype(np.uint32).ravel()
            got = _device_sort(gpu, a).astype(">u4").view(np.uint8).reshape(-1, 4)
            assert got.tolist() == case["expected"], case["source"]


def test_host_entry_point_sorts_numpy_in_place(gpu, oracle):
    for dtype in DTYPES:
        a = random_bits(300_001, dtype, seed=5).copy()
        exp = a.copy()
        oracle.sort(exp, threads=4)
        gpu.radix_sort_unstable(a)          # rdst_hip_sort: H2D, sort, D2H
        assert same_bits(a, exp), dtype


@pytest.mark.parametrize("dtype", DTYPES)
def test_sort_matches_oracle_over_sizes(gpu, oracle, dtype):
    # default tile of the key type: 4-byte keys 16 896, u64 10 752 (two-stage shape), i64 / f64 7 680 (whole-tile shape 5)
    tile = 16896 if np.dtype(dtype).itemsize == 4 else (10752 if dtype == "uint64" else 7680)
    sizes = [0, 1, 2, 3, 10, 100, 127, 128, 129, 5_000, tile - 1, tile, tile + 1, 2 * tile - 1, 2 * tile + 1, 18432, 50_000,
             100_000, 8 * tile, 8 * tile + 31, 300_000, 1_000_003, 5_000_011]
    for i, n in enumerate(sizes):
        a = random_bits(n, dtype, seed=100 + i).copy()
        exp = a.copy()
        oracle.sort(exp, threads=4)
        got = _device_sort(gpu, a)
        assert same_bits(got, exp), (dtype, n)
        assert same_bits(exp, reference_sorted(a))  # oracle vs independent numpy, same input


@pytest.mark.parametrize("dtype", DTYPES + SMALL_DTYPES)
def test_one_workgroup_sort_boundaries(gpu, dtype):
    """Slices of at most 64 KiB are sorted by one workgroup in LDS (no workspace, no look-back): every round
    count of that kernel, its size limit and the first size beyond it, random and degenerate inputs, and the
    same inputs through the general pipeline."""
    limit = 1024 * {1: 16, 2: 16, 4: 16, 8: 8}[np.dtype(dtype).itemsize]
    sizes = [2, 63, 64, 65, 1023, 1024, 1025, 2048, 3000, limit // 2 + 1, limit - 1, limit, limit + 1]
    for i, n in enumerate(sizes):
        base = random_bits(n, dtype, seed=500 + i).copy()
        for variant in ("random", "equal", "sorted", "reverse", "two values"):
            a = base.copy()
            if variant == "equal":
                a[:] = a[0]
            elif variant == "sorted":
                a = reference_sorted(a)
            elif variant == "reverse":
                a = reference_sorted(a)[::-1].copy()
            elif variant == "two values":
                a[:] = np.where(np.arange(n) % 3 == 0, a[0], a[-1])
            exp = reference_sorted(a)
            assert same_bits(_device_sort(gpu, a), exp), (dtype, n, variant)
            if variant == "random":
                try:
                    gpu.set_tuning(small_sort=False)
                    assert same_bits(_device_sort(gpu, a), exp), (dtype, n, "general pipeline")
                finally:
                    gpu.set_tuning()


@pytest.mark.parametrize("dtype", ["uint32", "float32", "uint64", "float64", "uint16"])
def test_every_built_pass_shape(gpu, dtype):
    """The scatter-kernel shapes the product library carries (two-stage 18 432 / 21 504-key tiles, whole-tile
    16 896 and 15 360) each sort every key width, not only the one they are the default for."""
    a = random_bits(1_500_007, dtype, seed=77).copy()
    exp = reference_sorted(a)
    try:
        for cfg in (2, 3, 4, 5):
            gpu.set_tuning(pass_config=cfg)
            assert same_bits(_device_sort(gpu, a), exp), (dtype, cfg)
    finally:
        gpu.set_tuning()


@pytest.mark.parametrize("dtype", ["uint32", "uint64", "float32"])
def test_chain_split_shapes(gpu, dtype):
    """Every pass splits its source into 8 segments with a look-back chain each (position ranges for
    the first pass, groups of the previous digit afterwards, one chain after a skipped level).
    Inputs that bend that: a skipped middle level, previous digits crowded into one group or
    missing from most, segments shorter than a tile — each sorted with the split on and off and
    against numpy on the mapped keys."""
    u = np.dtype({"float32": "uint32"}.get(dtype, dtype))
    bits = u.itemsize * 8
    rng = np.random.default_rng(77)
    n = 2_000_003
    full = rng.integers(0, 1 << 63, size=n, dtype=np.uint64).astype(u) if bits == 32 else rng.integers(0, 1 << 64, size=n, dtype=np.uint64)
    cases = {
        "level 1 constant": (full & ~u.type(0xFF00)) | u.type(0x4200),
        "level 0 in one group": (full & ~u.type(0xE0)),
        "level 0 90% in group 7": np.where(rng.random(n) < 0.9, full | u.type(0xE0), full),
        "level 1 only two digits": (full & ~u.type(0xFE00)),
        "levels 0-1 constant": (full & ~u.type(0xFFFF)) | u.type(0x1234),
        "top levels only": full & (u.type(0xFF) << u.type(bits - 8)),
        "uniform": full,
        "short": full[:40_000],
        "tiny": full[:700],
    }
    try:
        for name, a in cases.items():
            a = np.ascontiguousarray(a).view(dtype)
            exp = reference_sorted(a)
            for split in (True, False):
                gpu.set_tuning(chain_split=split)
                assert same_bits(_device_sort(gpu, a), exp), (dtype, name, split)
            for fast in (0, 2):  # ballots only; returning-add ranking with every round sent through its fallback
                gpu.set_tuning(fast_rank=fast)
                assert same_bits(_device_sort(gpu, a), exp), (dtype, name, "fast_rank", fast)
    finally:
        gpu.set_tuning()


@pytest.mark.parametrize("dtype", ["uint32", "int32", "float32", "uint64", "int64", "float64"])
@pytest.mark.parametrize("vdtype", ["int32", "int64"])
def test_key_value_pairs(gpu, dtype, vdtype):
    """rdst_hip_sort_pairs_device: keys sorted, values carried; stable, so the whole
    result is pinned by numpy's stable argsort of the mapped keys (rdst itself promises no order among
    equal keys: any such output is one it may produce)."""
    import torch
    pair = np.dtype(dtype).itemsize + np.dtype(vdtype).itemsize
    tile = 768 * {8: 11, 12: 7, 16: 5}[pair]
    rng = np.random.default_rng(4242)
    for i, n in enumerate((0, 1, 2, 129, tile - 1, tile, tile + 1, 8 * tile + 31, 100_003, 2_000_003)):
        a = random_bits(n, dtype, seed=700 + i).copy()
        if i % 2 == 1 and n > 4:  # few distinct keys: long runs of equal keys, where stability shows
            a = a[rng.integers(0, 7, size=n)].copy()
        vals = rng.integers(-(2**31), 2**31, size=n).astype(vdtype)
        order = np.argsort(mapped_key(a), kind="stable")
        tk, tv = to_device(a), torch.from_numpy(vals).cuda()
        gpu.sort_pairs_device_tensor(tk, tv)
        assert same_bits(to_host(tk, a.dtype), a[order]), (dtype, vdtype, n)
        assert np.array_equal(tv.cpu().numpy(), vals[order]), (dtype, vdtype, n)


@pytest.mark.parametrize("kdt,vdt", [("int32", "int32"), ("int32", "int64"), ("int64", "int32"), ("int64", "int64")])
def test_key_value_pairs_many_tiles(gpu, kdt, vdt):
    """10^7 pairs, every key / value width, several times: thousands of tiles in flight (this is where a missing
    wait state after the 16-byte status store once corrupted look-back words; 2 M pairs never showed it)."""
    import torch
    n = 10_000_019
    g = torch.Generator(device="cuda").manual_seed(11)
    info = torch.iinfo(getattr(torch, kdt))
    src = torch.randint(info.min, info.max, (n,), dtype=getattr(torch, kdt), device="cuda", generator=g)
    ut = torch.uint32 if kdt == "int32" else torch.uint64
    for _ in range(6):
        keys = src.clone()
        vals = torch.arange(n, dtype=getattr(torch, vdt), device="cuda")
        gpu.sort_pairs_device_tensor(keys.view(ut), vals)
        k = keys ^ info.min
        assert bool((k[1:] >= k[:-1]).all())
        assert bool((src[vals.long()] == keys).all())
        same = keys[1:] == keys[:-1]
        assert bool((vals[1:][same] > vals[:-1][same]).all())


def test_records_sorted_by_a_key_field(gpu):
    """A slice of structs with a built-in key field (benches/struct_sort.rs:11-27): rows follow their key."""
    import torch
    rng = np.random.default_rng(99)
    rec = rng.standard_normal((300_007, 5)).astype(np.float32)
    rec[::1000, 2] = np.float32(-0.0)
    rec[5::1000, 2] = np.float32("nan")
    out = gpu.sort_records_by_key(torch.from_numpy(rec).cuda(), 2).cpu().numpy()
    order = np.argsort(mapped_key(rec[:, 2].copy()), kind="stable")
    assert np.array_equal(out.view(np.uint32), rec[order].view(np.uint32))


def test_byte_array_keys(gpu, oracle):
    """[u8; N] (src/radix_key_impl.rs:78-85): level l reads byte N-1-l, rows end up in lexicographic order —
    the reference's own known answer (src/radix_sort.rs:221-229), the oracle's b3 / b4 types, numpy lexsort."""
    ka = json.load(open(os.path.join(HERE, "golden", "reference_known_answers.json")))
    for case in ka["sorts"]:
        if case["type"] in ("b3", "b4"):
            a = np.array(case["input"], dtype=np.uint8)
            gpu.radix_sort_unstable(a, key="bytes")
            assert a.tolist() == case["expected"], case["source"]
    rng = np.random.default_rng(808)
    for nb in range(1, 17):
        for n in (0, 1, 2, 257, 100_003):
            a = rng.integers(0, 256, size=(n, nb), dtype=np.uint8)
            a[rng.random((n, nb)) < 0.3] = 0  # ties on leading bytes
            exp = a[np.lexsort(a.T[::-1])] if n else a.copy()
            if nb in (3, 4) and n:
                o = a.copy()
                oracle.sort(o, threads=2, kind=f"b{nb}")
                assert np.array_equal(o, exp)
            got = a.copy()
            gpu.radix_sort_unstable(got, key="bytes")
            assert np.array_equal(got, exp), (nb, n)


@pytest.mark.parametrize("ktype", ["<f4", "<i8", "<u4", "<f8"])
def test_host_records_entry_point(gpu, ktype):
    """rdst_hip_sort_records: a host slice of structs ordered by one built-in field, rows moved whole."""
    rng = np.random.default_rng(31)
    for rec_dt in (np.dtype([("id", "<u4"), ("key", ktype), ("pad", "<u4")], align=True),
                   np.dtype([("w", "<f8", (3,)), ("key", ktype), ("tag", "<u8")], align=True)):
        for n in (0, 1, 2, 1000, 123_457):
            a = np.zeros(n, dtype=rec_dt)
            raw = rng.integers(0, 256, size=(n, rec_dt.itemsize), dtype=np.uint8)
            a.view(np.uint8).reshape(n, rec_dt.itemsize)[:] = raw
            if n > 10:
                a["key"][::3] = a["key"][0]  # repeated keys
            order = np.argsort(mapped_key(a["key"].copy()), kind="stable")
            exp = a.view(np.uint8).reshape(n, rec_dt.itemsize)[order].copy()  # whole rows, padding bytes included
            gpu.sort_host_records(a, "key")
            assert np.array_equal(a.view(np.uint8).reshape(n, rec_dt.itemsize), exp), (ktype, rec_dt.itemsize, n)


@pytest.mark.parametrize("dtype", SMALL_DTYPES)
def test_narrow_key_types(gpu, oracle, dtype):
    """u8 / u16 / i8 / i16 (src/radix_key_impl.rs:3-19, :87-103): one- and two-level keys."""
    levels = np.dtype(dtype).itemsize
    for i, n in enumerate((0, 1, 2, 129, 18431, 18432, 18433, 300_000, 4_000_001)):
        a = random_bits(n, dtype, seed=300 + i).copy()
        exp = a.copy()
        oracle.sort(exp, threads=4)
        assert same_bits(_device_sort(gpu, a), exp), (dtype, n)
        assert same_bits(exp, reference_sorted(a))
    a = random_bits(1_000_000, dtype, seed=9).copy()
    t = to_device(a)
    for level in range(levels):
        c, srt, first, last = gpu.level_counts(t, level)
        oc, osrt, ofirst, olast = oracle.get_counts_with_ends(a, level)
        assert np.array_equal(np.asarray(c, dtype=np.uint64), oc) and (srt, first, last) == (osrt, ofirst, olast)
        dst, _ = gpu.scatter_level(t, level)
        exp, _ = oracle.out_of_place_sort(a, level, "plain")
        assert same_bits(to_host(dst, dtype), exp)
    h = a.copy()
    gpu.radix_sort_unstable(h)  # host entry point
    assert same_bits(h, reference_sorted(a))


@pytest.mark.parametrize("kind", ["u128", "i128"])
def test_128_bit_keys(gpu, oracle, kind):
    """u128 / i128 (src/radix_key_impl.rs:39-46, :123-130): 16 levels, keys as [low, high] limbs."""
    import torch
    rng = np.random.default_rng(128)
    for n in (0, 1, 2, 129, 4607, 4608, 4609, 100_003, 2_000_001):
        a = rng.integers(0, 1 << 64, size=(n, 2), dtype=np.uint64)
        if n > 1000:
            a[: n // 2, 1] = a[0, 1]  # equal high limbs: the low levels decide
        exp = a.copy()
        oracle.sort(exp, threads=4, kind=kind)
        t = torch.from_numpy(a.view(np.int64).copy()).cuda()
        gpu.radix_sort_unstable(t, key=kind)
        assert np.array_equal(t.cpu().numpy().view(np.uint64), exp), (kind, n)
        if n in (129, 100_003):
            h = a.copy()
            gpu.radix_sort_unstable(h, key=kind)  # host entry point
            assert np.array_equal(h, exp)
            vals = [(int(hi) << 64) | int(lo) for lo, hi in exp.tolist()]
            key = (lambda v: v ^ (1 << 127)) if kind == "i128" else (lambda v: v)
            assert vals == sorted(vals, key=key)  # oracle vs Python big integers


@pytest.mark.parametrize("dtype,shift", [("uint32", 16), ("uint64", 32), ("int32", 16), ("int64", 32)])
def test_bimodal_shift_inputs(gpu, oracle, dtype, shift):
    """gen_inputs (src/test_utils.rs:51-61): empty high / low levels -> level skipping on device."""
    for n in (10_000, 500_000, 2_000_000):
        a = gen_inputs(n, shift, dtype, seed=n)
        exp = a.copy()
        oracle.sort(exp, threads=4)
        assert same_bits(_device_sort(gpu, a), exp), (dtype, n)


def test_u32_patterns_on_device(gpu):
    """validate_u32_patterns (src/test_utils.rs:148-262)."""
    for a in u32_patterns():
        assert np.array_equal(_device_sort(gpu, a), np.sort(a))


@pytest.mark.parametrize("dtype", ["uint32", "uint64"])
def test_degenerate_orders(gpu, dtype):
    n = 1_000_000
    base = random_bits(n, dtype, seed=1).copy()
    for name, a in (("all_equal", np.full(n, 0xABCD1234, dtype=dtype)), ("sorted", np.sort(base)),
                    ("reverse", np.sort(base)[::-1].copy()), ("two_values", (base & 1).astype(dtype)),
                    ("one_digit_differs", (base & 0xFF00).astype(dtype))):
        assert np.array_equal(_device_sort(gpu, a), np.sort(a)), name


@pytest.mark.parametrize("dtype", ["uint32", "uint64", "float32"])
def test_already_sorted_detection_sees_every_adjacent_pair(gpu, dtype):
    """The histogram sweep skips every pass when it finds no inversion (the whole-slice form of
    rdst's already_sorted exits, src/sorter.rs:59-65).  One swapped neighbour pair anywhere —
    inside a 16-byte vector, across lanes, waves, sweeps, block pieces, at either end — must
    still be found."""
    n = 5_000_003
    base = reference_sorted(random_bits(n, dtype, seed=77))
    assert same_bits(_device_sort(gpu, base), base)          # sorted stays sorted
    vec = 16 // np.dtype(dtype).itemsize
    piece = -(-n // 256)                                      # ~ one block's share of the sweep
    spots = [0, 1, vec - 1, vec, 64 * vec - 1, 64 * vec, 1024 * vec - 1, 1024 * vec, 4 * 1024 * vec - 1, 4 * 1024 * vec,
             piece - 1, piece, piece + 1, 2 * piece, n // 2, n - 3, n - 2]
    k = mapped_key(base)
    for i in spots:
        if k[i] == k[i + 1]:
            continue
        a = base.copy()
        a[i], a[i + 1] = base[i + 1], base[i]
        assert same_bits(_device_sort(gpu, a), base), (dtype, i)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_float_specials(gpu, oracle, dtype):
    sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -np.nan, 1.0, -1.0, 5e-324, -5e-324, 1e30, -1e30], dtype=dtype)
    a = np.tile(sp, 1000)
    exp = a.copy()
    oracle.sort(exp)
    got = _device_sort(gpu, a)
    assert same_bits(got, exp)
    k = mapped_key(got)
    assert (k[1:] >= k[:-1]).all() and np.isnan(got[0]) and np.signbit(got[0]) and np.isnan(got[-1])


def test_unaligned_device_pointer(gpu, oracle):
    """A slice that starts 4 bytes into an allocation: element-aligned only (scalar-load kernels)."""
    import torch
    a = random_bits(100_001, "uint32", seed=2).copy()
    t = to_device(a)
    view = t[1:]
    assert view.data_ptr() % 16 != 0
    gpu.sort_device_tensor(view)
    got = to_host(t, "uint32")
    exp = a[1:].copy()
    oracle.sort(exp)
    assert got[0] == a[0] and np.array_equal(got[1:], exp)


# ---- parity hooks: K1 / K6 vs get_counts_with_ends, K3 vs out_of_place_sort ------------------

@pytest.mark.parametrize("dtype", DTYPES)
def test_level_counts_hook_matches_oracle(gpu, oracle, dtype):
    levels = np.dtype(dtype).itemsize
    for n in (1, 1000, 400_001):
        for a in (random_bits(n, dtype, seed=n).copy(), reference_sorted(random_bits(n, dtype, seed=n + 1))):
            t = to_device(a)
            allc = gpu.all_level_counts(t)
            for level in range(levels):
                c, srt, first, last = gpu.level_counts(t, level)
                oc, osrt, ofirst, olast = oracle.get_counts_with_ends(a, level)
                assert np.array_equal(np.asarray(c, dtype=np.uint64), oc), (dtype, n, level)
                assert np.array_equal(allc[level], oc)
                assert (srt, first, last) == (osrt, ofirst, olast), (dtype, n, level)
    c, srt, first, last = gpu.level_counts(to_device(np.zeros(0, dtype=dtype)), 0)
    assert sum(c) == 0 and srt and (first, last) == (0, 0)  # sort_utils.rs:116-118


@pytest.mark.parametrize("dtype", ["uint32", "uint64", "float32", "int64"])
def test_scatter_level_hook_is_the_reference_stable_pass(gpu, oracle, dtype):
    """One K3 launch == out_of_place_sort on that digit, bit for bit (stability makes it unique)."""
    levels = np.dtype(dtype).itemsize
    for n in (2, 9, 8193, 250_000, 3_000_001):
        a = random_bits(n, dtype, seed=n).copy()
        t = to_device(a)
        for level in (0, levels // 2, levels - 1):
            dst, counts = gpu.scatter_level(t, level)
            exp, _ = oracle.out_of_place_sort(a, level, "plain")
            assert same_bits(to_host(dst, dtype), exp), (dtype, n, level)
            oc, _, _, _ = oracle.get_counts_with_ends(a, level)
            assert np.array_equal(counts, oc)
        assert same_bits(to_host(t, dtype), a)  # source untouched


def _heavy_digit_inputs(n, dtype, level, seed):
    """inputs that put the lr variants of out_of_place_sort to work (src/sorts/out_of_place_sort.rs:202-389: a
    bucket written from both ends when many equal digits sit next to each other) and the heavy-digit ranking of K3"""
    rng = np.random.default_rng(seed)
    w = np.dtype(dtype).itemsize * 8
    ut = f"uint{w}"
    base = random_bits(n, ut, seed).copy()
    sh = np.array(8 * level, dtype=ut)
    clear = ~(np.array(0xFF, dtype=ut) << sh)

    def with_digits(d):
        return ((base & clear) | (d.astype(ut) << sh)).view(dtype)

    out = {}
    d = rng.integers(0, 256, size=n)
    d[rng.random(n) < 0.9] = 0x5A
    out["90% one digit"] = with_digits(d)
    out["two digits only"] = with_digits(rng.choice([3, 200], size=n))
    out["one digit per 64-key round"] = with_digits(np.repeat(rng.integers(0, 256, size=(n + 63) // 64), 64)[:n])
    out["sorted by digit"] = with_digits(np.sort(rng.integers(0, 256, size=n)))
    out["long equal runs"] = with_digits(np.repeat(rng.integers(0, 256, size=(n + 999) // 1000), 1000)[:n])
    return out


@pytest.mark.parametrize("dtype", ("uint32", "int64", "float32"))
def test_scatter_level_hook_on_heavy_digits_matches_every_oracle_variant(gpu, oracle, dtype):
    """the reference's four out_of_place_sort variants (plain, with_counts, lr,
    lr_with_counts) produce ONE result — each is a stable pass — and K3 must produce the same, also when one
    digit crowds a wave (heavy-digit / single-digit-round ranking instead of the per-key LDS atomics).  The
    next-level counts the *_with_counts variants return are checked against the all-levels hook (K1)."""
    levels = np.dtype(dtype).itemsize
    for n in (70_001, 1_200_007):
        for level in (0, levels - 1):
            for name, a in _heavy_digit_inputs(n, dtype, level, seed=n + level).items():
                t = to_device(a)
                dst, counts = gpu.scatter_level(t, level)
                got = to_host(dst, dtype)
                all_counts = gpu.all_level_counts(t)
                for variant in ("plain", "with_counts", "lr", "lr_with_counts"):
                    if variant.endswith("with_counts") and level + 1 >= levels:
                        continue  # there is no next level to count (src/sorts/lsb_sort.rs:88-92 never asks for it)
                    exp, next_counts = oracle.out_of_place_sort(a, level, variant)
                    assert same_bits(got, exp), (dtype, n, level, name, variant)
                    if next_counts is not None and level + 1 < levels:
                        assert np.array_equal(next_counts, all_counts[level + 1]), (dtype, n, level, name, variant)
                oc, _, _, _ = oracle.get_counts_with_ends(a, level)
                assert np.array_equal(counts, oc), (dtype, n, level, name)


def test_custom_tuner_round_trip(gpu):
    """with_tuner: pick_algorithm sees the top-level histogram of the slice (src/sorter.rs:67-76)."""
    from rdst_amd.tuner import Algorithm, StandardTuner, Tuner
    seen = {}

    class T(Tuner):
        def pick_algorithm(self, p, counts):
            seen["p"], seen["sum"] = p, sum(counts)
            return Algorithm.GpuLsd

    a = random_bits(200_000, "uint32", seed=8).copy()
    t = to_device(a)
    gpu.radix_sort_builder(t).with_tuner(T()).sort()
    assert np.array_equal(to_host(t, "uint32"), np.sort(a))
    assert seen["sum"] == 200_000 and seen["p"].level == 3 and seen["p"].parent_len is None
    with pytest.raises(NotImplementedError):
        gpu.radix_sort_builder(to_device(a)).with_tuner(StandardTuner()).sort()  # CPU algorithm: not shipped here


def test_two_streams_share_the_workspace_safely(gpu):
    import torch
    a = random_bits(2_000_000, "uint32", seed=21).copy()
    b = random_bits(2_000_000, "uint64", seed=22).copy()
    ta, tb = to_device(a), to_device(b)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        with torch.cuda.stream(s1):
            gpu.sort_device_tensor(ta, check=False)
        with torch.cuda.stream(s2):
            gpu.sort_device_tensor(tb, check=False)
    torch.cuda.synchronize()
    gpu.device_status()
    assert np.array_equal(to_host(ta, "uint32"), np.sort(a)) and np.array_equal(to_host(tb, "uint64"), np.sort(b))
