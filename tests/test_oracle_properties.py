"""The reference's own property suites (src/test_utils.rs, src/radix_sort.rs:146-340 and the
per-algorithm test modules under src/sorts/) re-created with seeded inputs and run against the
oracle; the second opinion is numpy's sort on the independently mapped key."""
import numpy as np
import pytest

from helpers import (INPUT_SET_LENGTHS, gen_inputs, mapped_key, random_bits, reference_sorted, same_bits, u32_patterns)

ALGOS = ["MtOop", "MtLsb", "Scanning", "Recombinating", "Comparative", "LrLsb", "Lsb", "Regions", "Ska"]
INT_TYPES = [("uint8", 0), ("uint16", 8), ("uint32", 16), ("uint64", 32), ("int8", 0), ("int16", 8), ("int32", 16), ("int64", 32)]


@pytest.mark.parametrize("dtype,shift", INT_TYPES)
@pytest.mark.parametrize("tuner", ["standard", "low_memory"])
def test_full_sort_integers(oracle, dtype, shift, tuner):
    """test_full_sort_* (src/radix_sort.rs:146-219): sort_comparison_suite through the public API."""
    for i, n in enumerate(INPUT_SET_LENGTHS[:-2] if tuner == "low_memory" else INPUT_SET_LENGTHS):
        for sh in (0, shift):
            a = gen_inputs(n, sh, dtype, seed=1000 + i)
            exp = reference_sorted(a)
            oracle.sort(a, tuner=tuner, threads=4)
            assert same_bits(a, exp), (dtype, n, sh)


@pytest.mark.parametrize("kind", ["u128", "i128"])
def test_full_sort_128bit(oracle, kind):
    rng = np.random.default_rng(5)
    for n in (0, 1, 100, 129, 5_000, 300_000):
        a = rng.integers(0, 1 << 64, size=(n, 2), dtype=np.uint64)
        vals = [(int(h) << 64) | int(l) for l, h in a.tolist()]
        if kind == "i128":
            key = lambda v: v ^ (1 << 127)  # noqa: E731
        else:
            key = lambda v: v  # noqa: E731
        exp = sorted(vals, key=key)
        oracle.sort(a, threads=4, kind=kind)
        got = [(int(h) << 64) | int(l) for l, h in a.tolist()]
        assert got == exp, (kind, n)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_floats_total_order(oracle, dtype):
    """src/radix_sort.rs:97-144: floats (any bit pattern) come out in total_cmp order."""
    for n in (10, 1_000, 200_000, 1_500_000):
        a = random_bits(n, dtype, seed=n).copy()
        exp = reference_sorted(a)
        oracle.sort(a, threads=4)
        assert same_bits(a, exp), (dtype, n)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -np.nan, 1.0, -1.0, 5e-324, -5e-324], dtype=dtype)
    k = mapped_key(special)
    oracle.sort(special)
    assert np.array_equal(mapped_key(special), np.sort(k))
    assert np.signbit(special[0]) and np.isnan(special[0]) and np.isnan(special[-1])  # -NaN first, +NaN last


@pytest.mark.parametrize("algo", ALGOS)
def test_each_algorithm_suite(oracle, algo):
    """The per-algorithm `test_*_sort` modules: every algorithm, several types, bimodal shifts."""
    sizes = (0, 1, 10, 100, 5_000, 50_000, 300_000, 1_000_000)
    for dtype, shift in (("uint8", 0), ("uint16", 8), ("uint32", 16), ("uint64", 32), ("int32", 16), ("float32", 0)):
        for i, n in enumerate(sizes):
            a = gen_inputs(n, shift, dtype, seed=77 + i)
            exp = reference_sorted(a)
            oracle.sort_single_algorithm(a, algo, threads=4)
            assert same_bits(a, exp), (algo, dtype, n)


@pytest.mark.parametrize("algo", ALGOS)
def test_u32_patterns(oracle, algo):
    """validate_u32_patterns (src/test_utils.rs:148-262)."""
    for a in u32_patterns():
        exp = np.sort(a)
        b = a.copy()
        oracle.sort_single_algorithm(b, algo, threads=4)
        assert np.array_equal(b, exp)


def test_mt_lsb_single_tile_regression(oracle):
    """src/sorts/mt_lsb_sort.rs:323-328: one tile, 400 elements."""
    a = random_bits(400, "uint32", seed=3).copy()
    out = oracle.mt_lsb_sort(a, tile_size=400, level=0, threads=2)
    assert np.array_equal(out, a[np.argsort(a & 0xFF, kind="stable")])


def test_lsb_adapter_as_reference_unit_tests_call_it(oracle):
    """src/sorts/lsb_sort.rs:153-165: lsb_sort_adapter(false, inputs, &counts, 0, LEVELS-1)."""
    for dtype, levels, shift in (("uint8", 1, 0), ("uint16", 2, 8), ("uint32", 4, 16), ("uint64", 8, 32)):
        for lr in (False, True):
            a = gen_inputs(200_000, shift, dtype, seed=levels)
            exp = np.sort(a)
            oracle.lsb_sort_adapter(a, 0, levels - 1, lr=lr)
            assert np.array_equal(a, exp)


@pytest.mark.parametrize("variant", ["plain", "with_counts", "lr", "lr_with_counts"])
def test_out_of_place_variants_are_one_stable_pass(oracle, variant):
    """All four scatters (src/sorts/out_of_place_sort.rs) produce the stable counting sort of one digit."""
    for n in (0, 1, 2, 7, 8, 9, 1000, 100_003):
        a = random_bits(n, "uint32", seed=n).copy()
        for level in (0, 2):
            dst, nxt = oracle.out_of_place_sort(a, level, variant)
            exp = a[np.argsort((a >> (8 * level)) & 0xFF, kind="stable")]
            assert np.array_equal(dst, exp), (variant, n, level)
            if nxt is not None and n >= 2:
                assert np.array_equal(nxt, np.bincount((a >> (8 * (level + 1))) & 0xFF, minlength=256))


def test_counts_with_ends_and_parallel_form_agree(oracle):
    """get_counts_with_ends vs par_get_counts_with_ends (src/sort_utils.rs:35-180), incl. sortedness stitch."""
    rng = np.random.default_rng(11)
    n = 1_200_000  # above the 400 000 serial cut-off (:42)
    for a in (rng.integers(0, 1 << 32, size=n, dtype=np.uint32), np.sort(rng.integers(0, 1 << 32, size=n, dtype=np.uint32)),
              np.arange(n, dtype=np.uint32)):
        for level in (0, 3):
            c1, s1, f1, l1 = oracle.get_counts_with_ends(a, level)
            c2, s2, f2, l2 = oracle.get_counts_with_ends(a, level, threads=4)
            d = (a >> (8 * level)) & 0xFF
            assert np.array_equal(c1, np.bincount(d, minlength=256)) and np.array_equal(c1, c2)
            assert s1 == s2 == bool((d[1:] >= d[:-1]).all())
            assert (f1, l1) == (f2, l2) == (int(d[0]), int(d[-1]))
    c, s, f, l = oracle.get_counts_with_ends(np.zeros(0, dtype=np.uint32), 0)
    assert c.sum() == 0 and s and (f, l) == (0, 0)  # sort_utils.rs:116-118


def test_custom_tuner_always_lsb(oracle):
    """src/radix_sort.rs:75-92, :319-327: a user Tuner that always answers Lsb."""
    for dtype in ("uint32", "uint64"):
        a = random_bits(300_000, dtype, seed=9).copy()
        exp = np.sort(a)
        seen = []
        oracle.sort_with_tuner(a, lambda p, c: (seen.append(p["level"]), "Lsb")[1], threads=2)
        assert np.array_equal(a, exp) and seen == [np.dtype(dtype).itemsize - 1]


def test_regions_on_f64_single_threaded_pool(oracle):
    """src/radix_sort.rs:331-340: f64 through Regions with with_parallel(false) (reference: no-panic only)."""
    a = random_bits(1_000_000, "float64", seed=4).copy()
    exp = reference_sorted(a)
    oracle.sort_with_tuner(a, lambda p, c: "Regions" if p["input_len"] > 128 else "Comparative", multi_threaded=False, threads=1)
    assert same_bits(a, exp)
