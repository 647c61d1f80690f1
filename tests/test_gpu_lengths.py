"""Lengths between the parity suites' 5·10^6 and the full-size tests' 4·10^8 — where round 2's one silently wrong
result lived (surplus pass-B blocks at lengths where the grid exceeded areas x tiles per area; tools/stress.py found it, no
suite did).  A seeded set of lengths from [2^24, 2^29], chosen around every grid-size formula of the atomic route's two
launches (make_layout / run_pipeline in rdst_kernels.hip, restated below): whole tiles +- 1 key, the lengths at which the
tiles per area step up, and the lengths at which pass B's grid switches between `areas x tiles per area` and `tiles + 256`.
Every output is compared with torch.sort of the mapped keys — bit-exact, the key maps being bijections."""
import math
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LO, HI = 1 << 24, 1 << 29


def geometry(n, key_bytes):
    """the atomic route's launch geometry for n keys (make_layout, run_pipeline)"""
    tile = 12 * 64 * (22 if key_bytes == 4 else 11)
    tiles = -(-n // tile)
    slices = 8 if tiles >= 64 * 8 else 1
    mean = n / (256 * slices)
    slack = max(mean * 0.125, 8.0 * math.sqrt(mean))
    cap_a = (int(mean + slack) + 2 * tile // 256 + 64) // 64 * 64
    tpa = -(-cap_a // tile)
    return {"tile": tile, "tiles": tiles, "slices": slices, "cap_a": cap_a, "tpa": tpa, "areas_grid": 256 * slices * tpa,
            "grid_b": max(256 * slices * tpa, tiles + 256)}


def interesting_lengths(key_bytes, count, seed):
    """lengths in [LO, HI] next to a change of the geometry, plus a few whole-tile borders and random ones"""
    tile = geometry(LO, key_bytes)["tile"]
    edges = []
    prev = None
    for t in range(LO // tile, HI // tile + 1):
        g = geometry(t * tile, key_bytes)
        key = (g["tpa"], g["slices"], g["grid_b"] == g["areas_grid"])
        if prev is not None and key != prev:
            edges.append(t * tile)
        prev = key
    rng = random.Random(seed)
    picks = set()
    for e in edges:                      # both sides of every change, off the tile border by a key or a few thousand
        picks.add(e - rng.choice((1, tile // 2, tile + 1)))
        picks.add(e + rng.choice((0, 1, tile - 1)))
    while len(picks) < count:
        t = rng.randrange(LO // tile, HI // tile)
        picks.add(t * tile + rng.choice((-1, 0, 1, rng.randrange(tile))))
    picks = sorted(p for p in picks if LO <= p <= HI)
    rng.shuffle(picks)
    return sorted(picks[:count])


def _mapped(torch, x, kind):
    mn = torch.iinfo(x.dtype).min
    if kind == "u":
        return x ^ mn
    if kind == "i":
        return x
    return torch.where(x < 0, ~x ^ mn, x)


def _check(torch, gpu, name, n, seed, shape):
    itype = torch.int32 if np.dtype(name).itemsize == 4 else torch.int64
    g = torch.Generator(device="cuda").manual_seed(seed)
    info = torch.iinfo(itype)
    src = torch.randint(info.min, info.max, (n,), dtype=itype, device="cuda", generator=g)
    if shape == "half_top_bits":         # ids below half the range (from 2^26 keys up the sample lowers the window by a bit)
        src &= info.max
    elif shape == "one_heavy_byte":      # the first twentieth of the slice on one top byte: its areas overflow part-way
        sh = np.dtype(name).itemsize * 8 - 8
        src[: n // 20] = (src[: n // 20] & ((1 << sh) - 1)) | (0x47 << sh)
    keys = src.clone()
    gpu.sort_device_tensor(keys.view(getattr(torch, name)))
    route = gpu.last_route()
    kind = np.dtype(name).kind
    want = torch.sort(_mapped(torch, src, kind)).values
    got = _mapped(torch, keys, kind)
    assert bool(torch.equal(want, got)), (name, n, shape, route, geometry(n, np.dtype(name).itemsize))
    return route


@pytest.mark.parametrize("name,count,seed", [("uint32", 24, 0x5D570301), ("uint64", 10, 0x5D570302), ("float32", 10, 0x5D570303)])
def test_lengths_around_the_grid_formulas(gpu, name, count, seed):
    import torch
    kb = np.dtype(name).itemsize
    lengths = [n for n in interesting_lengths(kb, count, seed) if n * kb <= (1 << 32)]   # (u64: up to 2^29 keys = 4 GiB)
    assert len(lengths) >= count // 2
    routes = {}
    gpu.set_hybrid(True, 1)               # the routes of the 10^9-key workload, considered at every length
    try:
        for i, n in enumerate(lengths):
            shape = ("uniform", "half_top_bits", "uniform", "one_heavy_byte")[i % 4]
            r = _check(torch, gpu, name, n, seed + i, shape)
            routes[r] = routes.get(r, 0) + 1
            if shape == "uniform":
                assert r == "atomic", (name, n, r)
            if shape == "one_heavy_byte":
                assert r != "atomic", (name, n, r)
    finally:
        gpu.set_hybrid(True, 0)
    gpu.device_status()


def test_lengths_at_the_real_threshold(gpu):
    """the shipped setting (the atomic route is tried from 3 * 2^26 u32 keys and from 2^26 u64 keys up): the lengths right at
    the thresholds, and at 2^28, from where the K1h hybrid route may follow a failed attempt"""
    import torch
    t4, t8 = 3 << 26, 1 << 26
    for i, n in enumerate((t4 - 1, t4, t4 + 16_897, (1 << 28) - 1, 1 << 28, 300_000_017)):
        r = _check(torch, gpu, "uint32", n, 0x5D570310 + i, "uniform")
        assert r == ("lsd" if n < t4 else "atomic"), (n, r)
    for i, n in enumerate((t8 - 1, t8, t8 + 16_385, 100_000_003)):
        r = _check(torch, gpu, "uint64", n, 0x5D570318 + i, "uniform")
        assert r == ("lsd" if n < t8 else "atomic"), (n, r)
    gpu.device_status()


def test_overflow_raised_while_hundreds_of_tiles_run(gpu):
    """pass A's early exit on the overflow flag must be the BLOCK's decision (the flag goes up while the kernel
    runs; waves that left alone used to leave stale wave tables behind, and the survivors claimed garbage).  Below 2^28 keys
    there is no sample, so a skewed top byte is found out only by an area overflowing mid-pass — with ~1 800 tiles in
    flight here.  The sort must fall to the next route and still be exact; no device error."""
    import torch
    n = 30_000_001
    g = torch.Generator(device="cuda").manual_seed(0x5D570320)
    src = torch.randint(-(2**31), 2**31, (n,), dtype=torch.int32, device="cuda", generator=g)
    third = torch.arange(n, device="cuda") % 3 == 0
    src = torch.where(third, (src & 0x00FFFFFF) | (0x47 << 24), src)   # a third of the keys on one top byte, spread evenly
    del third
    gpu.set_hybrid(True, 1)
    try:
        for _ in range(3):
            keys = src.clone()
            gpu.sort_device_tensor(keys.view(torch.uint32))
            assert gpu.last_route() != "atomic"
            want = torch.sort(src ^ torch.iinfo(torch.int32).min).values
            assert bool(torch.equal(want, keys ^ torch.iinfo(torch.int32).min))
    finally:
        gpu.set_hybrid(True, 0)
    gpu.device_status()


@pytest.mark.parametrize("name", ["uint64", "int64", "float64"])
def test_wide_k4_settles_keys_that_agree_on_32_bits_below_the_bucket(gpu, name):
    """8-byte K4 counts bits [32, 48) and orders the keys of one value by bits [16, 32); members that agree on those as well used to
    send the whole bucket to the generic kernel.  Now they enter a list and are ordered by (low 16 bits, slot).  Keys built so that
    every bucket (~500 keys) holds ~64 groups of ~8 with staged bits from 64 values and low bits from 4: dozens of such members
    per bucket, among them keys that are equal in all 64 bits.  A second input makes the lists overflow (every group agrees on
    its staged bits): those buckets must still come out right (generic kernel)."""
    import torch
    n = 32_000_000
    g = torch.Generator(device="cuda").manual_seed(0x5D570330)
    r = torch.randint(-(2**63), 2**63 - 1, (n,), dtype=torch.int64, device="cuda", generator=g)
    top = r & (0xFFFF << 48) if False else (r >> 48) << 48            # random top 16 bits (sign included)
    v16 = ((r >> 40) & 63) * 1009 % 65536                              # 64 values of bits [32, 48)
    low = (r >> 8) & 3                                                 # 4 values of the low 16 bits
    for mids, want_route in ((64, "atomic"), (1, "atomic")):
        mid = ((r >> 20) & (mids - 1)) * 257
        src = top | (v16 << 32) | (mid << 16) | low
        keys = src.clone()
        gpu.set_hybrid(True, 1)
        try:
            gpu.sort_device_tensor(keys.view(getattr(torch, name)))
            assert gpu.last_route() == want_route
        finally:
            gpu.set_hybrid(True, 0)
        kind = np.dtype(name).kind
        want = torch.sort(_mapped(torch, src, kind)).values
        assert bool(torch.equal(want, _mapped(torch, keys, kind))), (name, mids)
    gpu.device_status()
