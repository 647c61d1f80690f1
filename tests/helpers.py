"""Shared test helpers: an independent numpy statement of the mapped-key order, seeded input
generators shaped like the reference's test inputs (src/test_utils.rs), torch<->numpy glue."""
import numpy as np

SEED_C1 = 0x5D570001
SEED_C2 = 0x5D570002
SEED_C3 = 0x5D570003
SEED_C4 = 0x5D570004

DTYPES = ("uint32", "uint64", "int32", "int64", "float32", "float64")
SMALL_DTYPES = ("uint8", "uint16", "int8", "int16")


def uint_view(a):
    return a.view(f"u{a.dtype.itemsize}")


def mapped_key(a):
    """Order-preserving unsigned image of a built-in key type — numpy only, independent of the
    oracle and of the device code (formulae of src/radix_key_impl.rs)."""
    u = uint_view(a)
    w = a.dtype.itemsize * 8
    msb = np.array(1 << (w - 1), dtype=u.dtype)
    if a.dtype.kind == "u":
        return u
    if a.dtype.kind == "i":
        return u ^ msb
    neg = (u >> np.array(w - 1, dtype=u.dtype)) != 0
    return np.where(neg, ~u, u ^ msb)


def reference_sorted(a):
    """THE output of radix_sort_unstable() for a built-in key type: unique because the key map
    is a bijection on the value's bits."""
    order = np.argsort(mapped_key(a), kind="stable")
    return a[order]


def same_bits(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(uint_view(a), uint_view(b))


def random_bits(n, dtype, seed):
    """uniform random BIT PATTERNS of the type (floats: NaNs, infs, +-0, denormals all occur,
    like block_rand::<f32> in src/radix_sort.rs:133)"""
    rng = np.random.default_rng(seed)
    dt = np.dtype(dtype)
    u = rng.integers(0, 1 << (8 * dt.itemsize), size=n, dtype=f"u{dt.itemsize}", endpoint=False) \
        if dt.itemsize < 8 else rng.integers(0, 1 << 64, size=n, dtype=np.uint64, endpoint=False)
    return u.view(dt)


def gen_inputs(n, shift, dtype, seed):
    """gen_inputs (src/test_utils.rs:51-61): random, first half >>= shift, second half <<= shift."""
    a = random_bits(n, dtype, seed).copy()
    u = uint_view(a)
    if shift:
        s = np.array(shift, dtype=u.dtype)
        half = n // 2
        if a.dtype.kind == "i":  # Rust >> on iN is arithmetic
            si = a[:half] >> np.array(shift, dtype=a.dtype)
            a[:half] = si
        else:
            u[:half] >>= s
        u[half:] <<= s
    return a


# the 17 lengths of gen_input_set (src/test_utils.rs:63-95), capped for CI time
INPUT_SET_LENGTHS = (0, 1, 10, 100, 5_000, 10_000, 50_000, 100_000, 200_000, 300_000, 500_000, 1_000_000, 2_000_000)


def u32_patterns(seed=7):
    """validate_u32_patterns (src/test_utils.rs:148-262): 4 base inputs x 14 transforms."""
    rng = np.random.default_rng(seed)
    bases = [np.full(128, 0xFFFFFFFF, dtype=np.uint32),
             rng.integers(0, 1 << 32, size=128, dtype=np.uint32),
             rng.integers(0, 1 << 32, size=128_000, dtype=np.uint32),
             rng.integers(0, 1 << 32, size=4, dtype=np.uint32)]
    masks = [0x000000FF, 0x0000FF00, 0x00FF0000, 0xFF000000, 0x00FFFF00, 0xFF0000FF,  # byte-lane masks
             0x80000000, 0x00000001, 0xFFFFFFFE, 0x7FFFFFFF, 0xAAAAAAAA, 0x55555555]
    out = []
    for b in bases:
        for m in masks:
            out.append(b & np.uint32(m))
        out.append(b.copy())
        out.append(np.array([1, 2, 3, 4, 0xFFFFFFFF], dtype=np.uint32))  # the 5-element skew case
    return out


def to_device(a):
    import torch
    t = torch.from_numpy(a.view({1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}[a.dtype.itemsize]).copy()).cuda()
    return t.view(getattr(torch, a.dtype.name))


def to_host(t, dtype):
    import torch
    it = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[np.dtype(dtype).itemsize]
    return t.view(it).cpu().numpy().view(dtype)
