"""Host-side mirror of rdst's surface (rdst_amd/radix_sort.py, tuner.py, sharded.py): logic that
needs no device."""
import numpy as np
import pytest


def test_key_info_table():
    import rdst_amd
    assert rdst_amd.key_info("uint32") == (0, 4, 4)
    assert rdst_amd.key_info("int64") == (1, 8, 8)
    assert rdst_amd.key_info("float32") == (2, 4, 4)
    assert rdst_amd.key_info("torch.float64") == (2, 8, 8)
    assert rdst_amd.key_info("uint16") == (0, 2, 2) and rdst_amd.key_info("int8") == (1, 1, 1)
    with pytest.raises(TypeError):
        rdst_amd.key_info("float16")  # no RadixKey for it in the reference either


def test_len_le_1_is_a_noop_without_a_device(hiplib):
    import rdst_amd
    a = np.array([7], dtype=np.uint32)
    rdst_amd.radix_sort_unstable(a)  # radix_sort_builder.rs:151
    rdst_amd.radix_sort_builder(np.zeros(0, dtype=np.float64)).with_parallel(False).sort()
    assert a[0] == 7


def test_no_cpu_fallback(hiplib):
    """The device route must fail loudly, never sort on the CPU."""
    import torch
    import rdst_amd
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    a = np.array([3, 1, 2], dtype=np.uint32)
    with pytest.raises(rdst_amd.RdstHipError):
        rdst_amd.radix_sort_unstable(a)
    assert a.tolist() == [3, 1, 2]  # untouched on failure (include/rdst_hip.h contract)


def test_builder_rejects_non_slices(hiplib):
    import rdst_amd
    with pytest.raises(ValueError):
        rdst_amd.sort_host_array(np.zeros((4, 4), dtype=np.uint32))
    with pytest.raises(ValueError):
        rdst_amd.sort_host_array(np.arange(10, dtype=np.uint32)[::2])
    with pytest.raises(TypeError):
        rdst_amd.radix_sort_builder(np.zeros(4, dtype=np.uint32)).with_tuner(object())


def test_product_code_never_imports_the_oracle():
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "rdst_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert "librdst_oracle" not in text and "rdst_oracle.h" not in text, f


def test_split_digits_balances_and_is_contiguous():
    from rdst_amd.sharded import split_digits
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 4, 8):
        for counts in ([1000] * 256, rng.integers(0, 5000, size=256).tolist(), [0] * 255 + [10**6], [10**6] + [0] * 255,
                       [0] * 256):
            owner = split_digits(counts, world)
            assert len(owner) == 256 and all(0 <= o < world for o in owner)   # (a bucket goes where its middle key falls: one bucket holding everything is the middle rank's)
            assert all(b - a in (0, 1) or b > a for a, b in zip(owner, owner[1:])) and owner == sorted(owner)
            if counts == [1000] * 256 and 256 % world == 0:   # uniform keys: ranges end on multiples of 256 / world digits
                assert all(owner[d] == d // (256 // world) for d in range(256))
            if counts == [1000] * 256:
                per = [sum(c for c, o in zip(counts, owner) if o == r) for r in range(world)]
                assert max(per) - min(per) <= 1000 * (256 % world != 0) + 1000
