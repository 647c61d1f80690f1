"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header declares,
validates arguments without touching a device, and its tuner tables agree with the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rdst_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rdst_[a-z0-9_]+)\s*\(", text)))


def test_exports_every_declared_symbol(hiplib):
    from rdst_amd import _lib
    declared = _declared_functions()
    assert declared and set(declared) == set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(hiplib, name) is not None


def test_abi_version(hiplib):
    assert hiplib.rdst_hip_abi_version() == 2


def test_argument_validation_happens_before_any_device_call(hiplib):
    from rdst_amd import _lib
    buf = (ctypes.c_uint32 * 8)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    # LEVELS == 0 is a panic in the reference (src/radix_sort_builder.rs:22)
    assert hiplib.rdst_hip_sort(p, 8, 4, 0, 0, None) == -1
    assert b"level" in hiplib.rdst_hip_last_error()
    assert hiplib.rdst_hip_sort(p, 8, 3, 0, 3, None) == -2          # unsupported width
    assert hiplib.rdst_hip_sort(p, 8, 2, 2, 2, None) == -2          # no 2-byte float key
    assert hiplib.rdst_hip_sort(p, 8, 4, 7, 4, None) == -1          # unknown kind
    assert hiplib.rdst_hip_sort(None, 8, 4, 0, 4, None) == -1       # null pointer
    assert hiplib.rdst_hip_sort(ctypes.c_void_p(p.value + 2), 4, 4, 0, 4, None) == -6  # misaligned
    # len <= 1 is a no-op (src/radix_sort_builder.rs:151) and needs no device
    assert hiplib.rdst_hip_sort(p, 1, 4, 0, 4, None) == 0
    assert hiplib.rdst_hip_sort(p, 0, 4, 0, 4, None) == 0
    assert hiplib.rdst_hip_sort_device(p, p, 1, 8, 1, 8, None) == 0
    with pytest.raises(_lib.RdstHipError):
        _lib.check(-1)


def test_workspace_size_is_reported(hiplib):
    small = hiplib.rdst_hip_workspace_bytes(1000, 4)
    big = hiplib.rdst_hip_workspace_bytes(1_000_000_000, 4)
    assert 0 < small < big
    # the atomic route's workspace: areas of pass A (the slice again, plus an eighth: 4.6 GB), one slot per 16-bit prefix (low
    # halves, mean + 10 sigma each: 2.2 GB) and the LSD fallback's status words (~0.5 GB)
    assert big < 7_500_000_000
    # 8-byte keys: areas 9.1 GB, slots of whole keys 8.6 GB, status rows of eight levels ~2.1 GB
    assert hiplib.rdst_hip_workspace_bytes(1_000_000_000, 8) < 21_000_000_000
    # below the routes' thresholds (3 * 2^26 u32 keys, 2^26 u64 keys) only the LSD route's tables: a fraction of the slice
    assert hiplib.rdst_hip_workspace_bytes(200_000_000, 4) < 200_000_000
    assert hiplib.rdst_hip_workspace_bytes(60_000_000, 8) < 200_000_000 < hiplib.rdst_hip_workspace_bytes(70_000_000, 8)
    assert hiplib.rdst_hip_workspace_bytes(10, 3) == 0 and hiplib.rdst_hip_workspace_bytes(10, 16) > 0


def test_tuner_tables_match_oracle_on_a_grid(hiplib, oracle):
    from rdst_amd.tuner import LowMemoryTuner, SingleThreadedTuner, StandardTuner, TuningParams
    rng = np.random.default_rng(1)
    lens = [1, 100, 128, 129, 4_999, 5_000, 50_000, 50_001, 100_001, 150_000, 150_001, 200_000, 200_001, 260_000, 260_001,
            350_000, 350_001, 800_000, 800_001, 1_000_000, 1_000_001, 4_000_000, 4_000_001, 5_000_000, 5_000_001,
            50_000_000, 50_000_001, 10**9]
    for name, T in (("standard", StandardTuner), ("low_memory", LowMemoryTuner), ("single_threaded", SingleThreadedTuner)):
        for n in lens:
            for level in (0, 1, 3):
                for skew in (False, True):
                    c = rng.multinomial(n, [1 / 256] * 256).tolist()
                    if skew:
                        c = [n - 255 * (n // 1024)] + [n // 1024] * 255
                    parent = None if level == 3 else n * 7
                    got = T().pick_algorithm(TuningParams(8, level, 4, n, parent), c).name
                    exp = oracle.pick_algorithm(name, 8, level, 4, n, parent, c)
                    assert got == exp, (name, n, level, skew)


def test_gpu_tuner_routes_whole_slices_to_the_device(hiplib):
    from rdst_amd.tuner import Algorithm, GpuTuner, TuningParams
    c = [10**9 // 256] * 256
    t = GpuTuner(gpu_min_len=1_000_000)
    assert t.pick_algorithm(TuningParams(8, 3, 4, 10**9, None), c) == Algorithm.GpuLsd
    assert t.pick_algorithm(TuningParams(8, 3, 4, 999_999, None), [999_999 // 256] * 256) == Algorithm.Recombinating
    assert t.pick_algorithm(TuningParams(8, 2, 4, 10**7, 10**9), [10**7 // 256] * 256) == Algorithm.Recombinating
    with pytest.raises(ValueError):
        t.pick_algorithm(TuningParams(8, 3, 4, 10, None), [1, 2, 3])
