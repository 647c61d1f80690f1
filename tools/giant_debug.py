"""Development aid: which kind of giant bucket breaks the hybrid route?"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import rdst_amd
from helpers import random_bits, reference_sorted, same_bits, to_device, to_host

rng = np.random.default_rng(2024)


def bucket(prefix, size, kind):
    if kind == "dense":
        low = rng.integers(0, 1 << 16, size=size, dtype=np.uint32)
    elif kind == "sparse":
        low = rng.choice(np.array([0, 1, 300, 30000, 32767, 32768, 65000, 65535], dtype=np.uint32), size=size)
    elif kind == "two":
        low = rng.choice(np.array([0, 65535], dtype=np.uint32), size=size)
    elif kind == "one":
        low = np.full(size, 0x8000, dtype=np.uint32)
    else:
        low = rng.integers(1000, 1100, size=size, dtype=np.uint32)
    return low | np.uint32(prefix << 16)


cases = {
    "dense@1234 300k": [bucket(0x1234, 300_000, "dense")],
    "dense@0000 65536": [bucket(0x0000, 65_536, "dense")],
    "sparse@FFFF": [bucket(0xFFFF, 70_001, "sparse")],
    "two@1235": [bucket(0x1235, 65_537, "two")],
    "one@8000": [bucket(0x8000, 131_072, "one")],
    "narrow@7FFF 1.1M": [bucket(0x7FFF, 1_100_000, "narrow")],
    "pair 1234+1235": [bucket(0x1234, 300_000, "dense"), bucket(0x1235, 65_537, "two")],
}
allp = {"a": bucket(0x0000, 65_536, "dense"), "b": bucket(0xFFFF, 70_001, "sparse"), "c": bucket(0x1234, 300_000, "dense"),
        "d": bucket(0x1235, 65_537, "two"), "e": bucket(0x8000, 131_072, "one"), "f": bucket(0x7FFF, 1_100_000, "narrow"),
        "g": bucket(0x4000, 65_535, "dense"), "h": bucket(0x4001, 20_000, "two")}
cases = {"all": list(allp.values())}
for k in allp:
    cases["without " + k] = [v for kk, v in allp.items() if kk != k]
cases["f+g"] = [allp["f"], allp["g"]]
cases["f+e"] = [allp["f"], allp["e"]]
cases["a+b+c"] = [allp["a"], allp["b"], allp["c"]]
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 7
rdst_amd.set_hybrid(mode, 1)
dtypes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["uint32"]
for name, parts in [(n_ + " " + d_, p_) for n_, p_ in cases.items() for d_ in dtypes]:
    for shuffle in (True, False):
        a = np.concatenate(parts + [random_bits(500_000, "uint32", seed=5)])
        if shuffle:
            rng.shuffle(a)
        a = a.view(name.split()[-1])
        t = to_device(a)
        try:
            rdst_amd.sort_device_tensor(t)
            ok = same_bits(to_host(t, a.dtype), reference_sorted(a))
            print(f"{name:22s} shuffle={shuffle}: route={rdst_amd.last_route()} ok={ok}", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"{name:22s} shuffle={shuffle}: {str(e)[:120]}", flush=True)
