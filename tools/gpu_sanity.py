"""Development aid (not a test, not the product): run the device sort over a few sizes and
dtypes on the GPU box, check with numpy / torch, print timings.  Usage:
    python tools/gpu_sanity.py [--big] [--cfg N] [--chains C]
"""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import rdst_amd  # noqa: E402


def mapped_np(a):
    """order-preserving map to unsigned (independent restatement for this tool only)"""
    if a.dtype.kind == "u":
        return a
    w = a.dtype.itemsize * 8
    u = a.view(f"u{a.dtype.itemsize}")
    msb = np.array(1 << (w - 1), dtype=u.dtype)
    if a.dtype.kind == "i":
        return u ^ msb
    neg = (u >> np.array(w - 1, dtype=u.dtype)).astype(bool)
    return np.where(neg, ~u, u ^ msb)


def gen(n, dtype, seed):
    rng = np.random.default_rng(seed)
    nb = np.dtype(dtype).itemsize
    raw = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) if nb == 8 else rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    if nb == 8:
        raw ^= rng.integers(0, 2, size=n, dtype=np.uint64) << np.uint64(63)
    return raw.view(dtype)


def to_dev(a):
    t = torch.from_numpy(a.view({4: np.int32, 8: np.int64}[a.dtype.itemsize])).cuda()
    return t.view(getattr(torch, a.dtype.name))


def check_small(n, dtype, seed):
    a = gen(n, dtype, seed)
    t = to_dev(a)
    rdst_amd.radix_sort_unstable(t)
    got = t.view({4: torch.int32, 8: torch.int64}[a.dtype.itemsize]).cpu().numpy().view(a.dtype)
    order = np.argsort(mapped_np(a), kind="stable")
    exp = a[order]
    ok = np.array_equal(got.view(f"u{a.dtype.itemsize}"), exp.view(f"u{a.dtype.itemsize}"))
    print(f"  n={n:>10} {np.dtype(dtype).name:>8}: {'OK' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        bad = np.nonzero(got.view(f"u{a.dtype.itemsize}") != exp.view(f"u{a.dtype.itemsize}"))[0]
        print("    first bad idx", bad[:5], "count", bad.size)
    return ok


def time_sort(n, dtype, iters=5):
    it = {4: torch.int32, 8: torch.int64}[np.dtype(dtype).itemsize]
    g = torch.Generator(device="cuda").manual_seed(1234)
    lo, hi = (-(2**31), 2**31) if it == torch.int32 else (-(2**63), 2**63 - 1)
    src = torch.randint(lo, hi, (n,), dtype=it, device="cuda", generator=g)
    keys = torch.empty_like(src)
    tmp = torch.empty_like(src)
    kv = keys.view(getattr(torch, np.dtype(dtype).name))
    tv = tmp.view(getattr(torch, np.dtype(dtype).name))
    times = []
    for i in range(iters + 1):
        keys.copy_(src)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rdst_amd.sort_device_tensor(kv, tv, check=False)
        e1.record()
        torch.cuda.synchronize()
        rdst_amd.device_status()
        if i:
            times.append(e0.elapsed_time(e1))
    # sortedness + checksum on device (mapped to a signed-comparable domain)
    if np.dtype(dtype).kind == "u":
        m = keys ^ (torch.iinfo(it).min)
        ms = src ^ (torch.iinfo(it).min)
    elif np.dtype(dtype).kind == "i":
        m, ms = keys, src
    else:
        m = torch.where(keys < 0, ~keys ^ torch.iinfo(it).min, keys)
        ms = torch.where(src < 0, ~src ^ torch.iinfo(it).min, src)
    sorted_ok = bool((m[1:] >= m[:-1]).all())
    sum_ok = int(keys.sum()) == int(src.sum()) and int((keys ^ (keys >> 7)).sum()) == int((src ^ (src >> 7)).sum())
    med = sorted(times)[len(times) // 2]
    nb = np.dtype(dtype).itemsize
    algo = n * nb * (2 * nb + 1)
    print(f"  n={n:>11} {np.dtype(dtype).name:>8}: {med:8.3f} ms  {n / med / 1e6:8.2f} Gkeys/s  "
          f"{algo / med / 1e9:7.3f} TB/s algorithmic  sorted={sorted_ok} multiset={sum_ok}  all={['%.2f' % x for x in times]}", flush=True)
    return sorted_ok and sum_ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--cfg", type=int, default=-1)
    ap.add_argument("--skip-small", action="store_true")
    args = ap.parse_args()
    print(torch.cuda.get_device_name(0), flush=True)
    rdst_amd.set_tuning(args.cfg, 0)
    ok = True
    if not args.skip_small:
        for dtype in (np.uint32, np.uint64, np.int32, np.int64, np.float32, np.float64):
            for n in (2, 3, 100, 8191, 8192, 8193, 100_000, 1_000_003, 16_777_259):
                ok &= check_small(n, dtype, n)
    sizes = [1 << 24, 1 << 28] + ([1_000_000_000, (1 << 30) + 12345] if args.big else [])
    for dtype in (np.uint32, np.uint64, np.float32):
        for n in sizes:
            if dtype is not np.uint32 and n > 1_000_000_000:
                continue
            ok &= time_sort(n, dtype, iters=3 if n > (1 << 28) else 5)
    print("ALL OK" if ok else "FAILED", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
