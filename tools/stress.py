"""Development aid: randomized stress of the device sort against torch.sort (sizes, types, distributions,
tuning modes).  Prints one line per failure and a summary."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rdst_amd
from rdst_amd import radix_sort as rs

torch.manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
g = torch.Generator(device="cuda"); g.manual_seed(torch.initial_seed())
types = [("uint32", torch.int32), ("int32", torch.int32), ("float32", torch.int32), ("uint64", torch.int64), ("int64", torch.int64),
         ("float64", torch.int64), ("uint16", torch.int16), ("uint8", torch.int8)]


def mapped(x, name):  # signed-comparable image of rdst's key order
    mn = torch.iinfo(x.dtype).min
    if name.startswith("uint"): return x ^ mn
    if name.startswith("int"): return x
    return torch.where(x < 0, ~x ^ mn, x)


def gen(n, it, kind):
    info = torch.iinfo(it)
    r = torch.randint(info.min, info.max, (n,), dtype=it, device="cuda", generator=g)
    bits = info.bits
    if kind == 1: r &= (1 << (bits // 2)) - 1                      # upper half zero
    elif kind == 2: r = (r >> (bits // 2)) << (bits // 2)           # lower half zero
    elif kind == 3: r = torch.cat([r[: n // 2] & 0xFFFF, (r[n // 2:] >> 16) << 16]) if bits >= 32 else r
    elif kind == 4: r &= ~0xE0                                      # low byte in one digit group
    elif kind == 5: r = r & ~0xFF00 | 0x4200 if bits >= 16 else r   # level 1 constant
    elif kind == 6: r = r.sort().values                             # sorted as integers
    elif kind == 7: r &= 0x0F0F0F0F if bits == 32 else r            # 16 values per digit
    elif kind == 8 and bits >= 32:                                    # a few 16-bit prefixes hold everything (giant buckets)
        k = int(torch.randint(1, 200, (1,)))
        pref = torch.randint(0, 1 << 16, (k,), dtype=torch.int64, device="cuda", generator=g)
        pick = pref[torch.randint(0, k, (n,), device="cuda", generator=g)]
        low = r.to(torch.int64) & ((1 << (bits - 16)) - 1)
        r = ((pick << (bits - 16)) | low).to(it)
    elif kind == 9 and bits == 32:                                    # a float column (sorted as whatever the type is)
        r = (torch.randn(n, dtype=torch.float32, device="cuda", generator=g) * float(10 ** (torch.rand(1) * 6 - 3))).view(torch.int32)
    elif kind == 9 and bits == 64:
        r = (torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * float(10 ** (torch.rand(1) * 6 - 3))).view(torch.int64)
    elif kind == 10: r = torch.full_like(r, int(r[0]))               # one value
    elif kind == 11: r = torch.where(r > 0, r[:1], r[-1:])            # two values
    elif kind == 12 and bits >= 32:                                   # dense ids, shuffled
        r = torch.randperm(n, device="cuda", generator=g).to(it) + int(torch.randint(0, 1 << 20, (1,)))
    elif kind == 13 and bits >= 32:                                   # bimodal (gen_inputs with shift = half the width)
        h = bits // 2
        r = torch.cat([(r[: n // 2] >> h) & ((1 << h) - 1), r[n // 2:] << h])
    return r.contiguous()


t0, runs, fails, total_keys, big = time.time(), 0, 0, 0, 0
last_note = t0
while time.time() - t0 < budget:
    if time.time() - last_note > 30:   # (a long run must show signs of life)
        last_note = time.time()
        print(f"  ... {runs} sorts, {total_keys:.3e} keys, {fails} failures after {last_note - t0:.0f} s", flush=True)
    name, it = types[int(torch.randint(0, len(types), (1,)))]
    e = float(sys.argv[3]) + float(torch.rand(1)) * float(sys.argv[4]) if len(sys.argv) > 4 else 3.0 + float(torch.rand(1)) * 4.6
    n = max(1, int(10 ** e))
    if torch.iinfo(it).bits == 64: n = min(n, 12_000_000 if not os.environ.get("RDST_STRESS_BIG64") else 450_000_000)
    if os.environ.get("RDST_STRESS_TYPES") and name not in os.environ["RDST_STRESS_TYPES"].split(","): continue
    kind = int(torch.randint(0, 14, (1,)))
    # route knobs: the default, the routes considered at every length, and the A/B modes
    mode = [1, 1, 1, 7, 8, 10, 11, 3, 9, 14, 15, 12, 16, 17, 17][int(torch.randint(0, 15, (1,)))]
    rdst_amd.set_hybrid(mode, 1 if int(torch.randint(0, 3, (1,))) else 0)
    total_keys += n; big += n > 1_000_000
    split, fast = bool(torch.randint(0, 2, (1,))), int(torch.randint(0, 3, (1,)))
    rs.set_tuning(chain_split=split, fast_rank=fast)
    src = gen(n, it, kind)
    if torch.iinfo(it).bits >= 32 and int(torch.randint(0, 4, (1,))) == 0:  # a key-value sort now and then
        vt = torch.int32 if int(torch.randint(0, 2, (1,))) else torch.int64
        keys = src.clone(); vals = torch.arange(n, dtype=vt, device="cuda")
        rs.sort_pairs_device_tensor(keys.view(getattr(torch, name)), vals)
        m = mapped(keys, name)
        okp = bool((m[1:] >= m[:-1]).all()) and bool((src[vals.long()] == keys).all())
        runs += 1
        if not okp:
            fails += 1
            print(f"FAIL pairs {name} {vt} n={n} kind={kind}", flush=True)
        del keys, vals, m
        continue
    keys = src.clone()
    view = keys.view(getattr(torch, name))
    rs.sort_device_tensor(view)
    m = mapped(keys, name)
    exp = mapped(src, name).sort().values
    ok = bool((m == exp).all())
    runs += 1
    if not ok:
        fails += 1
        print(f"FAIL {name} n={n} kind={kind} split={split} fast={fast} mode={mode} route={rdst_amd.last_route()}", flush=True)
    del src, keys, m, exp
rs.set_tuning()
rdst_amd.set_hybrid(True)
print(f"stress: {runs} sorts ({big} above 10^6 keys, {total_keys:.3e} keys in all), {fails} failures", flush=True)
