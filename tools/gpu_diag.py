"""Development aid: localise a device-path failure (histogram vs scatter pass vs full sort)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import rdst_amd


def to_dev(a):
    return torch.from_numpy(a.view(np.int32)).cuda().view(torch.uint32)


def main():
    rng = np.random.default_rng(7)
    for n in (1_000_003, 16_777_259):
        a = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
        for chains in (0, 1, 2, 3):
            rdst_amd.set_tuning(chains, 0)
            t = to_dev(a)
            msg = [f"n={n} cfg={chains}:"]
            try:
                h = rdst_amd.all_level_counts(t)
                exp = np.stack([np.bincount((a >> (8 * l)) & 0xFF, minlength=256) for l in range(4)]).astype(np.uint64)
                msg.append("hist=" + ("ok" if np.array_equal(h, exp) else "BAD"))
                for level in (0, 3):
                    d, c = rdst_amd.scatter_level(t, level)
                    got = d.view(torch.int32).cpu().numpy().view(np.uint32)
                    order = np.argsort((a >> (8 * level)) & 0xFF, kind="stable")
                    ok = np.array_equal(got, a[order])
                    msg.append(f"scatter{level}=" + ("ok" if ok else "BAD"))
                    if not ok:
                        bad = np.nonzero(got != a[order])[0]
                        msg.append(f"(first bad {bad[0]}, nbad {bad.size}, )")
                t2 = to_dev(a)
                rdst_amd.radix_sort_unstable(t2)
                got = t2.view(torch.int32).cpu().numpy().view(np.uint32)
                msg.append("sort=" + ("ok" if np.array_equal(got, np.sort(a)) else "BAD"))
            except Exception as e:  # noqa: BLE001
                msg.append(f"EXC {e}")
            print(" ".join(msg), flush=True)


if __name__ == "__main__":
    main()
