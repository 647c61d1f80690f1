"""Writes tests/golden/derived_vectors.json.

These vectors are DERIVED from the reference's source text (formulae and tables), not produced
by running the reference (it is a Rust crate; no cargo/rustc exists here).  They are computed
with plain Python integers / numpy only — no oracle, no device code — so that the oracle and
the C ABI can both be checked against something independent.

  * key-map known answers: the f32 / f64 / iN formulae of src/radix_key_impl.rs:87-185 evaluated
    with Python ints, cross-checked against IEEE total order (what the reference's float tests
    compare with: src/radix_sort.rs:97-144).
  * tuner known answers: the range tables of src/tuners/*.rs evaluated by hand-written Python
    conditionals on the points SURVEY.md §8(c) lists plus every table boundary.
"""
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))


def f32_key(bits):  # radix_key_impl.rs:162-173, with 32-bit two's complement done by hand
    s = bits - (1 << 32) if bits & 0x80000000 else bits
    sh = (s >> 31) & 0xFFFFFFFF          # (s >> 31) as u32  (arithmetic shift)
    s ^= sh >> 1                          # ... >> 1, as i32, xor
    return (s ^ -(1 << 31)) & 0xFFFFFFFF  # ^ i32::MIN


def f64_key(bits):  # radix_key_impl.rs:175-185
    s = bits - (1 << 64) if bits & (1 << 63) else bits
    sh = (s >> 63) & 0xFFFFFFFFFFFFFFFF
    s ^= sh >> 1
    return (s ^ -(1 << 63)) & 0xFFFFFFFFFFFFFFFF


def main():
    out = {"_comment": "derived from reference source text, see make_derived_vectors.py", "key_map": [], "tuner": []}
    f32s = [0.0, -0.0, 1.0, -1.0, float("inf"), float("-inf"), 1e-45, -1e-45, 3.4028235e38, -3.4028235e38, 0.5, -0.5]
    for v in f32s:
        bits = struct.unpack("<I", struct.pack("<f", v))[0]
        out["key_map"].append({"type": "float32", "bits": bits, "key": f32_key(bits)})
    for bits in (0x7FC00000, 0xFFC00000, 0x7F800001, 0xFF800001, 0x00000001, 0x80000001):
        out["key_map"].append({"type": "float32", "bits": bits, "key": f32_key(bits)})
    for v in [0.0, -0.0, 1.0, -1.0, float("inf"), float("-inf"), 5e-324, -5e-324, 2.5, -2.5]:
        bits = struct.unpack("<Q", struct.pack("<d", v))[0]
        out["key_map"].append({"type": "float64", "bits": bits, "key": f64_key(bits)})
    for bits in (0x7FF8000000000000, 0xFFF8000000000000):
        out["key_map"].append({"type": "float64", "bits": bits, "key": f64_key(bits)})
    for v in (0, 1, -1, 2**31 - 1, -(2**31), 123456789, -123456789):
        bits = v & 0xFFFFFFFF
        out["key_map"].append({"type": "int32", "bits": bits, "key": bits ^ 0x80000000})  # radix_key_impl.rs:105-112
    for v in (0, 1, -1, 2**63 - 1, -(2**63)):
        bits = v & 0xFFFFFFFFFFFFFFFF
        out["key_map"].append({"type": "int64", "bits": bits, "key": bits ^ (1 << 63)})  # :114-121
    # the eight pairs SURVEY.md §8(c) quotes must come out of the formula above
    quoted = {0x00000000: 0x80000000, 0x80000000: 0x7FFFFFFF, 0x3F800000: 0xBF800000, 0xBF800000: 0x407FFFFF,
              0x7F800000: 0xFF800000, 0xFF800000: 0x007FFFFF, 0x7FC00000: 0xFFC00000, 0xFFC00000: 0x003FFFFF}
    for b, k in quoted.items():
        assert f32_key(b) == k, (hex(b), hex(f32_key(b)), hex(k))

    # tuner: (tuner, threads, level, total_levels, input_len, parent_len, counts spec, expected)
    def uniform(n):
        return {"kind": "uniform", "len": n}

    def skew(n):  # one bin holds 2 * (n // 256) + 1 elements, the rest spread evenly
        return {"kind": "skew", "len": n}

    T = out["tuner"]
    # SURVEY.md §8(c) list (StandardTuner)
    T += [
        ["standard", 8, 3, 4, 10_000_000, None, uniform(10_000_000), "Recombinating"],
        ["standard", 8, 2, 4, 39_062, 10_000_000, uniform(39_062), "Lsb"],
        ["standard", 8, 3, 4, 1_000_000_000, None, uniform(1_000_000_000), "Scanning"],
        ["standard", 8, 2, 4, 3_906_250, 1_000_000_000, uniform(3_906_250), "Recombinating"],
        ["standard", 8, 1, 4, 15_259, 3_906_250, uniform(15_259), "Lsb"],
        ["standard", 8, 3, 4, 100, None, uniform(100), "Comparative"],
        ["standard", 8, 3, 4, 10_000_000, None, skew(10_000_000), "Regions"],
        ["standard", 8, 3, 4, 1_000_000, None, skew(1_000_000), "MtLsb"],
    ]
    # every boundary of standard_tuner.rs:26-61
    for n, e in ((128, "Comparative"), (129, "Lsb"), (150_000, "Lsb"), (150_001, "Ska"), (260_000, "Ska"),
                 (260_001, "Recombinating"), (50_000_000, "Recombinating"), (50_000_001, "Scanning")):
        T.append(["standard", 8, 3, 4, n, None, uniform(n), e])
    for n, e in ((200_000, "Lsb"), (200_001, "Ska"), (800_000, "Ska"), (800_001, "Recombinating"),
                 (50_000_000, "Recombinating"), (50_000_001, "Scanning")):
        T.append(["standard", 8, 2, 4, n, n * 256, uniform(n), e])
    for n, e in ((5_000, "LrLsb"), (200_000, "LrLsb"), (200_001, "Ska"), (350_000, "Ska"), (350_001, "MtLsb"),
                 (4_000_000, "MtLsb"), (4_000_001, "Regions")):
        T.append(["standard", 8, 3, 4, n, None, skew(n), e])
    for n, e in ((200_000, "LrLsb"), (200_001, "Ska"), (800_000, "Ska"), (800_001, "Recombinating"),
                 (5_000_000, "Recombinating"), (5_000_001, "Regions")):
        T.append(["standard", 8, 1, 4, n, n * 300, skew(n), e])
    T.append(["standard", 8, 3, 4, 4_999, None, skew(4_999), "Lsb"])  # skew test needs len >= 5000 (:20)
    # low_memory_tuner.rs:20-41
    for n, e in ((128, "Comparative"), (50_000, "Lsb"), (50_001, "Ska"), (1_000_000, "Ska"), (1_000_001, "Regions")):
        T.append(["low_memory", 8, 3, 4, n, None, uniform(n), e])
    for n, e in ((50_000, "LrLsb"), (50_001, "Ska"), (1_000_000, "Ska"), (1_000_001, "Regions")):
        T.append(["low_memory", 8, 3, 4, n, None, skew(n), e])
    # single_threaded_tuner.rs:22-41
    for lvl, n, c, e in ((3, 800_000, "u", "Lsb"), (3, 800_001, "u", "Ska"), (2, 800_001, "u", "Lsb"),
                         (3, 100_000, "s", "LrLsb"), (3, 100_001, "s", "Ska"), (2, 100_001, "s", "Ska"),
                         (1, 100_001, "s", "LrLsb"), (3, 128, "u", "Comparative")):
        T.append(["single_threaded", 1, lvl, 4, n, None, uniform(n) if c == "u" else skew(n), e])
    with open(os.path.join(HERE, "derived_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["key_map"]), "key-map and", len(T), "tuner vectors")


if __name__ == "__main__":
    main()
