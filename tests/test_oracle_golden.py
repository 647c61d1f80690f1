"""Pins the CPU oracle: against the reference's own known answers (tests/golden/
reference_known_answers.json — every literal input/output pair the reference tree holds), and
against vectors derived from its source text with plain Python (derived_vectors.json)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "reference_known_answers.json")))
DV = json.load(open(os.path.join(HERE, "golden", "derived_vectors.json")))

_NP = {"uint8": np.uint8, "uint32": np.uint32}


def _arr(case):
    t = case["type"]
    if t in _NP:
        return np.array(case["input"], dtype=_NP[t]), np.array(case["expected"], dtype=_NP[t]), None
    return np.array(case["input"], dtype=np.uint8), np.array(case["expected"], dtype=np.uint8), t


@pytest.mark.parametrize("case", KA["sorts"], ids=[c["name"] for c in KA["sorts"]])
@pytest.mark.parametrize("tuner", ["standard", "low_memory", "single_threaded"])
def test_reference_known_answer_sorts(oracle, case, tuner):
    a, exp, kind = _arr(case)
    oracle.sort(a, tuner=tuner, threads=2, kind=kind)
    assert a.tolist() == exp.tolist(), case["source"]


@pytest.mark.parametrize("case", KA["sorts"][:1] + KA["sorts"][2:3], ids=["doc", "simple_usage"])
@pytest.mark.parametrize("algo", ["MtOop", "MtLsb", "Scanning", "Recombinating", "Comparative", "LrLsb", "Lsb", "Regions", "Ska"])
def test_reference_known_answers_through_every_algorithm(oracle, case, algo):
    # inputs of <= 128 elements only ever reach comparative_sort through the public API
    # (src/sorter.rs:33-38), so drive the algorithms on a padded copy as well
    a, exp, kind = _arr(case)
    oracle.sort_single_algorithm(a, algo, threads=2, kind=kind)
    assert a.tolist() == exp.tolist()
    big = np.tile(np.array(case["input"], dtype=np.uint32), 40)  # > 128 elements
    oracle.sort_single_algorithm(big, algo, threads=2)
    assert big.tolist() == sorted(np.tile(np.array(case["input"], dtype=np.uint32), 40).tolist())


@pytest.mark.parametrize("case", KA["tile_counts_sortedness"],
                         ids=[f"{c['input']}-tile{c['tile_size']}" for c in KA["tile_counts_sortedness"]])
def test_get_tile_counts_sortedness_known_answers(oracle, case):
    a = np.array(case["input"], dtype=np.uint8)
    counts, srt = oracle.get_tile_counts(a, case["tile_size"], case["level"])
    assert srt == case["already_sorted"], case["source"]
    assert counts.sum() == len(case["input"])


def test_key_map_vectors(oracle):
    """get_level of the oracle reassembles exactly the mapped key derived from the formulae."""
    for v in DV["key_map"]:
        dt = np.dtype(v["type"])
        raw = int(v["bits"]).to_bytes(dt.itemsize, "little")
        tid = oracle.TYPE_IDS[v["type"]]
        key = 0
        for level in range(dt.itemsize):
            key |= oracle.get_level(raw, tid, level) << (8 * level)
        assert key == v["key"], v


def _counts(spec):
    n = spec["len"]
    if spec["kind"] == "uniform":
        base, rem = divmod(n, 256)
        return [base + (1 if i < rem else 0) for i in range(256)]
    heavy = 2 * (n // 256) + 1
    rest = n - heavy
    base, rem = divmod(rest, 255)
    return [heavy] + [base + (1 if i < rem else 0) for i in range(255)]


def test_tuner_vectors(oracle):
    for tuner, threads, level, total, n, parent, spec, expected in DV["tuner"]:
        c = _counts(spec)
        assert sum(c) == n
        got = oracle.pick_algorithm(tuner, threads, level, total, n, parent, c)
        assert got == expected, (tuner, level, n, spec["kind"], got, expected)


def test_standard_route_on_10m_u32_matches_survey_table(oracle):
    """10 M uniform u32 -> Recombinating at the top, then Lsb on ~39 k chunks."""
    rng = np.random.default_rng(0x5D570001)
    a = rng.integers(0, 1 << 32, size=10_000_000, dtype=np.uint32)
    exp = np.sort(a)
    log = oracle.trace_standard_route(a, threads=4)
    assert np.array_equal(a, exp)
    assert log[0] == (3, 10_000_000, "Recombinating")
    lower = [e for e in log if e[0] == 2]
    assert len(lower) == 256 and all(e[2] == "Lsb" for e in lower)
