"""ctypes front end of the CPU oracle (oracle/rdst_oracle.c).

TEST INFRASTRUCTURE.  Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never from rdst_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librdst_oracle.so")

TYPE_IDS = {
    "uint8": 0, "uint16": 1, "uint32": 2, "uint64": 3, "u128": 4,
    "int8": 5, "int16": 6, "int32": 7, "int64": 8, "i128": 9,
    "float32": 10, "float64": 11, "b3": 12, "b4": 13, "pk_even": 14, "pk_odd": 15,
}
LEVELS = {0: 1, 1: 2, 2: 4, 3: 8, 4: 16, 5: 1, 6: 2, 7: 4, 8: 8, 9: 16, 10: 4, 11: 8, 12: 3, 13: 4, 14: 2, 15: 2}
ELEM_BYTES = {0: 1, 1: 2, 2: 4, 3: 8, 4: 16, 5: 1, 6: 2, 7: 4, 8: 8, 9: 16, 10: 4, 11: 8, 12: 3, 13: 4, 14: 4, 15: 4}
TUNERS = {"standard": 0, "low_memory": 1, "single_threaded": 2}
# src/tuner.rs:12-22
ALGORITHMS = ("MtOop", "MtLsb", "Scanning", "Recombinating", "Comparative", "LrLsb", "Lsb", "Regions", "Ska")


class TuningParams(ctypes.Structure):
    _fields_ = [("threads", ctypes.c_size_t), ("level", ctypes.c_size_t), ("total_levels", ctypes.c_size_t),
                ("input_len", ctypes.c_size_t), ("parent_len", ctypes.c_int64)]


PICK_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(TuningParams), ctypes.POINTER(ctypes.c_size_t))
TRACE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int)

_lib = None


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("rdst_oracle.c", "rdst_oracle_impl.h", "rdst_oracle.h", "Makefile")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.run(["make", "-C", _HERE, "-s", "-B", "librdst_oracle.so"], check=True)
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(LIB_PATH)
        sz, vp, ci = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int
        szp, u8p = ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_uint8)
        lib.rdst_oracle_get_level.argtypes = [vp, ci, sz]
        lib.rdst_oracle_pick_algorithm.argtypes = [ci, ctypes.POINTER(TuningParams), szp]
        lib.rdst_oracle_sort.argtypes = [vp, sz, ci, ci, ci, ci]
        lib.rdst_oracle_sort_with_tuner.argtypes = [vp, sz, ci, PICK_FN, vp, ci, ci, TRACE_FN]
        lib.rdst_oracle_sort_single_algorithm.argtypes = [vp, sz, ci, ci, ci]
        lib.rdst_oracle_get_counts_with_ends.argtypes = [vp, sz, ci, sz, szp, u8p, u8p, u8p]
        lib.rdst_oracle_par_get_counts_with_ends.argtypes = [vp, sz, ci, sz, ci, szp, u8p, u8p, u8p]
        lib.rdst_oracle_get_tile_counts.argtypes = [vp, sz, ci, sz, sz, ci, szp, sz, u8p]
        lib.rdst_oracle_get_tile_counts.restype = ctypes.c_long
        lib.rdst_oracle_out_of_place_sort.argtypes = [vp, vp, sz, ci, sz, ci, szp, szp]
        lib.rdst_oracle_lsb_sort_adapter.argtypes = [vp, sz, ci, ci, sz, sz]
        lib.rdst_oracle_mt_lsb_sort.argtypes = [vp, vp, sz, ci, sz, sz, ci]
        _lib = lib
    return _lib


def type_id(arr, kind=None):
    """numpy array -> oracle type id.  kind: 'u128' / 'i128' (array of shape (n, 2) uint64,
    little-endian limbs) or 'b3' (shape (n, 3) uint8) for the types numpy has no dtype for."""
    if kind is not None:
        return TYPE_IDS[kind]
    return TYPE_IDS[arr.dtype.name]


def _n(arr, tid):
    nbytes = ELEM_BYTES[tid]
    assert arr.flags.c_contiguous and arr.nbytes % nbytes == 0
    return arr.nbytes // nbytes


def _ptr(arr):
    return ctypes.c_void_p(arr.ctypes.data)


def num_threads(threads=None):
    return int(threads) if threads else (os.cpu_count() or 1)


def sort(arr, tuner="standard", multi_threaded=True, threads=None, kind=None):
    """radix_sort_builder().with_*().sort() in place (src/radix_sort_builder.rs:19-157)."""
    tid = type_id(arr, kind)
    rc = load().rdst_oracle_sort(_ptr(arr), _n(arr, tid), tid, TUNERS[tuner], int(multi_threaded), num_threads(threads))
    assert rc == 0
    return arr


def sort_single_algorithm(arr, algorithm, threads=None, kind=None):
    """sort_single_algorithm (src/test_utils.rs:264-278): full public API with SingleAlgoTuner."""
    tid = type_id(arr, kind)
    algo = ALGORITHMS.index(algorithm) if isinstance(algorithm, str) else int(algorithm)
    rc = load().rdst_oracle_sort_single_algorithm(_ptr(arr), _n(arr, tid), tid, algo, num_threads(threads))
    assert rc == 0
    return arr


def sort_with_tuner(arr, pick, multi_threaded=True, threads=None, trace=None, kind=None):
    """with_tuner(&custom).sort(): pick(params_dict, counts_list) -> algorithm name or ordinal."""
    tid = type_id(arr, kind)

    def _pick(_ctx, p, counts):
        pp = p.contents
        params = dict(threads=pp.threads, level=pp.level, total_levels=pp.total_levels, input_len=pp.input_len,
                      parent_len=None if pp.parent_len < 0 else pp.parent_len)
        r = pick(params, [counts[i] for i in range(256)])
        return ALGORITHMS.index(r) if isinstance(r, str) else int(r)

    def _trace(_ctx, level, length, algo):
        if trace:
            trace(level, length, ALGORITHMS[algo])

    cb, tb = PICK_FN(_pick), TRACE_FN(_trace)
    rc = load().rdst_oracle_sort_with_tuner(_ptr(arr), _n(arr, tid), tid, cb, None, int(multi_threaded),
                                            num_threads(threads), tb)
    assert rc == 0
    return arr


def trace_standard_route(arr, threads=None, kind=None):
    """Sort with the StandardTuner tables while recording (level, len, algorithm) per chunk."""
    log = []

    def pick(params, counts):
        return pick_algorithm("standard", counts=counts, **params)

    sort_with_tuner(arr, pick, True, threads, lambda lv, ln, al: log.append((lv, ln, al)), kind)
    return log


def pick_algorithm(tuner, threads, level, total_levels, input_len, parent_len, counts):
    p = TuningParams(threads, level, total_levels, input_len, -1 if parent_len is None else parent_len)
    c = (ctypes.c_size_t * 256)(*[int(x) for x in counts])
    r = load().rdst_oracle_pick_algorithm(TUNERS[tuner], ctypes.byref(p), c)
    assert r >= 0
    return ALGORITHMS[r]


def get_level(value_bytes, tid, level):
    buf = (ctypes.c_uint8 * len(value_bytes)).from_buffer_copy(bytes(value_bytes))
    r = load().rdst_oracle_get_level(ctypes.cast(buf, ctypes.c_void_p), tid, level)
    assert r >= 0
    return r


def get_counts_with_ends(arr, level, kind=None, threads=None):
    """(counts[256], already_sorted, first, last) — src/sort_utils.rs:109-180; with threads:
    par_get_counts_with_ends (:35-106)."""
    tid = type_id(arr, kind)
    counts = (ctypes.c_size_t * 256)()
    s, f, l = ctypes.c_uint8(), ctypes.c_uint8(), ctypes.c_uint8()
    if threads:
        rc = load().rdst_oracle_par_get_counts_with_ends(_ptr(arr), _n(arr, tid), tid, level, int(threads), counts,
                                                         ctypes.byref(s), ctypes.byref(f), ctypes.byref(l))
    else:
        rc = load().rdst_oracle_get_counts_with_ends(_ptr(arr), _n(arr, tid), tid, level, counts, ctypes.byref(s),
                                                     ctypes.byref(f), ctypes.byref(l))
    assert rc == 0
    return np.array(list(counts), dtype=np.uint64), bool(s.value), f.value, l.value


def get_tile_counts(arr, tile_size, level, threads=1, kind=None):
    """(tile_counts[tiles, 256], already_sorted) — src/sort_utils.rs:193-244."""
    tid = type_id(arr, kind)
    n = _n(arr, tid)
    max_tiles = max(1, -(-n // tile_size))
    out = (ctypes.c_size_t * (max_tiles * 256))()
    s = ctypes.c_uint8()
    tiles = load().rdst_oracle_get_tile_counts(_ptr(arr), n, tid, tile_size, level, int(threads), out, max_tiles, ctypes.byref(s))
    assert tiles >= 0
    return np.array(list(out), dtype=np.uint64).reshape(max_tiles, 256)[:tiles], bool(s.value)


VARIANTS = {"plain": 0, "with_counts": 1, "lr": 2, "lr_with_counts": 3}


def out_of_place_sort(src, level, variant="plain", kind=None):
    """One stable counting-sort pass (src/sorts/out_of_place_sort.rs:52-424).  Returns
    (dst, next_counts or None)."""
    tid = type_id(src, kind)
    counts, _, _, _ = get_counts_with_ends(src, level, kind)
    c = (ctypes.c_size_t * 256)(*[int(x) for x in counts])
    nc = (ctypes.c_size_t * 256)()
    dst = np.empty_like(src)
    rc = load().rdst_oracle_out_of_place_sort(_ptr(src), _ptr(dst), _n(src, tid), tid, level, VARIANTS[variant], c, nc)
    assert rc == 0
    return dst, (np.array(list(nc), dtype=np.uint64) if VARIANTS[variant] & 1 else None)


def lsb_sort_adapter(arr, start_level, end_level, lr=False, kind=None):
    tid = type_id(arr, kind)
    rc = load().rdst_oracle_lsb_sort_adapter(_ptr(arr), _n(arr, tid), tid, int(lr), start_level, end_level)
    assert rc == 0
    return arr


def mt_lsb_sort(src, tile_size, level, threads=2, kind=None):
    tid = type_id(src, kind)
    dst = np.empty_like(src)
    rc = load().rdst_oracle_mt_lsb_sort(_ptr(src), _ptr(dst), _n(src, tid), tid, tile_size, level, int(threads))
    assert rc == 0
    return dst
