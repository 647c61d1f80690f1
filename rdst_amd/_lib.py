"""ctypes binding of the C ABI in include/rdst_hip.h.

The shared library is built in-tree (``rdst_amd/librdst_hip.so``) by
``__graft_entry__.build()`` / ``make -C rdst_amd/csrc``.  There is no CPU fallback: if the
library is missing or a call fails, the caller gets an exception.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RDST_HIP_LIB: another build of the same library (tools/ A/B runs only; tests and bench use the in-tree one)
LIB_PATH = os.environ.get("RDST_HIP_LIB") or os.path.join(_HERE, "librdst_hip.so")

RDST_KEY_UNSIGNED, RDST_KEY_SIGNED, RDST_KEY_FLOAT, RDST_KEY_BYTES_BE = 0, 1, 2, 3
RDST_OK = 0

# every symbol include/rdst_hip.h declares; tests check that the library exports all of them
SYMBOLS = (
    "rdst_hip_sort",
    "rdst_hip_sort_device",
    "rdst_hip_sort_device_lowmem",
    "rdst_hip_partition_device",
    "rdst_regions_plan",
    "rdst_hip_host_timing",
    "rdst_hip_sort_pairs_device",
    "rdst_hip_sort_records",
    "rdst_hip_device_status",
    "rdst_hip_level_counts",
    "rdst_hip_all_level_counts",
    "rdst_hip_scatter_level",
    "rdst_hip_split_top_level_device",
    "rdst_hip_split_top16_device",
    "rdst_pick_algorithm",
    "rdst_hip_workspace_bytes",
    "rdst_hip_release_workspace",
    "rdst_hip_set_tuning",
    "rdst_hip_set_chain_split",
    "rdst_hip_set_fast_rank",
    "rdst_hip_set_small_sort",
    "rdst_hip_set_profiling",
    "rdst_hip_profile_runs",
    "rdst_hip_profile_run",
    "rdst_hip_profile_run_stages",
    "rdst_hip_set_hybrid",
    "rdst_hip_last_route",
    "rdst_hip_debug_raise_device_error",
    "rdst_hip_stream_copy",
    "rdst_hip_stream_read",
    "rdst_hip_stream_fill",
    "rdst_hip_last_error",
    "rdst_hip_abi_version",
)


class TuningParamsC(ctypes.Structure):
    _fields_ = [
        ("threads", ctypes.c_uint64),
        ("level", ctypes.c_uint64),
        ("total_levels", ctypes.c_uint64),
        ("input_len", ctypes.c_uint64),
        ("parent_len", ctypes.c_int64),
    ]


class HipOptsC(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("low_memory", ctypes.c_int32), ("reserved1", ctypes.c_uint64)]


class RdstHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rdst_hip call failed with status {code}: {msg}")
        self.code = code


_lib = None


def load():
    """Load librdst_hip.so once.  Importing torch first makes the loader resolve
    libamdhip64.so.7 to the copy torch already mapped, so device pointers and streams
    handed over from torch belong to the same HIP runtime."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RdstHipError(-100, f"{LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
    try:
        import torch  # noqa: F401  (runtime unification, see docstring)
    except Exception:  # pragma: no cover - torch is optional for host-only use
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    vp, u64, u32, ci = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
    u64p, u8p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint8)
    lib.rdst_hip_sort.argtypes = [vp, u64, u32, ci, u32, ctypes.POINTER(HipOptsC)]
    lib.rdst_hip_host_timing.argtypes = [ctypes.POINTER(ctypes.c_float)] * 3
    lib.rdst_hip_sort_device_lowmem.argtypes = [vp, u64, u32, ci, u32, vp, u64, vp]
    lib.rdst_hip_partition_device.argtypes = [vp, u64, u32, ci, u32, u32, vp, u64, u64p, vp]
    lib.rdst_hip_sort_device.argtypes = [vp, vp, u64, u32, ci, u32, vp]
    lib.rdst_hip_sort_pairs_device.argtypes = [vp, vp, vp, vp, u64, u32, ci, u32, u32, vp]
    lib.rdst_hip_sort_records.argtypes = [vp, u64, u32, u32, u32, ci, ctypes.POINTER(HipOptsC)]
    lib.rdst_hip_device_status.argtypes = [vp]
    lib.rdst_hip_level_counts.argtypes = [vp, u64, u32, ci, u32, u64p, u8p, u8p, u8p, vp]
    lib.rdst_hip_all_level_counts.argtypes = [vp, u64, u32, ci, u32, u64p, vp]
    lib.rdst_hip_scatter_level.argtypes = [vp, vp, u64, u32, ci, u32, u64p, vp]
    lib.rdst_hip_split_top_level_device.argtypes = [vp, vp, u64, u32, ci, vp, vp]
    lib.rdst_hip_split_top16_device.argtypes = [vp, vp, u64, u32, ci, vp, vp]
    lib.rdst_pick_algorithm.argtypes = [ci, ctypes.POINTER(TuningParamsC), u64p, u64]
    lib.rdst_hip_workspace_bytes.argtypes = [u64, u32]
    lib.rdst_hip_workspace_bytes.restype = u64
    lib.rdst_hip_release_workspace.argtypes = []
    lib.rdst_hip_set_tuning.argtypes = [ci, ci]
    lib.rdst_hip_set_chain_split.argtypes = [ci]
    lib.rdst_hip_set_fast_rank.argtypes = [ci]
    lib.rdst_hip_set_small_sort.argtypes = [ci]
    lib.rdst_hip_set_profiling.argtypes = [ci]
    lib.rdst_hip_profile_run.argtypes = [ci, ctypes.POINTER(ctypes.c_float), u32, ctypes.POINTER(u32)]
    lib.rdst_hip_profile_run_stages.argtypes = [ci, ctypes.POINTER(u32), u32, ctypes.POINTER(u32)]
    lib.rdst_hip_set_hybrid.argtypes = [ci, u64]
    lib.rdst_hip_last_route.argtypes = [vp, ctypes.POINTER(u32)]
    lib.rdst_hip_debug_raise_device_error.argtypes = [u32, vp]
    lib.rdst_hip_stream_copy.argtypes = [vp, vp, u64, vp]
    lib.rdst_hip_stream_read.argtypes = [vp, u64, vp]
    lib.rdst_hip_stream_fill.argtypes = [vp, u64, vp]
    lib.rdst_hip_last_error.restype = ctypes.c_char_p
    for name in SYMBOLS:
        if name not in ("rdst_hip_workspace_bytes", "rdst_hip_last_error"):
            getattr(lib, name).restype = ctypes.c_int
    _lib = lib
    return lib


def check(rc):
    if rc != RDST_OK:
        raise RdstHipError(rc, load().rdst_hip_last_error().decode("utf-8", "replace"))
