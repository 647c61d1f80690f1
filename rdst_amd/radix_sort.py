"""Host mirror of rdst's entry points for the device route.

Reference surface (paths in the reference tree):
  * ``RadixSort::radix_sort_unstable`` / ``radix_sort_builder``   src/radix_sort.rs:4-45
  * ``RadixSortBuilder`` and its ``with_*`` methods / ``sort``     src/radix_sort_builder.rs:8-158
  * ``RadixKey`` built-in mappings -> (kind, elem_bytes, levels)   src/radix_key_impl.rs:1-185

Accepted containers: a C-contiguous 1-D ``numpy.ndarray`` (host slice -> ``rdst_hip_sort``)
or a contiguous 1-D ``torch.Tensor`` on a HIP device (device slice ->
``rdst_hip_sort_device``; PyTorch only provides the memory and the stream).  Sorting is in
place, returns ``None`` like the reference, and raises instead of falling back when the
device path is unavailable: this package ships the device route only — the CPU algorithms
remain the reference crate's own.
"""
import ctypes

import numpy as np

from . import _lib
from .tuner import Algorithm, GpuTuner, LowMemoryTuner, SingleThreadedTuner, StandardTuner, Tuner, TuningParams

# dtype name -> (rdst_key_kind, elem_bytes); LEVELS == elem_bytes for every built-in type
_KEY_TABLE = {
    "uint8": (_lib.RDST_KEY_UNSIGNED, 1),
    "uint16": (_lib.RDST_KEY_UNSIGNED, 2),
    "int8": (_lib.RDST_KEY_SIGNED, 1),
    "int16": (_lib.RDST_KEY_SIGNED, 2),
    "uint32": (_lib.RDST_KEY_UNSIGNED, 4),
    "uint64": (_lib.RDST_KEY_UNSIGNED, 8),
    "int32": (_lib.RDST_KEY_SIGNED, 4),
    "int64": (_lib.RDST_KEY_SIGNED, 8),
    "float32": (_lib.RDST_KEY_FLOAT, 4),
    "float64": (_lib.RDST_KEY_FLOAT, 8),
    # no numpy / torch dtype exists for these: pass key="u128" / "i128" with a container of shape
    # (n, 2) whose rows are the little-endian 64-bit limbs [low, high] of one key
    "u128": (_lib.RDST_KEY_UNSIGNED, 16),
    "i128": (_lib.RDST_KEY_SIGNED, 16),
}


def key_info(dtype_name: str):
    """(kind, elem_bytes, levels) of a built-in RadixKey (src/radix_key_impl.rs)."""
    name = str(dtype_name).replace("torch.", "")
    if name not in _KEY_TABLE:
        raise TypeError(f"no device RadixKey mapping for dtype {dtype_name}; supported: {sorted(_KEY_TABLE)}")
    kind, nbytes = _KEY_TABLE[name]
    return kind, nbytes, nbytes


def _is_torch_tensor(x):
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def _stream_handle(tensor):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(tensor.device).cuda_stream)


def _wide(key):
    return key in ("u128", "i128")


def _check_wide_shape(shape, itemsize):
    if len(shape) != 2 or shape[1] * itemsize != 16:
        raise ValueError("a 128-bit key container has shape (n, 2) with 8-byte limbs [low, high]")


def sort_device_tensor(keys, tmp=None, check=True, key=None):
    """``rdst_hip_sort_device`` on a 1-D contiguous HIP tensor (``key="u128"/"i128"``: shape (n, 2)).  ``tmp``: optional scratch
    tensor of the same shape/dtype (allocated when omitted).  With ``check`` the call blocks
    and raises if a kernel reported failure; without it the sort stays asynchronous on the
    tensor's current stream (call :func:`device_status` later)."""
    import torch
    if not keys.is_cuda:
        raise ValueError("sort_device_tensor needs a tensor on a HIP device")
    if _wide(key):
        _check_wide_shape(tuple(keys.shape), keys.element_size())
    elif keys.dim() != 1:
        raise ValueError("keys must be a contiguous 1-D tensor (rdst sorts a slice)")
    if not keys.is_contiguous():
        raise ValueError("keys must be a contiguous 1-D tensor (rdst sorts a slice)")
    kind, nbytes, levels = key_info(key if key else keys.dtype)
    n = keys.numel() * keys.element_size() // nbytes
    if n <= 1:
        return
    if tmp is None:
        tmp = torch.empty_like(keys)
    elif tmp.dtype != keys.dtype or tmp.numel() < keys.numel() or not tmp.is_contiguous() or tmp.device != keys.device:
        raise ValueError("tmp must be a contiguous tensor of the same dtype/device with at least len elements")
    lib = _lib.load()
    with torch.cuda.device(keys.device):
        s = _stream_handle(keys)
        _lib.check(lib.rdst_hip_sort_device(ctypes.c_void_p(keys.data_ptr()), ctypes.c_void_p(tmp.data_ptr()),
                                            n, nbytes, kind, levels, s))
        if check:
            _lib.check(lib.rdst_hip_device_status(s))


def sort_pairs_device_tensor(keys, values, tmp_keys=None, tmp_values=None, check=True):
    """``rdst_hip_sort_pairs_device``: sort the 1-D HIP tensor ``keys`` (4- or 8-byte built-in key type) in place
    and permute ``values`` (4- or 8-byte elements, same length) with it.  Stable: equal keys keep their
    input order (rdst promises none, src/radix_sort.rs:21-45)."""
    import torch
    if not (keys.is_cuda and values.is_cuda) or keys.device != values.device:
        raise ValueError("keys and values must live on the same HIP device")
    if keys.dim() != 1 or values.dim() != 1 or keys.numel() != values.numel():
        raise ValueError("keys and values must be 1-D tensors of the same length")
    if not (keys.is_contiguous() and values.is_contiguous()):
        raise ValueError("keys and values must be contiguous")
    kind, nbytes, levels = key_info(keys.dtype)
    vbytes = values.element_size()
    n = keys.numel()
    if n <= 1:
        return
    tmp_keys = torch.empty_like(keys) if tmp_keys is None else tmp_keys
    tmp_values = torch.empty_like(values) if tmp_values is None else tmp_values
    for t, ref in ((tmp_keys, keys), (tmp_values, values)):
        if t.dtype != ref.dtype or t.numel() < n or not t.is_contiguous() or t.device != ref.device:
            raise ValueError("tmp tensors must match their originals in dtype, device and length")
    lib = _lib.load()
    with torch.cuda.device(keys.device):
        s = _stream_handle(keys)
        _lib.check(lib.rdst_hip_sort_pairs_device(ctypes.c_void_p(keys.data_ptr()), ctypes.c_void_p(values.data_ptr()),
                                                  ctypes.c_void_p(tmp_keys.data_ptr()), ctypes.c_void_p(tmp_values.data_ptr()),
                                                  n, nbytes, kind, levels, vbytes, s))
        if check:
            _lib.check(lib.rdst_hip_device_status(s))


def sort_records_by_key(records, key_field):
    """Device route for a slice of structs whose ``RadixKey`` is one built-in field
    (benches/struct_sort.rs:11-27, examples/impl_radix_key.rs:32-56): ``records`` is a 2-D
    HIP tensor (n, fields), ``key_field`` the column holding the key (the tensor's dtype decides the key kind).  Extracts (key, row index), sorts the pairs on
    the device, gathers the rows; returns the reordered tensor (rows with equal keys keep their order)."""
    import torch
    if records.dim() != 2 or not records.is_cuda:
        raise ValueError("records must be a 2-D HIP tensor (n, fields)")
    n = records.shape[0]
    if n <= 1:
        return records.clone()
    keys = records[:, key_field].contiguous()
    idx = torch.arange(n, dtype=torch.int32 if n < 2**31 else torch.int64, device=records.device)
    sort_pairs_device_tensor(keys, idx)
    return records.index_select(0, idx.long() if idx.dtype != torch.int64 else idx)


def sort_device_tensor_lowmem(keys, scratch=None):
    """``rdst_hip_sort_device_lowmem``: sort a 1-D contiguous HIP tensor IN PLACE with a scratch of only ``scratch.numel()``
    elements (default: len / 64, at least 65 536) instead of a second array — the device twin of the route
    ``with_low_mem_tuner()`` selects in the reference (Regions / Ska: src/tuners/low_memory_tuner.rs:36-41,
    src/sorts/regions_sort.rs:51-286).  Blocking.  Same result as :func:`sort_device_tensor`."""
    import torch
    if not keys.is_cuda or keys.dim() != 1 or not keys.is_contiguous():
        raise ValueError("sort_device_tensor_lowmem needs a contiguous 1-D tensor on a HIP device")
    kind, nbytes, levels = key_info(keys.dtype)
    n = keys.numel()
    if n <= 1:
        return
    if scratch is None:
        scratch = torch.empty(max(65536, -(-n // 64)), dtype=keys.dtype, device=keys.device)
    elif scratch.dtype != keys.dtype or scratch.device != keys.device or not scratch.is_contiguous():
        raise ValueError("scratch must be a contiguous tensor of the same dtype and device")
    lib = _lib.load()
    with torch.cuda.device(keys.device):
        _lib.check(lib.rdst_hip_sort_device_lowmem(ctypes.c_void_p(keys.data_ptr()), n, nbytes, kind, levels,
                                                   ctypes.c_void_p(scratch.data_ptr()), scratch.numel(), _stream_handle(keys)))


def partition_device(keys, level, digit, scratch=None):
    """``rdst_hip_partition_device`` — ``partition_index`` (src/sort_utils.rs:295-331) with the predicate "digit `level` of
    the key == `digit`": those keys first, the others after, in place; returns the split index.  Blocking."""
    import torch
    if not keys.is_cuda or keys.dim() != 1 or not keys.is_contiguous():
        raise ValueError("partition_device needs a contiguous 1-D tensor on a HIP device")
    kind, nbytes, _levels = key_info(keys.dtype)
    n = keys.numel()
    if scratch is None:
        scratch = torch.empty(max(65536, -(-n // 64)), dtype=keys.dtype, device=keys.device)
    split = ctypes.c_uint64(0)
    lib = _lib.load()
    with torch.cuda.device(keys.device):
        _lib.check(lib.rdst_hip_partition_device(ctypes.c_void_p(keys.data_ptr()), n, nbytes, kind, int(level), int(digit),
                                                 ctypes.c_void_p(scratch.data_ptr()), scratch.numel(), ctypes.byref(split),
                                                 _stream_handle(keys)))
    return int(split.value)


def device_status(device=None):
    """Block on the current stream and raise if a kernel reported failure."""
    import torch
    lib = _lib.load()
    with torch.cuda.device(device):
        _lib.check(lib.rdst_hip_device_status(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))


def sort_host_array(arr, device=-1, key=None):
    """``rdst_hip_sort`` on a host numpy array (H2D, device sort, D2H), in place.  ``key="bytes"``: a uint8
    array of shape (n, N), N in 1..16, each row one ``[u8; N]`` key (src/radix_key_impl.rs:78-85: rows
    end up in lexicographic order)."""
    if not isinstance(arr, np.ndarray) or not arr.flags.c_contiguous or not arr.flags.writeable:
        raise ValueError("need a writeable C-contiguous 1-D numpy array (rdst sorts a mutable slice)")
    if key == "bytes":
        if arr.ndim != 2 or arr.dtype != np.uint8 or not 1 <= arr.shape[1] <= 16:
            raise ValueError("[u8; N] keys: a uint8 array of shape (n, N) with N in 1..16")
        if arr.shape[0] <= 1:
            return
        n_bytes = int(arr.shape[1])
        opts = _lib.HipOptsC(int(device), 0, 0)
        _lib.check(_lib.load().rdst_hip_sort(ctypes.c_void_p(arr.ctypes.data), arr.shape[0], n_bytes, _lib.RDST_KEY_BYTES_BE, n_bytes,
                                             ctypes.byref(opts)))
        return
    if _wide(key):
        _check_wide_shape(arr.shape, arr.dtype.itemsize)
    elif arr.ndim != 1:
        raise ValueError("need a writeable C-contiguous 1-D numpy array (rdst sorts a mutable slice)")
    kind, nbytes, levels = key_info(key if key else arr.dtype.name)
    if arr.nbytes // nbytes <= 1:
        return
    lib = _lib.load()
    opts = _lib.HipOptsC(int(device), 0, 0)
    _lib.check(lib.rdst_hip_sort(ctypes.c_void_p(arr.ctypes.data), arr.nbytes // nbytes, nbytes, kind, levels, ctypes.byref(opts)))


def sort_host_records(arr, field, device=-1):
    """``rdst_hip_sort_records`` on a numpy structured array, in place: the rows are ordered by ``field``
    (a 4- or 8-byte integer or float field); rows with equal keys keep their order."""
    if not isinstance(arr, np.ndarray) or arr.dtype.fields is None or arr.ndim != 1:
        raise ValueError("need a 1-D numpy structured array")
    if not arr.flags.c_contiguous or not arr.flags.writeable:
        raise ValueError("need a writeable C-contiguous array (rdst sorts a mutable slice)")
    ftype, offset = arr.dtype.fields[field][:2]
    kind, nbytes, _levels = key_info(ftype.name)
    if arr.shape[0] <= 1:
        return
    lib = _lib.load()
    opts = _lib.HipOptsC(int(device), 0, 0)
    _lib.check(lib.rdst_hip_sort_records(ctypes.c_void_p(arr.ctypes.data), arr.shape[0], arr.dtype.itemsize, int(offset), nbytes, kind,
                                         ctypes.byref(opts)))


class RadixSortBuilder:
    """src/radix_sort_builder.rs:8-158.  ``with_parallel`` and the CPU tuners are accepted for
    source compatibility; they select among the reference's CPU algorithms, which this package
    does not ship, so on the device route they only take part in the top-level
    ``pick_algorithm`` call (a tuner that does not return a ``Gpu*`` algorithm raises)."""

    def __init__(self, data, key=None):
        self._data = data
        self._key = key
        self._multi_threaded = True
        self._tuner: Tuner = GpuTuner(0)

    def with_parallel(self, parallel: bool):
        self._multi_threaded = bool(parallel)
        return self

    def with_low_mem_tuner(self):
        self._tuner = LowMemoryTuner()
        return self

    def with_single_threaded_tuner(self):
        self._tuner = SingleThreadedTuner()
        return self

    def with_tuner(self, tuner: Tuner):
        if not hasattr(tuner, "pick_algorithm"):
            raise TypeError("tuner must implement pick_algorithm(p, counts)")
        self._tuner = tuner
        return self

    def _len_and_levels(self):
        d = self._data
        if self._key == "bytes":  # [u8; N] rows of a (n, N) uint8 array
            return int(d.shape[0]), int(d.shape[1])
        if self._key:
            _, nbytes, levels = key_info(self._key)
            total = d.numel() * d.element_size() if _is_torch_tensor(d) else d.nbytes
            return total // nbytes, levels
        if _is_torch_tensor(d):
            return d.numel(), key_info(d.dtype)[2]
        return d.size, key_info(d.dtype.name)[2]

    def sort(self):
        """Sorts in place and returns None, like the reference — except when the tuner answers
        ``Algorithm.GpuSharded``: the slice is then this rank's shard of a distributed array, the call is
        collective over the default torch.distributed group (one rank per GPU), and the return value is a NEW
        tensor, this rank's contiguous slice of the global order (its length differs from the input's, so it
        cannot be written in place)."""
        n, levels = self._len_and_levels()
        algo = None
        try:
            import torch.distributed as dist
            collective = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        except Exception:  # noqa: BLE001
            collective = False
        # radix_sort_builder.rs:151.  Only a custom tuner can answer GpuSharded (the stock ones and the default device
        # tuner never do): only then may an empty or one-key shard have to take part in a collective sort.
        custom = not isinstance(self._tuner, (GpuTuner, LowMemoryTuner, SingleThreadedTuner, StandardTuner))
        if n <= 1 and not (collective and custom):
            return
        if isinstance(self._tuner, LowMemoryTuner) and not self._key:
            # with_low_mem_tuner(): the reference trades speed for memory (Ska / Regions instead of the out-of-place
            # sorts, src/tuners/low_memory_tuner.rs:13-43); so does the device: in place, scratch of len / 64 elements
            import torch
            if _is_torch_tensor(self._data):
                sort_device_tensor_lowmem(self._data)
            else:
                dev = torch.from_numpy(self._data.view(_same_width_int(self._data.dtype))).cuda().view(getattr(torch, self._data.dtype.name))
                sort_device_tensor_lowmem(dev)
                self._data.view(_same_width_int(self._data.dtype))[:] = dev.view(getattr(torch, np.dtype(_same_width_int(self._data.dtype)).name)).cpu().numpy()
            return
        if not isinstance(self._tuner, GpuTuner):
            if self._key:
                raise NotImplementedError("custom tuners are wired for the dtype-described key types only")
            # top-level pick_algorithm, as Sorter::handle_chunk does (src/sorter.rs:67-76)
            counts = top_level_counts(self._data)
            algo = self._tuner.pick_algorithm(
                TuningParams(threads=1, level=levels - 1, total_levels=levels, input_len=n, parent_len=None), counts)
        if collective and custom and _is_torch_tensor(self._data):
            # GpuSharded is collective: every rank must take it or none (a rank that stays out leaves the others waiting in
            # the all-gather for ever).  One small all-reduce settles it; disagreement raises on EVERY rank.
            import torch
            import torch.distributed as dist
            dev = self._data.device if dist.get_backend() == "nccl" else "cpu"
            votes = torch.tensor([int(algo == Algorithm.GpuSharded), int(algo != Algorithm.GpuSharded)], dtype=torch.int32, device=dev)
            dist.all_reduce(votes)
            yes, no = (int(v) for v in votes.cpu())
            if yes and no:
                raise RuntimeError(f"the tuner answered Algorithm.GpuSharded on {yes} rank(s) and something else on {no}: a tuner that "
                                   "can route to the sharded sort must answer identically on all ranks (decide on world-level "
                                   "quantities, not on the local shard)")
        if algo is not None and algo not in (Algorithm.GpuLsd, Algorithm.GpuSharded):
            raise NotImplementedError(
                f"tuner picked {Algorithm(algo).name}: the CPU algorithms stay in the reference crate; "
                "this package implements the device routes (Algorithm.GpuLsd, Algorithm.GpuSharded) only")
        if algo == Algorithm.GpuSharded:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("Algorithm.GpuSharded is collective: it needs an initialised torch.distributed "
                                   "process group with one rank per GPU (rdst_amd/sharded.py)")
            if not _is_torch_tensor(self._data) or self._key:
                raise NotImplementedError("Algorithm.GpuSharded sorts a HIP tensor of a dtype-described key type (this rank's shard)")
            from .sharded import sharded_sort
            out = sharded_sort(self._data)
            if out.is_cuda:
                device_status(out.device)
            return out
        if _is_torch_tensor(self._data):
            if self._key == "bytes":
                raise NotImplementedError("[u8; N] keys go through the host entry point: pass a numpy array")
            sort_device_tensor(self._data, key=self._key)
        else:
            sort_host_array(self._data, key=self._key)


def top_level_counts(data):
    """256-bin histogram of the most significant level (what handle_chunk hands the tuner)."""
    import torch
    if _is_torch_tensor(data):
        dev = data
    else:
        dev = torch.from_numpy(data.view(_same_width_int(data.dtype))).cuda()
        dev = dev.view(getattr(torch, data.dtype.name))
    counts, _, _, _ = level_counts(dev, key_info(dev.dtype)[2] - 1)
    return counts


def _same_width_int(dt):
    return {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}[np.dtype(dt).itemsize]


def level_counts(keys, level):
    """Parity hook: (counts[256], already_sorted, first_digit, last_digit) of one level
    (get_counts_with_ends, src/sort_utils.rs:109-180) over a HIP tensor."""
    kind, nbytes, _ = key_info(keys.dtype)
    lib = _lib.load()
    counts = (ctypes.c_uint64 * 256)()
    srt, first, last = ctypes.c_uint8(1), ctypes.c_uint8(0), ctypes.c_uint8(0)
    import torch
    with torch.cuda.device(keys.device):
        _lib.check(lib.rdst_hip_level_counts(ctypes.c_void_p(keys.data_ptr()), keys.numel(), nbytes, kind, level,
                                             counts, ctypes.byref(srt), ctypes.byref(first), ctypes.byref(last),
                                             _stream_handle(keys)))
    return list(counts), bool(srt.value), first.value, last.value


def all_level_counts(keys):
    """Parity hook for the fused histogram kernel: numpy uint64 array [levels, 256]."""
    kind, nbytes, levels = key_info(keys.dtype)
    lib = _lib.load()
    out = np.zeros((levels, 256), dtype=np.uint64)
    import torch
    with torch.cuda.device(keys.device):
        _lib.check(lib.rdst_hip_all_level_counts(ctypes.c_void_p(keys.data_ptr()), keys.numel(), nbytes, kind, levels,
                                                 out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), _stream_handle(keys)))
    return out


def scatter_level(src, level, dst=None):
    """Parity hook: one stable counting-sort pass on digit ``level`` (out_of_place_sort,
    src/sorts/out_of_place_sort.rs:52-108).  Returns (dst tensor, counts[256])."""
    import torch
    kind, nbytes, _ = key_info(src.dtype)
    lib = _lib.load()
    if dst is None:
        dst = torch.empty_like(src)
    counts = np.zeros(256, dtype=np.uint64)
    with torch.cuda.device(src.device):
        _lib.check(lib.rdst_hip_scatter_level(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()),
                                              src.numel(), nbytes, kind, level,
                                              counts.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), _stream_handle(src)))
    return dst, counts


def radix_sort_builder(data, key=None) -> RadixSortBuilder:
    """``RadixSort::radix_sort_builder`` (src/radix_sort.rs:29-31 / :42-44).  ``key``: "u128" / "i128"
    for (n, 2) limb containers, "bytes" for a (n, N) uint8 array of ``[u8; N]`` keys; otherwise the key
    type is the container's dtype."""
    if key == "bytes":
        n_levels = int(data.shape[1]) if getattr(data, "ndim", 0) == 2 else 0
    else:
        n_levels = key_info(key if key else (data.dtype if _is_torch_tensor(data) else data.dtype.name))[2]
    assert n_levels != 0, "RadixKey must have at least 1 level"  # radix_sort_builder.rs:22
    return RadixSortBuilder(data, key)


def radix_sort_unstable(data, key=None) -> None:
    """``RadixSort::radix_sort_unstable`` (src/radix_sort.rs:25-27 / :38-40)."""
    radix_sort_builder(data, key).sort()


def set_tuning(pass_config=-1, hist_blocks_per_cu=0, chain_split=True, fast_rank=True, small_sort=True):
    _lib.check(_lib.load().rdst_hip_set_tuning(int(pass_config), int(hist_blocks_per_cu)))
    _lib.check(_lib.load().rdst_hip_set_chain_split(int(bool(chain_split))))
    _lib.check(_lib.load().rdst_hip_set_fast_rank(2 if fast_rank == 2 else int(bool(fast_rank))))
    _lib.check(_lib.load().rdst_hip_set_small_sort(int(bool(small_sort))))


def set_profiling(enabled: bool):
    """Record HIP events between the kernels of every following sort (rdst_hip_set_profiling)."""
    _lib.check(_lib.load().rdst_hip_set_profiling(int(bool(enabled))))


def profile_runs() -> int:
    """Number of pipelines recorded since profiling was enabled (current device)."""
    return int(_lib.load().rdst_hip_profile_runs())


STAGE_NAMES = {1: "clear", 2: "histogram", 3: "scan", 4: "pass", 5: "copy_back", 6: "histogram16", 7: "route", 8: "local_sort", 10: "msd_pass_a", 11: "msd_pass_b", 12: "sample"}


def profile_run(run: int, levels: int):
    """Stage times (ms) of recorded run `run` (negative: from the most recent): dict with 'clear',
    'histogram', 'scan', 'passes' (one per level the call covered; a level the plan skipped shows its
    early-exit time), 'copy_back', and on calls that tried the hybrid route 'histogram16', 'route',
    'local_sort'; 'stages' lists (name, level or None, ms) in launch order."""
    lib = _lib.load()
    buf = (ctypes.c_float * 64)()
    kinds = (ctypes.c_uint32 * 64)()
    n, nk = ctypes.c_uint32(0), ctypes.c_uint32(0)
    _lib.check(lib.rdst_hip_profile_run(int(run), buf, 64, ctypes.byref(n)))
    _lib.check(lib.rdst_hip_profile_run_stages(int(run), kinds, 64, ctypes.byref(nk)))
    if n.value == 0 or n.value != nk.value:
        return None
    out = {"clear": 0.0, "histogram": 0.0, "scan": 0.0, "passes": [], "copy_back": 0.0, "stages": []}
    for i in range(n.value):
        code, level = kinds[i] & 0xFF, (kinds[i] >> 8) & 0xFF
        name = STAGE_NAMES.get(code, f"stage{code}")
        ms = float(buf[i])
        out["stages"].append((name, level if name == "pass" else None, ms))
        if name == "pass":
            out["passes"].append(ms)
        else:
            out[name] = out.get(name, 0.0) + ms
    if len(out["passes"]) < levels:
        return None
    return out


def set_hybrid(enabled=True, min_len=0):
    """Route choice knob (rdst_hip_set_hybrid): consider the byte-saving routes for sorts of at least `min_len` keys (0 = the
    built-in threshold).  True / 1: the default — 4- and 8-byte keys try the atomic route.  False / 0: LSD only.  A/B and
    test modes: 7 the K1h hybrid route for every width; 2 the same with the generic ranked K4; 3 counting K4 fed with whole
    keys (no 16-bit hand-off); 5 no presample; 6 8-byte keys with the one-block-per-CU K4; 8 the atomic route for 4-byte
    keys only (8-byte keys on the hybrid route); 9 the hybrid route without the expanding K4 (buckets up to one tile only);
    10 the default without the hybrid route as the atomic route's first fallback; 11 the default without the giant kernels
    (4-byte keys: a bucket of 65 536 keys and more sends the sort down the LSD route); 12 the default without the exact form
    of the MSD passes; 14 the default without the sample's prediction of the LSD route; 15 the default with the second form
    of the 8-byte K4 (local_wide2_sort_kernel); 16 the default without the split of 8-byte slices beyond the atomic route's
    window (run_split_sort); 17 the default with that split at every length, in eight parts (tests)."""
    _lib.check(_lib.load().rdst_hip_set_hybrid(int(enabled) if enabled in (2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15, 16, 17) else int(bool(enabled)), int(min_len)))


def release_workspace(device=None) -> None:
    """Give the library-owned device workspace back (rdst_hip_release_workspace): the byte-saving routes keep 1.7 x the
    slice between calls.  Blocking; the next sort allocates again."""
    import torch
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        _lib.check(_lib.load().rdst_hip_release_workspace())


def last_route(device=None) -> str:
    """'lsd', 'hybrid' or 'atomic': the route the most recent sort on the current stream's device took."""
    import torch
    lib = _lib.load()
    r = ctypes.c_uint32(0)
    with torch.cuda.device(device):
        _lib.check(lib.rdst_hip_last_route(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.byref(r)))
    return {1: "hybrid", 2: "atomic"}.get(r.value, "lsd")


def last_profile(levels: int):
    return profile_run(-1, levels) if profile_runs() else None
