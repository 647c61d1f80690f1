"""rdst_amd — MI355X (gfx950) device route for rdst's radix sort, behind rdst's own surface.

    import rdst_amd
    rdst_amd.radix_sort_unstable(x)                       # numpy array or HIP torch tensor, in place
    rdst_amd.radix_sort_builder(x).with_tuner(t).sort()

The compute is hand-written HIP (rdst_amd/csrc) behind the C ABI of include/rdst_hip.h;
this package is the thin host mirror of src/radix_sort.rs, src/radix_sort_builder.rs and
src/tuner.rs of the reference.
"""
from . import tuner  # noqa: F401
from .radix_sort import (  # noqa: F401
    RadixSortBuilder,
    all_level_counts,
    device_status,
    key_info,
    level_counts,
    radix_sort_builder,
    radix_sort_unstable,
    scatter_level,
    set_tuning,
    set_profiling,
    set_hybrid,
    last_route,
    release_workspace,
    last_profile,
    profile_run,
    profile_runs,
    sort_device_tensor,
    sort_device_tensor_lowmem,
    partition_device,
    sort_host_array,
    sort_host_records,
    sort_pairs_device_tensor,
    sort_records_by_key,
)
from ._lib import RdstHipError  # noqa: F401

__all__ = [
    "radix_sort_unstable", "radix_sort_builder", "RadixSortBuilder", "tuner", "RdstHipError",
    "sort_device_tensor", "sort_device_tensor_lowmem", "partition_device", "sort_host_array", "sort_host_records", "sort_pairs_device_tensor", "sort_records_by_key", "level_counts", "all_level_counts", "scatter_level",
    "device_status", "set_tuning", "set_profiling", "set_hybrid", "last_route", "release_workspace", "last_profile", "key_info",
]
