"""Host mirror of rdst's ``tuner`` module (src/tuner.rs:1-40) and its stock tuners
(src/tuners/*.rs).  The decision tables themselves live behind the C ABI
(``rdst_pick_algorithm`` in rdst_amd/csrc/rdst_tuner.cpp); the classes here only carry
the tuner id across, exactly as a Rust shim would."""
import ctypes
import enum
from dataclasses import dataclass
from typing import Optional, Sequence

from . import _lib


class Algorithm(enum.IntEnum):
    """src/tuner.rs:12-22, same order, plus the two device routes."""
    MtOop = 0
    MtLsb = 1
    Scanning = 2
    Recombinating = 3
    Comparative = 4
    LrLsb = 5
    Lsb = 6
    Regions = 7
    Ska = 8
    GpuLsd = 9
    GpuSharded = 10


@dataclass
class TuningParams:
    """src/tuner.rs:2-8."""
    threads: int
    level: int
    total_levels: int
    input_len: int
    parent_len: Optional[int] = None


class Tuner:
    """src/tuner.rs:33-35: ``fn pick_algorithm(&self, p: &TuningParams, counts: &[usize]) -> Algorithm``."""

    def pick_algorithm(self, p: TuningParams, counts: Sequence[int]) -> Algorithm:  # pragma: no cover
        raise NotImplementedError


class _TableTuner(Tuner):
    tuner_id = 0
    gpu_min_len = 0

    def pick_algorithm(self, p: TuningParams, counts: Sequence[int]) -> Algorithm:
        if len(counts) != 256:
            raise ValueError("counts must have 256 entries (src/sorter.rs:67-76)")
        lib = _lib.load()
        c = (ctypes.c_uint64 * 256)(*[int(x) for x in counts])
        cp = _lib.TuningParamsC(p.threads, p.level, p.total_levels, p.input_len,
                                -1 if p.parent_len is None else int(p.parent_len))
        rc = lib.rdst_pick_algorithm(self.tuner_id, ctypes.byref(cp), c, int(self.gpu_min_len))
        if rc < 0:
            raise _lib.RdstHipError(rc, "rdst_pick_algorithm rejected its arguments")
        return Algorithm(rc)


class StandardTuner(_TableTuner):
    """src/tuners/standard_tuner.rs:10-64"""
    tuner_id = 0


class LowMemoryTuner(_TableTuner):
    """src/tuners/low_memory_tuner.rs:13-43"""
    tuner_id = 1


class SingleThreadedTuner(_TableTuner):
    """src/tuners/single_threaded_tuner.rs:13-43"""
    tuner_id = 2


#: measured break-even of the host entry point (allocation + H2D + sort + D2H) against the CPU route
#: (tools/breakeven.py): what a shim that still has the CPU algorithms should pass as
#: ``gpu_min_len``.  This package has no CPU route, so its own default stays 0.
GPU_MIN_LEN_HOST_SLICE = 1 << 16


class GpuTuner(_TableTuner):
    """StandardTuner, except that a whole top-level slice of at least ``gpu_min_len``
    elements is routed to the device (Algorithm.GpuLsd)."""
    tuner_id = 3

    def __init__(self, gpu_min_len: int = 0):
        self.gpu_min_len = int(gpu_min_len)
