"""GPU-time profiler module of the straggler path: device timestamps instead of CUPTI activity records.

Presents the interface of the reference's pybind11 module ``nvrx_cupti_module``
(cupti_src/cupti_module_py.cpp:33-55) -- ``CuptiProfiler(bufferSize, numBuffers,
statsMaxLenPerKernel)`` with ``initialize / shutdown / start / stop / get_stats / reset`` and
``KernelStats{min,max,median,avg,stddev,num_calls}`` -- so ``CuptiManager`` and its tests carry over.
The alias module ``nvrx_cupti_module`` re-exports it under the reference's import name.

What changes on MI355X: there is no CUPTI, and tracing every kernel through rocprofiler would put a
tool thread and a string hash on every launch.  Instead ``start(key)`` / ``stop()`` bracket the
profiled region on the caller's current HIP stream: by default with two one-thread kernels that read
the device's constant-rate wall clock in stream order, the second of which writes the elapsed GPU
time (microseconds, f32, as CuptiProfiler.cpp:191) straight into the device ring keyed by ``key`` --
nothing is read back and the host never waits (``NVRX_GPU_TIMING=event`` selects the older hipEvent
pair, whose elapsed time is harvested by the host when the events complete).  Statistics over a ring follow the reference's native conventions (mean-of-middles median,
population stddev; CuptiProfiler.cpp:44-74) and are computed by the HIP statistics kernel.
Granularity is therefore one timing row per profiled REGION rather than per kernel name
(documented deviation; per-kernel names are SURVEY section 8(f) row 1).

What the region granularity CANNOT do: the reference drops ``ncclDev*`` kernels from the GPU score
(reporting.py:330-336) because a collective's duration is the time spent waiting for the slowest peer.  A region
timed as a whole includes whatever collectives (and stream idle gaps) fall inside it, so in a synchronous
multi-rank job the regions of all ranks equalise, ``gpu_relative_perf_scores`` sits near 1.0 for everybody and a
slow GPU can go unflagged, while a host-bound rank looks like a slow GPU.  Keep collectives OUT of
``profile_cuda=True`` sections in this mode, or select ``NVRX_GPU_TIMING=kernels`` (rocprofiler-sdk kernel
records, per-kernel keys, ``ncclDev*`` excluded as in the reference).  A multi-rank job that opens a GPU-timed
region in region mode gets this as a one-time warning.
"""
from __future__ import annotations

import logging
import math
import os
import weakref
from typing import Dict, Optional

from . import _native
from . import backend as _backend_mod

DEFAULT_KEY = "gpu_region"
_log = logging.getLogger(__name__)
_warned_region_mode = False


def _warn_region_mode_once() -> None:
    """One-time notice for multi-rank jobs: see the module docstring (collectives inside a region-timed section)."""
    global _warned_region_mode
    if _warned_region_mode:
        return
    _warned_region_mode = True
    try:
        import torch.distributed as dist

        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    except Exception:  # noqa: BLE001
        multi = False
    if multi:
        _log.warning(
            "nvrx straggler: GPU time is measured per profiled REGION (NVRX_GPU_TIMING=%s). Collectives inside a "
            "profile_cuda=True section add peer-wait time to it and cannot be excluded the way the reference excludes "
            "ncclDev* kernels; relative GPU scores may flatten. Keep collectives outside GPU-timed sections or set "
            "NVRX_GPU_TIMING=kernels (the default of a multi-rank job when nvrx_straggler is imported before the HIP "
            "runtime starts).", os.environ.get("NVRX_GPU_TIMING", "stamp"))


class KernelStats:
    """Timing statistics of one key; field names as in the reference (CuptiProfiler.h:34-45)."""

    __slots__ = ("min", "max", "median", "avg", "stddev", "num_calls")

    def __init__(self):
        self.num_calls = 0
        self.min = self.max = self.median = self.avg = self.stddev = math.nan

    def __str__(self) -> str:
        return (f" num calls: {self.num_calls}, min: {self.min}, max: {self.max}, median: {self.median}"
                f", avg: {self.avg}, stddev: {self.stddev}")

    __repr__ = __str__


class CuptiProfiler:
    """hipEvent-based stand-in for the CUPTI profiler object.  One live instance per process
    (the reference throws on a second one, CuptiProfiler.cpp:86-88)."""

    _live: Optional["weakref.ReferenceType[CuptiProfiler]"] = None

    def __init__(self, bufferSize: int = 1024 * 1024 * 8, numBuffers: int = 8, statsMaxLenPerKernel: int = 1024,
                 rings=None, max_keys: int = 64):
        live = CuptiProfiler._live() if CuptiProfiler._live is not None else None
        if live is not None and not live._closed:
            raise RuntimeError("Only one CuptiProfiler instance is allowed.")
        # bufferSize / numBuffers sized CUPTI's activity buffers; event pairs need no such pool
        self._owns_rings = rings is None
        if rings is None:
            rings = _backend_mod.get_backend().make_rings(1, max_keys, int(statsMaxLenPerKernel))
        self._rings = rings
        self._stats_max_len = int(statsMaxLenPerKernel)
        self._initialized = False
        self._started = False
        self._active_row: Optional[int] = None
        self._closed = False
        # device timestamps unless asked otherwise (or the backend cannot: the CPU test backend)
        self._stamps = os.environ.get("NVRX_GPU_TIMING", "stamp").strip().lower() != "event" and hasattr(rings, "stamp_begin")
        CuptiProfiler._live = weakref.ref(self)

    # ---- lifecycle -----------------------------------------------------------------------------
    def initialize(self) -> None:
        self._initialized = True

    def shutdown(self) -> None:
        self._initialized = False

    def close(self) -> None:
        """Release the singleton slot (the C++ destructor's job in the reference)."""
        if not self._closed:
            self._closed = True
            if self._owns_rings:
                self._rings.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ---- region timing ---------------------------------------------------------------------------
    def _stream_handle(self) -> int:
        return self._rings.backend.current_stream_handle()

    def start(self, key: str = DEFAULT_KEY) -> None:
        """Open a GPU-timed region on the current stream (reference: cuptiActivityEnable)."""
        if self._started:
            return  # reference prints "subsequent call" and carries on (CuptiProfiler.cpp:121-123)
        row = self._rings.row_for(_native.KIND_KERNEL, key)
        if not _warned_region_mode:
            _warn_region_mode_once()
        if self._stamps:
            self._rings.stamp_begin(row, self._stream_handle())
        else:
            self._rings.event_begin(row, self._stream_handle())
        self._active_row = row
        self._started = True

    def stop(self, cpu_row: int = -1, cpu_value: float = 0.0) -> bool:
        """Close the region opened by ``start`` (reference: cuptiActivityDisable).

        With device timestamps the closing kernel can also append ``cpu_value`` to ring row ``cpu_row``
        (the enclosing section's wall time); returns True when it did."""
        if not self._started:
            return False
        took = False
        if self._stamps:
            # (False: the stream is being captured into a hipGraph -- nothing was enqueued, the caller keeps its sample)
            took = self._rings.stamp_end(self._active_row, self._stream_handle(), cpu_row, cpu_value) and cpu_row >= 0
        else:
            self._rings.event_end(self._active_row, self._stream_handle())
        self._active_row = None
        self._started = False
        return took

    # ---- results -----------------------------------------------------------------------------------
    def harvest(self, wait: bool = True) -> int:
        return self._rings.harvest(wait)

    def active_rows(self) -> Dict[str, int]:
        """key -> ring row for the keys that currently hold samples."""
        r = self._rings
        return {k: row for k, row in r.kernel_row_names.items() if r.count(row) > 0}

    def get_stats(self) -> Dict[str, KernelStats]:
        """Wait for the recorded regions, then min/max/median/avg/stddev/num_calls per key."""
        self.harvest(wait=True)
        rows = self.active_rows()
        out: Dict[str, KernelStats] = {}
        if not rows:
            return out
        stats = self._rings.peek_stats()
        for key, row in rows.items():
            v = stats[row]
            ks = KernelStats()
            ks.min, ks.max, ks.median, ks.avg, ks.stddev = (float(v[i]) for i in range(5))
            ks.num_calls = int(v[5])
            out[key] = ks
        return out

    def reset(self) -> None:
        """Forget every recorded duration (reference: flush + map.clear(), CuptiProfiler.cpp:148-152)."""
        self.harvest(wait=True)
        for row in self._rings.kernel_row_names.values():
            self._rings.set_count(row, 0)
