"""Thread-safe, reference-counted front of the GPU-time profiler.

Keeps the reference's ``CuptiManager`` contract (cupti.py:19-95): a lock around every call into the
native profiler, ``start_profiling``/``stop_profiling`` nest through a counter so only the outermost
pair opens/closes a timed region, the same RuntimeErrors ("CuptiManager was not initialized", "No
active profiling run."), and ``get_results`` returns a copy.  The native object underneath is the
hipEvent profiler (``hip_profiler.CuptiProfiler``; import name ``nvrx_cupti_module`` kept as an alias).
"""
from __future__ import annotations

import functools
import os
import threading


def _serialised(needs_init: bool = True):
    """Every call into the native profiler holds the manager's lock; all but ``initialize`` / ``shutdown`` refuse to
    run before ``initialize`` (the reference's RuntimeError text)."""

    def wrap(method):
        @functools.wraps(method)
        def locked(self, *args, **kwargs):
            with self.lock:
                if needs_init and not self.is_initialized:
                    raise RuntimeError("CuptiManager was not initialized")
                return method(self, *args, **kwargs)

        return locked

    return wrap


class CuptiManager:
    def __init__(self, bufferSize=1_000_000, numBuffers=8, statsMaxLenPerKernel=4096, rings=None):
        """
        Args:
            bufferSize, numBuffers: accepted for signature compatibility (CUPTI activity buffers).
            statsMaxLenPerKernel: ring capacity per timed key; oldest samples are overwritten.
            rings: device rings to record into (the Detector shares its own); default: private rings.
        """
        import nvrx_cupti_module as profiler_module  # lazy, like the reference (cupti.py:35)

        # mode "kernels" (NVRX_GPU_TIMING=kernels, or a multi-rank job: ktrace.timing_mode): per-kernel durations by
        # kernel name through rocprofiler-sdk (ktrace.py); otherwise one GPU-time row per profiled region (hip_profiler.py)
        from . import ktrace as _ktrace

        self.per_kernel = _ktrace.timing_mode() == "kernels"
        profiler_cls = profiler_module.KernelTraceProfiler if self.per_kernel else profiler_module.RegionProfiler
        self._ext_args = dict(bufferSize=bufferSize, numBuffers=numBuffers, statsMaxLenPerKernel=statsMaxLenPerKernel,
                              **({} if rings is None else {"rings": rings}))
        self.cupti_ext = profiler_cls(**self._ext_args)
        self.lock = threading.Lock()
        self.is_initialized = False
        self.started_cnt = 0  # nesting depth of start_profiling(): only the outermost pair opens / closes a region

    @_serialised(needs_init=False)
    def initialize(self):
        self.cupti_ext.initialize()
        self.is_initialized = True

    @_serialised(needs_init=False)
    def shutdown(self):
        ext = self.cupti_ext
        ext.shutdown()
        getattr(ext, "close", lambda: None)()
        self.is_initialized, self.started_cnt = False, 0

    @_serialised()
    def start_profiling(self, key=None):
        """Enter a GPU-timed region; only the outermost entry records the start event."""
        self.started_cnt += 1
        if self.started_cnt == 1:
            try:
                self.cupti_ext.start(*(() if key is None else (key,)))
            except BaseException:
                self.started_cnt = 0
                raise

    @_serialised()
    def stop_profiling(self, cpu_row: int = -1, cpu_value: float = 0.0) -> bool:
        """Leave a GPU-timed region.  ``cpu_row`` / ``cpu_value`` (optional): a host-measured sample the
        closing device kernel may append for the caller; True is returned when it was taken."""
        if self.started_cnt <= 0:
            raise RuntimeError("No active profiling run.")
        self.started_cnt -= 1
        if self.started_cnt:
            return False  # an inner region of a nest: the outermost one is still open
        if cpu_row < 0:
            self.cupti_ext.stop()
            return False
        return bool(self.cupti_ext.stop(cpu_row, cpu_value))

    @_serialised()
    def switch_to_regions(self) -> bool:
        """Per-kernel tracing -> one GPU-time row per profiled region (the job's ranks did not all get per-kernel tracing:
        ``Detector`` agrees on the common mode at the first collective report).  The kernel rows recorded so far stay
        where they are and simply stop growing.  False when a region is open right now (try again at the next report)."""
        if not self.per_kernel:
            return True
        if self.started_cnt:
            return False
        import nvrx_cupti_module as profiler_module

        old = self.cupti_ext
        old.shutdown()
        shared = "rings" in self._ext_args
        if shared:
            old._owns_rings = False
        old.close()  # (takes the tracer's sink off the rings)
        if not shared:
            self._ext_args.pop("rings", None)
        self.cupti_ext = profiler_module.RegionProfiler(**self._ext_args)
        self.cupti_ext.initialize()
        self.per_kernel = False
        return True

    @_serialised()
    def harvest(self, wait: bool = True) -> int:
        """Bring every finished GPU measurement into the device rings (report time)."""
        return self.cupti_ext.harvest(wait)

    @_serialised()
    def get_results(self):
        return dict(self.cupti_ext.get_stats())  # a copy, as in the reference (cupti.py:88-89)

    @_serialised()
    def reset_results(self):
        self.cupti_ext.reset()
