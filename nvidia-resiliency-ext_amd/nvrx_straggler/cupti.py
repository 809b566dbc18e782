"""Thread-safe, reference-counted front of the GPU-time profiler.

Keeps the reference's ``CuptiManager`` contract (cupti.py:19-95): a lock around every call into the
native profiler, ``start_profiling``/``stop_profiling`` nest through a counter so only the outermost
pair opens/closes a timed region, the same RuntimeErrors ("CuptiManager was not initialized", "No
active profiling run."), and ``get_results`` returns a copy.  The native object underneath is the
hipEvent profiler (``hip_profiler.CuptiProfiler``; import name ``nvrx_cupti_module`` kept as an alias).
"""
from __future__ import annotations

import os
import threading


class CuptiManager:
    def __init__(self, bufferSize=1_000_000, numBuffers=8, statsMaxLenPerKernel=4096, rings=None):
        """
        Args:
            bufferSize, numBuffers: accepted for signature compatibility (CUPTI activity buffers).
            statsMaxLenPerKernel: ring capacity per timed key; oldest samples are overwritten.
            rings: device rings to record into (the Detector shares its own); default: private rings.
        """
        import nvrx_cupti_module as profiler_module  # lazy, like the reference (cupti.py:35)

        kwargs = {} if rings is None else {"rings": rings}
        # NVRX_GPU_TIMING=kernels: per-kernel durations by kernel name through rocprofiler-sdk (ktrace.py);
        # otherwise one GPU-time row per profiled region (hip_profiler.py)
        self.per_kernel = os.environ.get("NVRX_GPU_TIMING", "stamp") == "kernels"
        profiler_cls = profiler_module.KernelTraceProfiler if self.per_kernel else profiler_module.CuptiProfiler
        self.cupti_ext = profiler_cls(
            bufferSize=bufferSize, numBuffers=numBuffers, statsMaxLenPerKernel=statsMaxLenPerKernel, **kwargs
        )
        self.is_initialized = False
        self.started_cnt = 0
        self.lock = threading.Lock()

    def _ensure_initialized(self):
        if not self.is_initialized:
            raise RuntimeError("CuptiManager was not initialized")

    def initialize(self):
        with self.lock:
            self.cupti_ext.initialize()
            self.is_initialized = True

    def shutdown(self):
        with self.lock:
            self.cupti_ext.shutdown()
            close = getattr(self.cupti_ext, "close", None)
            if close is not None:
                close()
            self.is_initialized = False
            self.started_cnt = 0

    def start_profiling(self, key=None):
        """Enter a GPU-timed region; only the outermost entry records the start event."""
        with self.lock:
            self._ensure_initialized()
            if self.started_cnt == 0:
                if key is None:
                    self.cupti_ext.start()
                else:
                    self.cupti_ext.start(key)
            self.started_cnt += 1

    def stop_profiling(self, cpu_row: int = -1, cpu_value: float = 0.0) -> bool:
        """Leave a GPU-timed region.  ``cpu_row`` / ``cpu_value`` (optional): a host-measured sample the
        closing device kernel may append for the caller; True is returned when it was taken."""
        with self.lock:
            self._ensure_initialized()
            if self.started_cnt <= 0:
                raise RuntimeError("No active profiling run.")
            self.started_cnt -= 1
            if self.started_cnt == 0:
                if cpu_row >= 0:
                    return bool(self.cupti_ext.stop(cpu_row, cpu_value))
                self.cupti_ext.stop()
            return False

    def harvest(self, wait: bool = True) -> int:
        """Bring every finished GPU measurement into the device rings (report time)."""
        with self.lock:
            self._ensure_initialized()
            return self.cupti_ext.harvest(wait)

    def get_results(self):
        with self.lock:
            self._ensure_initialized()
            return dict(self.cupti_ext.get_stats())

    def reset_results(self):
        with self.lock:
            self._ensure_initialized()
            self.cupti_ext.reset()
