"""Import-name alias: the reference's native module is ``nvrx_cupti_module``
(cupti_src/cupti_module_py.cpp:33).  On MI355X ``CuptiProfiler`` is, by GPU-timing mode (``nvrx_straggler.ktrace``):

* ``kernels`` (multi-rank jobs, ``NVRX_GPU_TIMING=kernels``): ``KernelTraceProfiler`` -- every kernel by name through
  rocprofiler-sdk, the reference's data model;
* ``stamp`` / ``event``: ``RegionProfiler`` -- one GPU-time row per profiled region.

The mode is settled once per process, so the name resolves to the same class every time it is asked for.
"""
from nvrx_straggler import ktrace as _ktrace
from nvrx_straggler.hip_profiler import CuptiProfiler as RegionProfiler, KernelStats  # noqa: F401
from nvrx_straggler.ktrace import KernelTraceProfiler  # noqa: F401


def __getattr__(name):
    if name == "CuptiProfiler":
        return KernelTraceProfiler if _ktrace.timing_mode() == "kernels" else RegionProfiler
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
