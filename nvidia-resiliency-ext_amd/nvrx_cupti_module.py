"""Import-name alias: the reference's native module is ``nvrx_cupti_module``
(cupti_src/cupti_module_py.cpp:33); on MI355X it is the hipEvent profiler."""
from nvrx_straggler.hip_profiler import CuptiProfiler, KernelStats  # noqa: F401
from nvrx_straggler.ktrace import KernelTraceProfiler  # noqa: F401  (NVRX_GPU_TIMING=kernels)
