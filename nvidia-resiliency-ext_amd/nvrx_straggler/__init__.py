"""MI355X-native straggler detection: drop-in for ``nvidia_resiliency_ext.attribution.straggler``.

Same exports as the reference package (``__init__.py:16-18``): ``Report``, ``StragglerId``,
``Statistic``, ``CallableId``, ``Detector``; submodules ``reporting``, ``straggler``, ``statistics``,
``name_mapper``, ``dist_utils``, ``interval_tracker``, ``cupti`` keep their names.  The compute engine
is the in-tree HIP library (``nvrx_straggler/lib/libnvrx_straggler_hip.so``); it is loaded on first
use and there is no CPU fallback.

``NVRX_GPU_TIMING`` selects how ``profile_cuda=True`` sections measure GPU time: ``stamp`` (default; device
timestamps around the region), ``event`` (hipEvent pair) or ``kernels`` (every kernel by name through
rocprofiler-sdk, the reference's CUPTI data model; see ``ktrace``).
"""
from . import ktrace as _ktrace

_ktrace.setup_from_env()  # NVRX_GPU_TIMING=kernels: register with rocprofiler-sdk before anything touches HIP

from . import cupti, dist_utils, interval_tracker, name_mapper, reporting, statistics, straggler  # noqa: F401
from .reporting import Report, StragglerId  # noqa: F401
from .statistics import Statistic  # noqa: F401
from .straggler import CallableId, Detector  # noqa: F401

__all__ = ["Report", "StragglerId", "Statistic", "CallableId", "Detector"]
