"""MI355X-native straggler detection: drop-in for ``nvidia_resiliency_ext.attribution.straggler``.

Same exports as the reference package (``__init__.py:16-18``): ``Report``, ``StragglerId``,
``Statistic``, ``CallableId``, ``Detector``; submodules ``reporting``, ``straggler``, ``statistics``,
``name_mapper``, ``dist_utils``, ``interval_tracker``, ``cupti`` keep their names.  The compute engine
is the in-tree HIP library (``nvrx_straggler/lib/libnvrx_straggler_hip.so``); it is loaded on first
use and there is no CPU fallback.
"""
from . import cupti, dist_utils, interval_tracker, name_mapper, reporting, statistics, straggler  # noqa: F401
from .reporting import Report, StragglerId  # noqa: F401
from .statistics import Statistic  # noqa: F401
from .straggler import CallableId, Detector  # noqa: F401

__all__ = ["Report", "StragglerId", "Statistic", "CallableId", "Detector"]
