"""Import-path shim: exposes the MI355X straggler package under the reference's module paths
(``nvidia_resiliency_ext.attribution.straggler`` -- current -- and ``nvidia_resiliency_ext.straggler``
-- the path the reference's docs still use) plus ``nvidia_resiliency_ext.ptl_resiliency`` for the
PyTorch-Lightning callback.  Only the straggler path exists here; the rest of nvidia-resiliency-ext
(fault tolerance, in-process restart, checkpointing, attribution services) is out of scope.
"""
import sys as _sys

import nvrx_straggler as _impl

_SUBMODULES = ("reporting", "straggler", "statistics", "name_mapper", "dist_utils", "interval_tracker", "cupti")


def _alias(prefix: str) -> None:
    _sys.modules[prefix] = _impl
    for _name in _SUBMODULES:
        _sys.modules[f"{prefix}.{_name}"] = getattr(_impl, _name)


_alias(__name__ + ".straggler")
straggler = _impl
