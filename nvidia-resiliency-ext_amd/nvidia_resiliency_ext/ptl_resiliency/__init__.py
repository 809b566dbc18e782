"""PyTorch-Lightning integration of the straggler path (reference: ptl_resiliency/__init__.py:23-26;
only ``StragglerDetectionCallback`` is in scope)."""
from .straggler_det_callback import StragglerDetectionCallback  # noqa: F401
