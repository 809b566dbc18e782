"""``nvidia_resiliency_ext.attribution`` shim: only ``straggler`` lives here (see package docstring)."""
import sys as _sys

import nvrx_straggler as straggler
from nvidia_resiliency_ext import _alias

_alias(__name__ + ".straggler")
