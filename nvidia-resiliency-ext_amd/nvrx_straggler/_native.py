"""ctypes binding of ``libnvrx_straggler_hip.so`` (C ABI: ``include/nvrx_straggler.h``).

There is deliberately no fallback here: if the HIP library is missing or cannot be loaded the
import of the product path fails with a clear error.  ``torch`` is imported first so that the
library resolves ``libamdhip64.so.7`` to the HIP runtime PyTorch-ROCm already loaded (one runtime
per process: device pointers and streams are shared with torch).
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_uint32, c_void_p

_LIB_NAME = "libnvrx_straggler_hip.so"
# NVRX_DEBUG_LIB_DIR: load the native libraries from another directory (the sanitizer build of `make -C csrc asan` lives in
# lib_asan/; tools/run_sanitized.sh points here)
_LIB_PATH = os.path.join(os.environ.get("NVRX_DEBUG_LIB_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib"), _LIB_NAME)

NVRX_ABI_VERSION = 2
STATS_STRIDE = 8
STAT_MIN, STAT_MAX, STAT_MED, STAT_AVG, STAT_STD, STAT_NUM, STAT_WEIGHT = range(7)
KIND_SECTION, KIND_KERNEL = 0, 1
MAX_RING_CAP = 65536
META_WORDS = 8
ERR_TIMEOUT = -62
ERR_RANGE = -34



class ReportDesc(ctypes.Structure):
    """``nvrx_report_desc`` (include/nvrx_straggler.h): one report's buffers and switches, filled once per shape."""

    _fields_ = [
        ("R", c_int32), ("K", c_int32), ("S", c_int32), ("names_ok", c_int32), ("rows_active", c_int32),
        ("do_indiv", c_int32), ("do_rel", c_int32), ("stats_rows", c_int32),
        ("thresholds", c_double * 4),
        ("d_stats", c_void_p), ("d_send", c_void_p), ("d_table", c_void_p),
        ("d_scores", c_void_p), ("d_flags", c_void_p), ("d_meta", c_void_p), ("d_stats_dst", c_void_p),
        ("d_done_counter", c_void_p),
        ("allgather_fn", c_void_p), ("comm", c_void_p), ("send_count", c_int32), ("seq", c_uint32),
        ("h_seq_word", c_void_p), ("timeout_s", c_double),
        ("order_after_stream", c_void_p), ("order_after_enabled", c_int32), ("resident", c_int32), ("prev_settled", c_int32), ("guard_rings", c_int32),
    ]


class WindowDesc(ctypes.Structure):
    """``nvrx_window_desc`` (include/nvrx_straggler.h): what ``nvrx_window_report`` does around the report itself."""

    _fields_ = [
        ("kt_sync", c_void_p), ("kt_hold", c_void_p), ("kt_counter", c_void_p), ("kt_patience_s", c_double),
        ("kt_rows_known", ctypes.c_uint64), ("kt_keys_without_row", ctypes.c_uint64),
        ("rows_used", c_int32), ("asynchronous", c_int32), ("harvest_regions", c_int32), ("out_names_ok", c_int32),
    ]


WINDOW_MISS, WINDOW_NAMES = 1, 2

# every symbol include/nvrx_straggler.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("nvrx_abi_version", c_int, []),
    ("nvrx_last_error", c_char_p, []),
    ("nvrx_row_stats", c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    ("nvrx_score", c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, POINTER(c_double), c_void_p, c_void_p, c_void_p,
                           c_void_p, c_uint32, c_void_p, c_void_p, c_int, c_void_p]),
    ("nvrx_ctx_create", c_int, [c_int, c_int, c_int, c_int, POINTER(c_void_p)]),
    ("nvrx_ctx_destroy", c_int, [c_void_p]),
    ("nvrx_ctx_set_stream", c_int, [c_void_p, c_void_p]),
    ("nvrx_ctx_info", c_int, [c_void_p, c_int]),
    ("nvrx_row_alloc", c_int, [c_void_p, c_int]),
    ("nvrx_row_configure", c_int, [c_void_p, c_int, c_int, c_int]),
    ("nvrx_ring_push", c_int, [c_void_p, c_int, c_float]),
    ("nvrx_ring_push_many", c_int, [c_void_p, c_int, c_void_p, c_int]),
    ("nvrx_ring_push_pairs", c_int, [c_void_p, c_void_p, c_void_p, c_int]),
    ("nvrx_ring_push_staged", c_int, [c_void_p, c_void_p, c_void_p, c_int]),
    ("nvrx_sink_push", c_int, [c_void_p, c_void_p, c_void_p, c_int]),
    ("nvrx_sink_row_alloc", c_int, [c_void_p, c_int]),
    ("nvrx_ring_push_device", c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p]),
    ("nvrx_ring_push_device_rows", c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    ("nvrx_ring_set_count", c_int, [c_void_p, c_int, c_int]),
    ("nvrx_ring_set_count_all", c_int, [c_void_p, c_int]),
    ("nvrx_ring_count", c_int, [c_void_p, c_int]),
    ("nvrx_ring_counts", c_int, [c_void_p, c_void_p, c_int]),
    ("nvrx_ring_occupancy_changed", c_int, [c_void_p, c_int]),
    ("nvrx_window_report", c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    ("nvrx_window_clocks", c_int, [POINTER(c_double)]),
    ("nvrx_ring_reset", c_int, [c_void_p]),
    ("nvrx_history_reset", c_int, [c_void_p, c_void_p]),
    ("nvrx_ring_flush", c_int, [c_void_p, c_void_p]),
    ("nvrx_ring_read", c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p]),
    ("nvrx_event_begin", c_int, [c_void_p, c_int, c_void_p]),
    ("nvrx_event_end", c_int, [c_void_p, c_int, c_void_p]),
    ("nvrx_event_harvest", c_int, [c_void_p, c_int]),
    ("nvrx_stamp_begin", c_int, [c_void_p, c_int, c_void_p]),
    ("nvrx_stamp_end", c_int, [c_void_p, c_int, c_int, c_float, c_void_p]),
    ("nvrx_report_local", c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    ("nvrx_report", c_int, [c_void_p, POINTER(ReportDesc), c_void_p]),
    ("nvrx_report_clocks", c_int, [POINTER(c_double)]),
    ("nvrx_report_desc_size", c_int, []),
    ("nvrx_peer_create", c_int, [c_int, c_int, c_int, c_int, POINTER(c_void_p)]),
    ("nvrx_peer_ipc_handle", c_int, [c_void_p, c_void_p]),
    ("nvrx_peer_connect", c_int, [c_void_p, c_int, c_void_p]),
    ("nvrx_peer_device_id", c_int, [c_void_p, c_char_p, c_int]),
    ("nvrx_peer_check_access", c_int, [c_void_p, c_int, c_char_p]),
    ("nvrx_peer_ready", c_int, [c_void_p, c_double]),
    ("nvrx_peer_allgather", c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_void_p]),
    ("nvrx_peer_allgather_address", c_void_p, []),
    ("nvrx_peer_error", c_int, [c_void_p, POINTER(c_uint32)]),
    ("nvrx_peer_destroy", c_int, [c_void_p]),
    ("nvrx_send_init", c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    ("nvrx_timing_enable", c_int, [c_void_p, c_int]),
    ("nvrx_timing_read", c_int, [c_void_p, POINTER(c_double), POINTER(c_int), c_int]),
    ("nvrx_host_alloc", c_int, [POINTER(c_void_p), POINTER(c_void_p), c_size_t]),
    ("nvrx_poll_u32", c_int, [c_void_p, c_uint32, c_double]),
    ("nvrx_host_free", c_int, [c_void_p]),
    ("nvrx_device_alloc", c_int, [c_void_p, c_size_t]),
    ("nvrx_device_free", c_int, [c_void_p]),
    ("nvrx_d2h_sync", c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    ("nvrx_copy_to_host", c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    ("nvrx_wait", c_int, [c_void_p]),
]

_lib = None
_lock = threading.Lock()


class NativeError(RuntimeError):
    """A call into libnvrx_straggler_hip.so failed (message from nvrx_last_error())."""


def lib_path() -> str:
    return _LIB_PATH


def load() -> ctypes.CDLL:
    """Load the HIP library (once).  Raises RuntimeError if it is missing -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_NAME} not found at {_LIB_PATH}. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C nvidia-resiliency-ext_amd/csrc`. "
                "The MI355X straggler path has no CPU fallback."
            )
        import torch  # noqa: F401  (load PyTorch-ROCm's HIP runtime first; see module docstring)

        try:
            lib = ctypes.CDLL(_LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        except OSError as e:  # pragma: no cover
            raise RuntimeError(f"failed to load {_LIB_PATH}: {e}") from e
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.nvrx_abi_version() != NVRX_ABI_VERSION:
            raise RuntimeError(f"{_LIB_NAME} ABI {lib.nvrx_abi_version()} != expected {NVRX_ABI_VERSION}")
        if lib.nvrx_report_desc_size() != ctypes.sizeof(ReportDesc):
            raise RuntimeError(f"nvrx_report_desc is {lib.nvrx_report_desc_size()} bytes in {_LIB_NAME}, "
                               f"{ctypes.sizeof(ReportDesc)} in the ctypes binding")
        _lib = lib
    return _lib


def check(rc: int) -> int:
    """Turn a negative return code into NativeError carrying the library's message."""
    if rc < 0:
        msg = load().nvrx_last_error()
        raise NativeError(f"nvrx error {rc}: {msg.decode() if msg else '?'}")
    return rc


def table_len(K: int, S: int) -> int:
    return 2 * (K + S) + K + 1


def score_len(S: int) -> int:
    return 2 + 2 * S
