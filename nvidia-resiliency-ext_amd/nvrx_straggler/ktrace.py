"""Per-kernel GPU timing with real kernel names: the CUPTI-activity equivalent on MI355X.

``NVRX_GPU_TIMING=kernels`` selects this profiler instead of the per-region device timestamps of
``hip_profiler``.  It keeps the interface of the reference's native module (``CuptiProfiler`` with
``initialize / shutdown / start / stop / get_stats / reset``, cupti_src/cupti_module_py.cpp:33-55) and its
data model: while a profiled section is open every kernel the process launches is recorded under the key
``<mangled name>_blk_x_y_z_grid_x_y_z`` with its duration in microseconds (CuptiProfiler.cpp:186-191), so
the rank's GPU score is the kernel-weighted mean of reporting.py:219-253 over real kernels, and RCCL's
``ncclDev*`` kernels are left out exactly as in the reference (reporting.py:330-336).

Underneath, ``libnvrx_ktrace.so`` (include/nvrx_ktrace.h) is a rocprofiler-sdk tool: dispatch records are
collected on the SDK's thread, ``harvest()`` drains them and appends each kernel's durations to its DEVICE
ring row, and the statistics (mean-of-middles median, population stddev; CuptiProfiler.cpp:44-74) are
computed by the same HIP kernel as every other row.

The SDK only accepts tools before the HIP runtime initialises.  Importing ``nvrx_straggler`` with
``NVRX_GPU_TIMING=kernels`` set adds the library to ``ROCP_TOOL_LIBRARIES`` at import time, so the SDK picks it up
when HIP starts; if HIP was initialised earlier the profiler raises with the advice to set the variable (or the
import order) accordingly.
"""
from __future__ import annotations

import ctypes
import os
import threading
import warnings
import weakref
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_uint32, c_uint64
from typing import Dict, Optional

import numpy as np

from . import _native
from . import backend as _backend_mod
from .hip_profiler import KernelStats

_LIB_NAME = "libnvrx_ktrace.so"
# NVRX_LIB_DIR: load the native libraries from another directory (the sanitizer build of `make -C csrc asan` lives in
# lib_asan/; tools/run_sanitized.sh points here)
_LIB_PATH = os.path.join(os.environ.get("NVRX_LIB_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib"), _LIB_NAME)


class Record(Structure):
    _fields_ = [("key", c_uint32), ("us", c_float)]


RECORD_DTYPE = np.dtype([("key", np.uint32), ("us", np.float32)])
_ROW_UNKNOWN = -2  # key id without an entry in the key -> row table yet (-1: the rings had no row left for it)

# every symbol include/nvrx_ktrace.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("nvrx_ktrace_setup", c_int, [c_int]),
    ("nvrx_ktrace_ready", c_int, []),
    ("nvrx_ktrace_start", c_int, []),
    ("nvrx_ktrace_stop", c_int, []),
    ("nvrx_ktrace_flush", c_int, []),
    ("nvrx_ktrace_drain", c_int, [POINTER(Record), c_int]),
    ("nvrx_ktrace_pending", c_int, []),
    ("nvrx_ktrace_dropped", c_uint64, []),
    ("nvrx_ktrace_num_keys", c_int, []),
    ("nvrx_ktrace_key_name", c_char_p, [c_uint32]),
    ("nvrx_ktrace_reset", c_int, []),
    ("nvrx_ktrace_last_error", c_char_p, []),
]

_lib = None
_lock = threading.Lock()
_setup_error: Optional[str] = None


def lib_path() -> str:
    return _LIB_PATH


def load() -> ctypes.CDLL:
    """Load libnvrx_ktrace.so (once); raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise RuntimeError(f"{_LIB_NAME} not found at {_LIB_PATH}; build it with `make -C nvidia-resiliency-ext_amd/csrc`")
            lib = ctypes.CDLL(_LIB_PATH, mode=ctypes.RTLD_GLOBAL)
            for name, restype, argtypes in SYMBOLS:
                fn = getattr(lib, name)
                fn.restype = restype
                fn.argtypes = argtypes
            _lib = lib
    return _lib


def _check(rc: int) -> int:
    if rc < 0:
        msg = load().nvrx_ktrace_last_error()
        raise RuntimeError(f"nvrx_ktrace error {rc}: {msg.decode() if msg else '?'}")
    return rc


def setup(max_pending: int = 0) -> None:
    """Register the tool with rocprofiler-sdk.  Call before the first HIP call of the process.

    Default: name the library in ``ROCP_TOOL_LIBRARIES`` -- the SDK then loads it when the HIP runtime
    initialises, the same route ``rocprofv3`` uses for its own tool.  ``NVRX_KTRACE_FORCE=1`` registers
    immediately through ``rocprofiler_force_configure`` instead.  Either way the SDK's own start-up is slow the
    first time on a freshly booted machine (~50 s measured on ROCm 7.2 while it pages its libraries in, ~3 s
    afterwards); the default route pays that inside the first HIP call rather than inside ``import``."""
    global _setup_error
    if os.environ.get("NVRX_KTRACE_FORCE", "") == "1":
        try:
            _check(load().nvrx_ktrace_setup(int(max_pending)))
            _setup_error = None
        except RuntimeError as e:
            _setup_error = str(e)
            raise
        return
    libs = [p for p in os.environ.get("ROCP_TOOL_LIBRARIES", "").split(":") if p]
    if _LIB_PATH not in libs:
        if not os.path.exists(_LIB_PATH):
            _setup_error = f"{_LIB_NAME} not found at {_LIB_PATH}; build it with `make -C nvidia-resiliency-ext_amd/csrc`"
            raise RuntimeError(_setup_error)
        os.environ["ROCP_TOOL_LIBRARIES"] = ":".join(libs + [_LIB_PATH])
    _setup_error = None


_prefetch_thread: Optional[threading.Thread] = None
prefetch_stats: Dict[str, float] = {}


def _prefetch_gpu_libraries() -> Optional[threading.Thread]:
    """OPT-IN (``NVRX_KTRACE_PREFETCH=1``): read the large GPU libraries PyTorch links into the page cache, sequentially,
    on a background thread, before HIP starts.

    With a tool attached that asks for code-object callbacks, the first HIP call loads EVERY GPU code object of every
    loaded library instead of deferring them: 10.7 GB of ``read()`` calls on this image (libmagma 1.3 GB, MIOpen 0.95,
    rocsolver 0.76, libtorch_hip 0.42, rocsparse 0.39 ...), against 0.00 GB without a tool.  From a warm page cache
    that takes 3 s; on a box whose storage is cold 1.5-1.8 GB of it come from storage at 10-14 MB/s (process in D
    state, ``submit_bio_wait``) and the call takes 130-165 s -- the "rocprofiler-sdk start-up stall" of rounds 1-2
    (tools/debug/ktrace_stall_io.sh, ktrace_eager_load.py).  Reading the files front to back was measured at ~900 MB/s on
    one box (there the read-ahead turns minutes into seconds) but a box that is slow for sequential reads as well
    gains nothing and reads 5.3 GB instead of 1.8: a cold box still ran into the 150 s limit of the GPU test with the
    read-ahead on.  Hence opt-in, for deployments that know their storage."""
    if os.environ.get("NVRX_KTRACE_PREFETCH", "0") in ("0", ""):
        return None

    def run() -> None:
        import glob
        import importlib.util
        import time

        t0 = time.monotonic()
        total = 0
        try:
            spec = importlib.util.find_spec("torch")
            if spec is None or not spec.origin:
                return
            libdir = os.path.join(os.path.dirname(spec.origin), "lib")
            files = sorted(((os.path.getsize(f), f) for f in glob.glob(os.path.join(libdir, "*.so*")) if os.path.isfile(f)),
                           reverse=True)
            chunk = memoryview(bytearray(16 << 20))
            for size, path in files:
                if size < (32 << 20):
                    break
                with open(path, "rb", buffering=0) as f:
                    try:
                        os.posix_fadvise(f.fileno(), 0, 0, os.POSIX_FADV_SEQUENTIAL)
                    except (AttributeError, OSError):
                        pass
                    while True:
                        n = f.readinto(chunk)  # releases the GIL
                        if not n:
                            break
                        total += n
        except Exception:  # noqa: BLE001  (an optimisation: never in the way)
            pass
        finally:
            prefetch_stats["seconds"] = time.monotonic() - t0
            prefetch_stats["gigabytes"] = total / 1e9

    t = threading.Thread(target=run, name="nvrx-ktrace-prefetch", daemon=True)
    t.start()
    return t


def setup_from_env() -> None:
    """Import-time hook: register early when ``NVRX_GPU_TIMING=kernels`` (errors surface at first use)."""
    if os.environ.get("NVRX_GPU_TIMING", "") == "kernels":
        global _prefetch_thread
        try:
            setup()
            if _prefetch_thread is None:
                _prefetch_thread = _prefetch_gpu_libraries()
        except Exception:  # noqa: BLE001  (reported by KernelTraceProfiler.__init__)
            pass


def drain_all() -> np.ndarray:
    """Flush, then pop every pending record: structured array with fields ``key`` (u32) and ``us`` (f32)."""
    lib = load()
    _check(lib.nvrx_ktrace_flush())
    chunks = []
    cap = 1 << 16
    buf = (Record * cap)()
    while True:
        n = _check(lib.nvrx_ktrace_drain(buf, cap))
        if n == 0:
            break
        chunks.append(np.frombuffer(buf, dtype=RECORD_DTYPE, count=n).copy())
        if n < cap:
            break
    return np.concatenate(chunks) if chunks else np.empty(0, dtype=RECORD_DTYPE)


def key_name(key: int) -> str:
    name = load().nvrx_ktrace_key_name(int(key))
    return name.decode() if name else f"unknown_key_{key}"


class KernelTraceProfiler:
    """rocprofiler-sdk stand-in for the CUPTI profiler object; one live instance per process."""

    _live: Optional["weakref.ReferenceType[KernelTraceProfiler]"] = None

    def __init__(self, bufferSize: int = 1024 * 1024 * 8, numBuffers: int = 8, statsMaxLenPerKernel: int = 1024,
                 rings=None, max_keys: int = 4096):
        live = KernelTraceProfiler._live() if KernelTraceProfiler._live is not None else None
        if live is not None and not live._closed:
            raise RuntimeError("Only one CuptiProfiler instance is allowed.")
        self._lib = load()
        if _setup_error is not None and not self._lib.nvrx_ktrace_ready():
            raise RuntimeError(_setup_error)
        if not self._lib.nvrx_ktrace_ready():
            setup()  # no-op when the import hook already ran; too late if HIP is already up (checked in initialize)
        self._owns_rings = rings is None
        if rings is None:
            rings = _backend_mod.get_backend().make_rings(1, int(max_keys), int(statsMaxLenPerKernel))
        self._rings = rings
        self._initialized = False
        self._started = False
        self._closed = False
        self._key_rows: Dict[int, int] = {}  # tracer key id -> ring row (-1: no row left)
        self._row_table = np.full(1024, _ROW_UNKNOWN, dtype=np.int32)  # the same as an array indexed by key id
        self.keys_without_row = 0
        KernelTraceProfiler._live = weakref.ref(self)

    # ---- lifecycle -----------------------------------------------------------------------------
    def _ensure_ready(self) -> None:
        if not self._lib.nvrx_ktrace_ready():
            import torch

            if _prefetch_thread is not None:
                _prefetch_thread.join(timeout=60.0)  # (opt-in read-ahead: let it finish before HIP starts reading page by page)
            torch.cuda.init()  # the SDK calls the tool's initialiser when the runtime comes up
            if not self._lib.nvrx_ktrace_ready():
                raise RuntimeError(
                    "kernel tracing did not come up: rocprofiler-sdk was configured before nvrx_ktrace registered. "
                    "Import nvrx_straggler with NVRX_GPU_TIMING=kernels before the first HIP call, or set "
                    f"ROCP_TOOL_LIBRARIES={_LIB_PATH}"
                )

    def initialize(self) -> None:
        self._ensure_ready()
        self._initialized = True

    def shutdown(self) -> None:
        if self._started:
            self.stop()
        self._initialized = False

    def close(self) -> None:
        if not self._closed:
            self._closed = True
            if self._owns_rings:
                self._rings.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ---- tracing window ----------------------------------------------------------------------------
    def start(self, key: str = "") -> None:
        """Trace every kernel launched from now on (reference: cuptiActivityEnable); ``key`` is unused --
        kernels name themselves."""
        if self._started:
            return
        _check(self._lib.nvrx_ktrace_start())
        self._started = True

    def stop(self, cpu_row: int = -1, cpu_value: float = 0.0) -> bool:
        if not self._started:
            return False
        _check(self._lib.nvrx_ktrace_stop())
        self._started = False
        return False  # the section's wall-time sample is the caller's to push

    # ---- results -----------------------------------------------------------------------------------
    def harvest(self, wait: bool = True) -> int:
        """Move the recorded kernel durations into their device ring rows.  ``wait``: let the device finish
        first, as the reference does before it reads its statistics (straggler.py:234)."""
        if wait:
            import torch

            torch.cuda.synchronize()
        return self.ingest(drain_all())

    def ingest(self, recs: np.ndarray) -> int:
        """Append drained ``(key, us)`` records (``RECORD_DTYPE``, arrival order) to the device rings: every key id is
        looked up in a key -> ring-row table (rows are handed out the first time a key shows up: cold) and ALL records
        go to the device with one ``nvrx_ring_push_pairs`` call = one scatter launch, whether they belong to two
        kernel keys or to four thousand (data_shared test sizes of the reference: tests/straggler/unit/test_data_shared.py:62-66)."""
        if recs.size == 0:
            return 0
        keys = recs["key"]
        table = self._row_table
        top = int(keys.max())
        if top >= table.size:
            grown = np.full(max(top + 1, 2 * table.size), _ROW_UNKNOWN, dtype=np.int32)
            grown[: table.size] = table
            table = self._row_table = grown
        rows = table[keys]
        if (rows == _ROW_UNKNOWN).any():
            rings = self._rings
            for k in np.unique(keys[rows == _ROW_UNKNOWN]).tolist():
                try:
                    row = rings.row_for(_native.KIND_KERNEL, key_name(k))
                except RuntimeError:
                    row = -1
                    self.keys_without_row += 1
                    if self.keys_without_row == 1:
                        warnings.warn("straggler rings are full: further kernel names are not recorded "
                                      "(raise max_rows in Detector.initialize)")
                table[k] = row
                self._key_rows[k] = row
            rows = table[keys]
        self._rings.push_pairs(rows, recs["us"])
        return int(recs.size)

    def active_rows(self) -> Dict[str, int]:
        r = self._rings
        return {k: row for k, row in r.kernel_row_names.items() if r.count(row) > 0}

    def get_stats(self) -> Dict[str, KernelStats]:
        self.harvest(wait=True)
        rows = self.active_rows()
        out: Dict[str, KernelStats] = {}
        if not rows:
            return out
        stats = self._rings.peek_stats()
        for key, row in rows.items():
            v = stats[row]
            ks = KernelStats()
            ks.min, ks.max, ks.median, ks.avg, ks.stddev = (float(v[i]) for i in range(5))
            ks.num_calls = int(v[5])
            out[key] = ks
        return out

    def reset(self) -> None:
        _check(self._lib.nvrx_ktrace_reset())
        for row in self._rings.kernel_row_names.values():
            self._rings.set_count(row, 0)

    @property
    def dropped(self) -> int:
        return int(self._lib.nvrx_ktrace_dropped())
